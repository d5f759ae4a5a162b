set -x
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_adm.py -x -q 2>&1 | tail -5
echo PREPASS; timeout 300 python tools/conv16_probe.py 2>&1 | tail -11
echo FUSED; DDNM_F16_PREPASS_MIN_COUT=100000 timeout 300 python tools/conv16_probe.py 2>&1 | tail -11
timeout 300 python tools/adm_probe.py 2>&1 | tail -7
