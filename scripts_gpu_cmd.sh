cd /root/repo
timeout 900 python -m pytest tests/test_gpu_adm.py -x -q -m gpu 2>&1 | tail -5
python tools/adm_probe.py 2>&1 | grep "fp16:" -A6
