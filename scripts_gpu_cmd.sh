cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_adm.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/conv16_probe.py 2>&1 | tail -11
python tools/adm_probe.py 2>&1 | grep "fp16:" -A3
