cd /root/repo
timeout 900 python -m pytest tests/test_gpu_adm.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -12
python tools/adm_probe.py 2>&1 | grep "fp16:" -A5
