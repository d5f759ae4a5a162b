cd /root/repo
timeout 900 python -m pytest tests/test_ddnm_plus.py tests/test_deblur.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -8
