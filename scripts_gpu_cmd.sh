cd /root/repo
timeout 600 python -m pytest tests/test_cs.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -8
