cd /root/repo
timeout 900 python -m pytest tests/test_hq_demo.py -x -q -m gpu -k cli 2>&1 | tail -25
