cd /root/repo
timeout 900 python -m pytest tests/test_classifier.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -8
python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', d['value'], d.get('ms_per_step'))"
