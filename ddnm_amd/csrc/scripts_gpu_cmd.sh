cd /root/repo
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -2
