cd /root/repo
timeout 600 python -m pytest tests/test_gpu_cli.py -m gpu -q -k "class_conditional" 2>&1 | tail -5
python bench.py --workload c5 --steps 1 --warmup 0 2>&1 | tail -1 | tee gpurun_out/bench_c5.json
