cd /root/repo
python bench.py --workload c3 --steps 1 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_c3.json
python bench.py --workload c4 --steps 1 --warmup 0 2>&1 | tail -1 | tee gpurun_out/bench_c4.json
