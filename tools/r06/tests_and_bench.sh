#!/bin/bash
# gpurun target: full GPU test suite, then the default bench line.
set +e
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/r06_gpu_tests.log
timeout 1200 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 > gpurun_out/r06_bench_line.json 2> gpurun_out/bench.log; cat gpurun_out/r06_bench_line.json
