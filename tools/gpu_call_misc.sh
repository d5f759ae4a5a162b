#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_batch_plans.py tests/test_gpu_conv16.py -q -m gpu -x -k "auto_graphs or micro_batched or large_magnitudes or graph_replay" 2>&1 | tail -8 | tee gpurun_out/misc_tests.log
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-workloads 2> gpurun_out/bench_lat.err | tee gpurun_out/bench_lat.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']); print(json.dumps(d.get('latency'), indent=1))"
tail -3 gpurun_out/bench_lat.err
