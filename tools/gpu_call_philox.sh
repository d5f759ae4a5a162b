#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_philox.py tests/test_gpu_sampler.py tests/test_gpu_s16.py -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/philox_tests.log
timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu -x -k "self_launches or four_ranks" 2>&1 | tail -8 | tee -a gpurun_out/philox_tests.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-workloads --no-side-path 2> gpurun_out/bench_ph.err | tee gpurun_out/bench_ph.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('noise'))"
DDNM_NOISE=torch timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-workloads --no-side-path --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torch noise tape?', d['value'], d['ms_per_step'])"
tail -3 gpurun_out/bench_ph.err
