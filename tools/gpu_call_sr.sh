#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_philox.py tests/test_gpu_full_configs.py -q -m gpu -x -k "srconv or in_kernel or sr_bicubic or c2" 2>&1 | tail -8 | tee gpurun_out/sr_tests.log
python __graft_entry__.py smoke 2>&1 | tail -1
for G in 0 1 0 1; do
 DDNM_SR_STEP_GEMM=$G timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-workloads --no-side-path --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GEMM route=$G', d['value'], d['ms_per_step'])"
done
