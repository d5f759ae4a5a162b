#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40 | tee gpurun_out/full_gpu_tests.log
