#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r05_gpu_tests.log
