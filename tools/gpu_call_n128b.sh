#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv16.py tests/test_classifier.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/n128b_tests.log
B=32 timeout 300 python tools/n128_bench.py 2>&1 | head -8 | tee gpurun_out/n128b_bench.log
for B in 8 32; do B=$B timeout 300 python tools/cls_step.py 6 2>&1 | tail -1 | tee -a gpurun_out/n128b_bench.log; done
