#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
B=32 timeout 300 python tools/n128_bench.py 2>&1 | tee gpurun_out/n128_bench_b32.log
B=8 timeout 300 python tools/n128_bench.py 2>&1 | tee gpurun_out/n128_bench_b8.log
