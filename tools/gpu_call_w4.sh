#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_s16.py -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/w4_tests.log
for W in 1 0 1 0; do
  echo "== celeba forward B=8 W4=$W"; DDNM_S16_W4=$W timeout 300 python tools/celeba_time.py 2>&1 | tail -2 | tee -a gpurun_out/w4_time.log
done
for W in 1 0; do echo "== s16_probe time W4=$W"; DDNM_S16_W4=$W timeout 300 python tools/s16_probe.py time 2>&1 | tail -18 | tee -a gpurun_out/w4_probe.log; done
