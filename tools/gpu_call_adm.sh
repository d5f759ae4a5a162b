#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for B in 4 8; do
 for W in 0 1 0 1; do
  echo "== ADM h16 B=$B n128_wide=$W"; MODES=h16 B=$B DDNM_P16_N128_WIDE=$W timeout 300 python tools/adm_time.py 2>&1 | grep "ms / forward" | tee -a gpurun_out/adm_wide.log
 done
done
DDNM_P16_N128_WIDE=1 timeout 600 python -m pytest tests/test_gpu_conv16.py tests/test_gpu_adm.py -q -m gpu -x 2>&1 | tail -5
