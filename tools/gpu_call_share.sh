#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_classifier.py -q -m gpu -x 2>&1 | tail -4
for S in 1 0 1 0; do
  DDNM_CLS_SHARE_PREFIX=$S timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('share=$S', d['value'], d['ms_per_step'])"
done
