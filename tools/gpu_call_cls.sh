#!/bin/bash
# development call: fp16-activation classifier -- unit tests, parity tests, timing (both engines), kernel trace
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_classifier.py -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/cls_tests.log
timeout 600 python -m pytest tests/test_gpu_canary.py tests/test_gpu_batch_plans.py -q -m gpu -k classifier 2>&1 | tail -8 | tee -a gpurun_out/cls_tests.log
for B in 8 32; do
  echo "== h16 B=$B"; B=$B timeout 300 python tools/cls_step.py 6 2>&1 | tail -1 | tee -a gpurun_out/cls_time.log
  echo "== gen1 B=$B"; DDNM_CLS_GEN1=1 B=$B timeout 300 python tools/cls_step.py 6 2>&1 | tail -1 | tee -a gpurun_out/cls_time.log
done
cd /tmp
RAW=/tmp/ddnm_prof; rm -rf $RAW; mkdir -p $RAW
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_cls -o cls -- python /root/repo/tools/cls_step.py 5 > /root/repo/gpurun_out/prof_cls.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find $RAW/prof_cls -name "*.db" | head -1) gpurun_out/r05_cls_kernel_stats.md --after-marker finalize_psnr --forwards 5 > /dev/null; head -50 gpurun_out/r05_cls_kernel_stats.md
