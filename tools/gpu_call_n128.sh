#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv16.py -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/n128_tests.log
timeout 900 python -m pytest tests/test_classifier.py -q -m gpu -x -k "fp16 or h16 or attention16" 2>&1 | tail -8 | tee -a gpurun_out/n128_tests.log
for B in 8 32; do
  echo "== h16 n128 B=$B"; B=$B timeout 300 python tools/cls_step.py 6 2>&1 | tail -1 | tee -a gpurun_out/n128_time.log
  echo "== h16 wide B=$B"; DDNM_P16_N128=0 B=$B timeout 300 python tools/cls_step.py 6 2>&1 | tail -1 | tee -a gpurun_out/n128_time.log
done
cd /tmp
RAW=/tmp/ddnm_prof; rm -rf $RAW; mkdir -p $RAW
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_cls -o cls -- python /root/repo/tools/cls_step.py 5 > /root/repo/gpurun_out/prof_cls.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find $RAW/prof_cls -name "*.db" | head -1) gpurun_out/r05_cls_kernel_stats.md --after-marker finalize_psnr --forwards 5 > /dev/null; head -40 gpurun_out/r05_cls_kernel_stats.md
