#!/bin/bash
# gpurun target: rocprofv3 kernel traces of the celeba and ADM forwards (forwards only), summaries into gpurun_out/
set +e
cd /tmp; export TMPDIR=/tmp; RAW=/tmp/ddnm_prof; rm -rf $RAW; mkdir -p $RAW /root/repo/gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_c2fwd -o fw -- python /root/repo/tools/forward_once.py 5 > /root/repo/gpurun_out/prof_c2fwd.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_adm16 -o adm -- python /root/repo/tools/adm_fwd.py 5 > /root/repo/gpurun_out/prof_adm16.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find $RAW/prof_c2fwd -name "*.db" | head -1) gpurun_out/r06_celeba_forward_kernel_stats.md --after-marker finalize_psnr --forwards 5 > /dev/null; head -50 gpurun_out/r06_celeba_forward_kernel_stats.md
python tools/prof_summary.py $(find $RAW/prof_adm16 -name "*.db" | head -1) gpurun_out/r06_adm_fp16_forward_kernel_stats.md --after-marker finalize_psnr --forwards 5 > /dev/null; head -30 gpurun_out/r06_adm_fp16_forward_kernel_stats.md
