#!/bin/bash
# stall-attribution PMC passes (headline kernel, conv16, n128) + HBM-fringe passes
set +e
cd /tmp
export TMPDIR=/tmp
RAW=/tmp/ddnm_pmc; rm -rf $RAW; mkdir -p $RAW /root/repo/gpurun_out
CA="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
CB="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD"
for T in c2 adm cls; do
  case $T in c2) CMD="python /root/repo/tools/forward_once.py 2";; adm) CMD="python /root/repo/tools/adm_fwd.py 2";; cls) CMD="python /root/repo/tools/cls_step.py 2";; esac
  timeout -k 10 300 rocprofv3 --pmc $CA --kernel-trace -d $RAW/$T/pmc_stallA -o p --output-format csv -- $CMD > /root/repo/gpurun_out/pmc_${T}_A.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc $CB --kernel-trace -d $RAW/$T/pmc_stallB -o p --output-format csv -- $CMD > /root/repo/gpurun_out/pmc_${T}_B.log 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace -d $RAW/hbm/trace -o p --output-format csv -- python /root/repo/tools/hbm_kernels.py 3 > /root/repo/gpurun_out/hbm_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/hbm/pmc_fetch -o p --output-format csv -- python /root/repo/tools/hbm_kernels.py 3 > /root/repo/gpurun_out/hbm_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/hbm/pmc_write -o p --output-format csv -- python /root/repo/tools/hbm_kernels.py 3 > /root/repo/gpurun_out/hbm_write.log 2>&1
cd /root/repo
R=${ROUND:-r05}
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv3x3_halo_f16_kernel<4, 2, 2, 2, false, true, (?:false|true), false>' PMC_PASSES="2 celeba forwards at B=8 per pass (tools/forward_once.py)" python tools/pmc_stalls.py $RAW/c2 gpurun_out/${R}_pmc_stalls_headline.json gpurun_out/${R}_pmc_stalls_headline.md
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv16_kernel<9, [24], 4>' PMC_PASSES="2 ADM fp16 forwards at B=4 per pass (tools/adm_fwd.py)" python tools/pmc_stalls.py $RAW/adm gpurun_out/${R}_pmc_stalls_conv16.json gpurun_out/${R}_pmc_stalls_conv16.md
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv16_n128_kernel' PMC_PASSES="2 classifier-guidance evaluations at B=8 per pass (tools/cls_step.py)" python tools/pmc_stalls.py $RAW/cls gpurun_out/${R}_pmc_stalls_n128.json gpurun_out/${R}_pmc_stalls_n128.md
python tools/pmc_hbm_summary.py $RAW/hbm gpurun_out/hbm_kernels_algorithmic.json gpurun_out/${R}_hbm_kernels.json gpurun_out/${R}_hbm_kernels.md
ls $RAW/*/* | head -30
