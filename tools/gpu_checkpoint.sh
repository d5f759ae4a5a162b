#!/bin/bash
# Round checkpoint on an MI355X box (run through gpurun from the repo root):
#   smoke, full GPU test suite, rocprofv3 kernel traces of the bench command and of the ADM fp16 forward (forwards only:
#   tools/adm_fwd.py / tools/forward_once.py launch a marker kernel after set-up and warm-up, the summaries cut there),
#   the PMC passes (separate runs) over the celeba forward (fp32 headline kernel) and the ADM forward (conv16), and LAST
#   the headline bench line (with the c3 / c4 / c5 workloads); summaries land in gpurun_out/ and the PMC json files are
#   installed into profiles/ before the bench line is taken (bench.py binds them by source digest).
# Every step runs under its own `timeout`: a faulting GPU once left rocprofv3 hanging for the whole remaining budget.
R=${ROUND:-r06}
set +e
python - <<'PY' || { echo 'GPU sanity check failed: not running the checkpoint on this box'; exit 3; }
import torch
x = torch.randn(1 << 20, device='cuda'); assert torch.isfinite((x * 2).sum()).item()
PY
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; echo 'smoke() failed: stopping before the long steps'; exit 4; }
tail -1 gpurun_out/smoke.log
if [ "$SKIP_TESTS" != "1" ]; then timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee gpurun_out/${R}_gpu_tests.log; fi
cd /tmp
# raw rocprofv3 output (databases, per-dispatch counter tables: > 64 MiB in all) stays under /tmp on the box: gpurun copies
# gpurun_out/ back only while it is small, and only the summaries are wanted
RAW=/tmp/ddnm_prof
rm -rf $RAW; mkdir -p $RAW
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $RAW/prof_bench -o c2 -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-workloads --no-side-path > /root/repo/gpurun_out/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_c2fwd -o fw -- python /root/repo/tools/forward_once.py 5 > /root/repo/gpurun_out/prof_c2fwd.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_adm16 -o adm -- python /root/repo/tools/adm_fwd.py 5 > /root/repo/gpurun_out/prof_adm16.log 2>&1
for k in mfma fetch write; do
  case $k in mfma) C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE";; fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
  timeout -k 10 420 rocprofv3 --pmc $C --kernel-trace -d $RAW/pmc_c2/pmc_$k -o p --output-format csv -- python /root/repo/tools/forward_once.py 2 > /root/repo/gpurun_out/pmc_c2_$k.log 2>&1
  timeout -k 10 420 rocprofv3 --pmc $C --kernel-trace -d $RAW/pmc16/pmc_$k -o p --output-format csv -- python /root/repo/tools/adm_fwd.py 2 > /root/repo/gpurun_out/pmc16_$k.log 2>&1
  # the same at B = 8 (the batch the class-conditional workload c5 runs): its own PMC file, bench.py picks it by batch
  ADM_B=8 timeout -k 10 420 rocprofv3 --pmc $C --kernel-trace -d $RAW/pmc16b8/pmc_$k -o p --output-format csv -- python /root/repo/tools/adm_fwd.py 2 > /root/repo/gpurun_out/pmc16b8_$k.log 2>&1
done
ADM_B=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_adm16b8 -o adm -- python /root/repo/tools/adm_fwd.py 3 > /root/repo/gpurun_out/prof_adm16b8.log 2>&1
# classifier guidance (fp16-activation path): kernel traces at B = 8 and at the grouped batch B = 32
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_cls -o cls -- python /root/repo/tools/cls_step.py 5 > /root/repo/gpurun_out/prof_cls.log 2>&1
B=32 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_cls32 -o cls -- python /root/repo/tools/cls_step.py 3 > /root/repo/gpurun_out/prof_cls32.log 2>&1
# counter-level stall attribution (two SQ passes each): headline kernel, conv16, the 256 x 128 tile of the classifier
CA="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
CB="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD"
for T in c2 adm cls; do
  case $T in c2) CMD="python /root/repo/tools/forward_once.py 2";; adm) CMD="python /root/repo/tools/adm_fwd.py 2";; cls) CMD="python /root/repo/tools/cls_step.py 2";; esac
  timeout -k 10 300 rocprofv3 --pmc $CA --kernel-trace -d $RAW/st_$T/pmc_stallA -o p --output-format csv -- $CMD > /root/repo/gpurun_out/pmc_${T}_A.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc $CB --kernel-trace -d $RAW/st_$T/pmc_stallB -o p --output-format csv -- $CMD > /root/repo/gpurun_out/pmc_${T}_B.log 2>&1
done
# HBM-bound fringe: per-kernel bytes and durations
timeout -k 10 300 rocprofv3 --kernel-trace -d $RAW/hbm/trace -o p --output-format csv -- python /root/repo/tools/hbm_kernels.py 3 > /root/repo/gpurun_out/hbm_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/hbm/pmc_fetch -o p --output-format csv -- python /root/repo/tools/hbm_kernels.py 3 > /root/repo/gpurun_out/hbm_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/hbm/pmc_write -o p --output-format csv -- python /root/repo/tools/hbm_kernels.py 3 > /root/repo/gpurun_out/hbm_write.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find $RAW/prof_cls -name "*.db" | head -1) gpurun_out/${R}_cls_kernel_stats.md --after-marker finalize_psnr --forwards 5 > /dev/null; head -22 gpurun_out/${R}_cls_kernel_stats.md
python tools/prof_summary.py $(find $RAW/prof_cls32 -name "*.db" | head -1) gpurun_out/${R}_cls_b32_kernel_stats.md --after-marker finalize_psnr --forwards 3 > /dev/null; head -12 gpurun_out/${R}_cls_b32_kernel_stats.md
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv3x3_s16_persist_kernel' PMC_PASSES="2 celeba forwards at B=8 per pass (tools/forward_once.py)" python tools/pmc_stalls.py $RAW/st_c2 gpurun_out/${R}_pmc_stalls_headline.json gpurun_out/${R}_pmc_stalls_headline.md | tail -7
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv16_kernel<9, [24], 4>' PMC_PASSES="2 ADM fp16 forwards at B=4 per pass (tools/adm_fwd.py)" python tools/pmc_stalls.py $RAW/st_adm gpurun_out/${R}_pmc_stalls_conv16.json gpurun_out/${R}_pmc_stalls_conv16.md | tail -8
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv16_n128_kernel' PMC_PASSES="2 classifier-guidance evaluations at B=8 per pass (tools/cls_step.py)" python tools/pmc_stalls.py $RAW/st_cls gpurun_out/${R}_pmc_stalls_n128.json gpurun_out/${R}_pmc_stalls_n128.md | tail -5
python tools/pmc_hbm_summary.py $RAW/hbm gpurun_out/hbm_kernels_algorithmic.json gpurun_out/${R}_hbm_kernels.json gpurun_out/${R}_hbm_kernels.md | tail -20
cp gpurun_out/${R}_cls_kernel_stats.md gpurun_out/${R}_cls_b32_kernel_stats.md gpurun_out/${R}_pmc_stalls_*.json gpurun_out/${R}_pmc_stalls_*.md gpurun_out/${R}_hbm_kernels.json gpurun_out/${R}_hbm_kernels.md profiles/ 2>/dev/null
BDB=$(find $RAW/prof_bench -name "*.db" | head -1); ADB=$(find $RAW/prof_adm16 -name "*.db" | head -1)
python tools/prof_summary.py $BDB gpurun_out/${R}_bench_kernel_stats.md > /dev/null; head -12 gpurun_out/${R}_bench_kernel_stats.md
python tools/prof_summary.py $ADB gpurun_out/${R}_adm_fp16_forward_kernel_stats.md --after-marker finalize_psnr --forwards 5 > /dev/null; head -24 gpurun_out/${R}_adm_fp16_forward_kernel_stats.md
python tools/prof_summary.py $(find $RAW/prof_c2fwd -name "*.db" | head -1) gpurun_out/${R}_celeba_forward_kernel_stats.md --after-marker finalize_psnr --forwards 5 > /dev/null; head -24 gpurun_out/${R}_celeba_forward_kernel_stats.md
python tools/fwd_timeline.py $ADB 5 > gpurun_out/${R}_adm_timeline.txt; tail -1 gpurun_out/${R}_adm_timeline.txt
# dominant kernel of the headline workload: the split-fp16 form of the 3x3 halo kernel (template arguments .., SRC16 = false,
# SPLIT = true, ASCALE = either) and its persistent form
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv3x3_s16_persist_kernel' python tools/pmc_summary.py $RAW/pmc_c2 gpurun_out/${R}_pmc_dominant_kernel.json gpurun_out/${R}_pmc_dominant_kernel.md $BDB | tail -8
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv16_kernel<9, [24], 4>' PMC_PASSES="2 ADM forwards (fp16 path) at B=4 per PMC pass, forwards only" python tools/pmc_summary.py $RAW/pmc16 gpurun_out/${R}_adm_pmc_conv16.json gpurun_out/${R}_adm_pmc_conv16.md $ADB | tail -8
B8DB=$(find $RAW/prof_adm16b8 -name "*.db" | head -1)
PMC_AFTER_MARKER=finalize_psnr PMC_KERNEL_RE='conv16_kernel<9, [24], 4>' PMC_PASSES="2 ADM forwards (fp16 path) at B=8 per PMC pass, forwards only" python tools/pmc_summary.py $RAW/pmc16b8 gpurun_out/${R}_adm_pmc_conv16_b8.json gpurun_out/${R}_adm_pmc_conv16_b8.md $B8DB | tail -4
cp gpurun_out/${R}_adm_pmc_conv16_b8.json gpurun_out/${R}_adm_pmc_conv16_b8.md profiles/
# bench.py reports HBM traffic / MFMA-busy only from a PMC summary stamped with the digest of the library it loads
# (profiles/*_pmc_dominant_kernel.json, profiles/*_adm_pmc_conv16.json): install this run's summaries first
# the stall table of the headline kernel travels inside its PMC file as well (`stall_attribution`)
python - <<PY
import json
d = json.load(open("gpurun_out/${R}_pmc_dominant_kernel.json")); st = json.load(open("gpurun_out/${R}_pmc_stalls_headline.json"))
if d.get("source_digest") == st.get("source_digest"):
    d["stall_attribution"] = {"source": "profiles/${R}_pmc_stalls_headline.json", "units": st["units"], "all_launches": st["all_launches"], "by_grid": st["by_grid"]}
    json.dump(d, open("gpurun_out/${R}_pmc_dominant_kernel.json", "w"), indent=1)
PY
cp gpurun_out/${R}_pmc_dominant_kernel.json gpurun_out/${R}_pmc_dominant_kernel.md gpurun_out/${R}_adm_pmc_conv16.json gpurun_out/${R}_adm_pmc_conv16.md profiles/
timeout 1200 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 > gpurun_out/${R}_bench_line.json 2> gpurun_out/bench.log; cat gpurun_out/${R}_bench_line.json
