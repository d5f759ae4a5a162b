#!/bin/bash
# Round checkpoint on an MI355X box (run through gpurun from the repo root):
#   full GPU test suite, the headline bench line, a rocprofv3 kernel trace of the same command and the three PMC
#   passes over two forwards; summaries land in gpurun_out/ and are copied into profiles/ by hand.
# Every step runs under its own `timeout`: a faulting GPU once left rocprofv3 hanging for the whole remaining budget.
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
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.log; cat gpurun_out/bench_line.json
cd /tmp
rm -rf /root/repo/gpurun_out/prof_bench /root/repo/gpurun_out/pmc_mfma /root/repo/gpurun_out/pmc_fetch /root/repo/gpurun_out/pmc_write
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench -o c2 -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/prof_bench.log 2>&1
timeout -k 10 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /root/repo/gpurun_out/pmc_mfma -o p --output-format csv -- python /root/repo/tools/forward_once.py 2 > /root/repo/gpurun_out/pmc1.log 2>&1
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /root/repo/gpurun_out/pmc_fetch -o p --output-format csv -- python /root/repo/tools/forward_once.py 2 > /root/repo/gpurun_out/pmc2.log 2>&1
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /root/repo/gpurun_out/pmc_write -o p --output-format csv -- python /root/repo/tools/forward_once.py 2 > /root/repo/gpurun_out/pmc3.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find gpurun_out/prof_bench -name "*.db" | head -1) gpurun_out/prof_bench_summary.md > /dev/null; head -12 gpurun_out/prof_bench_summary.md
python tools/pmc_summary.py gpurun_out gpurun_out/pmc_dominant.json gpurun_out/pmc_dominant.md | tail -12
