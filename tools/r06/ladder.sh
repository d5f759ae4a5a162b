#!/bin/bash
# gpurun target: ceiling ladder (plain + one PMC cross-check pass) and the product kernels on the same box.
set +e
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
python - <<'PY' || exit 3
import torch
x = torch.randn(1 << 20, device='cuda'); assert torch.isfinite((x * 2).sum()).item()
PY
timeout 300 tools/_build/mfma_ceiling > gpurun_out/r06_ladder.txt 2>&1; cat gpurun_out/r06_ladder.txt
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/lad_pmc -o p --output-format csv -- /root/repo/tools/_build/mfma_ceiling > /root/repo/gpurun_out/r06_ladder_pmcrun.txt 2>&1
cd /root/repo
python - <<'PY' > gpurun_out/r06_ladder_pmc.txt 2>&1
import glob, pandas as pd
c = pd.read_csv(glob.glob("/tmp/lad_pmc/**/*counter_collection.csv", recursive=True)[0])
k = pd.read_csv(glob.glob("/tmp/lad_pmc/**/*kernel_trace.csv", recursive=True)[0])
p = c.pivot_table(index=["Dispatch_Id", "Kernel_Name"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
p = p.merge(k[["Dispatch_Id", "Start_Timestamp", "End_Timestamp"]], on="Dispatch_Id")
p["us"] = (p.End_Timestamp - p.Start_Timestamp) / 1e3
p["busy"] = p.SQ_VALU_MFMA_BUSY_CYCLES / (1024.0 * p.GRBM_GUI_ACTIVE / 8.0)
p["ghz"] = p.GRBM_GUI_ACTIVE / 8.0 / (p.us * 1e3)
p["ldsc"] = p.SQ_LDS_BANK_CONFLICT / p.SQ_LDS_IDX_ACTIVE.clip(lower=1)
p["lds_act"] = p.SQ_LDS_IDX_ACTIVE / (256.0 * p.GRBM_GUI_ACTIVE / 8.0)
# last 40 launches of each kernel = the timed ones
g = p.groupby("Kernel_Name", sort=False).tail(40).groupby("Kernel_Name", sort=False).agg(n=("us", "size"), us=("us", "mean"), busy=("busy", "mean"), ghz=("ghz", "mean"), lds_conflict=("ldsc", "mean"), lds_active=("lds_act", "mean"))
print(g.to_string(float_format=lambda v: f"{v:.3f}"))
PY
cat gpurun_out/r06_ladder_pmc.txt
timeout 600 python tools/s16_probe.py time > gpurun_out/r06_s16_time.txt 2>&1; tail -25 gpurun_out/r06_s16_time.txt
