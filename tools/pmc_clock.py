#!/usr/bin/env python
"""MFMA-busy fraction, effective shader clock and LDS bank-conflict fraction per launch shape of one kernel, from ONE
rocprofv3 pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace
--output-format csv` (development tool).   python tools/pmc_clock.py <dir> <kernel substring>"""
import glob
import sys

import pandas as pd

root, pat = sys.argv[1], sys.argv[2]
c = pd.read_csv(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)[0])
k = pd.read_csv(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)[0])
c = c[c.Kernel_Name.str.contains(pat, regex=False)]
p = c.pivot_table(index=["Dispatch_Id", "Grid_Size"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
k = k[["Dispatch_Id", "Start_Timestamp", "End_Timestamp"]]
p = p.merge(k, on="Dispatch_Id")
p["us"] = (p.End_Timestamp - p.Start_Timestamp) / 1e3
p["busy"] = p.SQ_VALU_MFMA_BUSY_CYCLES / (1024.0 * p.GRBM_GUI_ACTIVE / 8.0)
p["ghz"] = p.GRBM_GUI_ACTIVE / 8.0 / (p.us * 1e3)
p["ldsc"] = p.SQ_LDS_BANK_CONFLICT / p.SQ_LDS_IDX_ACTIVE.clip(lower=1)
p["lds_act"] = p.SQ_LDS_IDX_ACTIVE / (256.0 * p.GRBM_GUI_ACTIVE / 8.0)
g = p.groupby("Grid_Size").agg(n=("us", "size"), us=("us", "mean"), busy=("busy", "mean"), ghz=("ghz", "mean"),
                               lds_conflict=("ldsc", "mean"), lds_active=("lds_act", "mean"))
print(g.to_string(float_format=lambda v: f"{v:.3f}"))
