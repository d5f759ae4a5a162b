#!/usr/bin/env python
"""Counter-level stall attribution of one kernel: where the wave cycles that are NOT MFMA issue go.

Two `rocprofv3 --pmc` passes (8 SQ counters each, separate runs) over the same target:
  pass A: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
          SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  pass B: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
          SQ_INST_CYCLES_VMEM_RD
MI355X_MICROARCH.md (SQ counters): WAIT_ANY (wave parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA
dependency / pipe busy; WAIT_INST_LDS is its LDS sub-bucket) + ACTIVE_INST_ANY (issuing) ~= WAVE_CYCLES, all in quad-cycles.

    PMC_KERNEL_RE=... PMC_AFTER_MARKER=finalize_psnr python tools/pmc_stalls.py <root with pmc_stallA/ pmc_stallB/> out.json out.md
"""
import glob
import json
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNEL_RE = os.environ["PMC_KERNEL_RE"]
MARKER = os.environ.get("PMC_AFTER_MARKER")
PASSES = os.environ.get("PMC_PASSES", "")


def pivot(path):
    df = pd.read_csv(path)
    if MARKER:
        hit = df[df.Kernel_Name.str.contains(MARKER, regex=False)]
        if len(hit):
            df = df[df.Dispatch_Id > hit.Dispatch_Id.max()]
    df = df[df.Kernel_Name.str.contains(KERNEL_RE, regex=True)]
    return df.pivot_table(index=["Dispatch_Id", "Grid_Size"], columns="Counter_Name", values="Counter_Value",
                          aggfunc="sum").reset_index()


def table(a, b):
    wc = a.SQ_WAVE_CYCLES.sum()
    row = {"launches": int(len(a)),
           "parked_waitcnt_barrier": a.SQ_WAIT_ANY.sum() / wc,
           "issue_stall": a.SQ_WAIT_INST_ANY.sum() / wc,
           "issue_stall_lds": a.SQ_WAIT_INST_LDS.sum() / wc,
           "issuing": a.SQ_ACTIVE_INST_ANY.sum() / wc,
           "issuing_valu_incl_mfma": a.SQ_ACTIVE_INST_VALU.sum() / wc,
           "issuing_lds": a.SQ_ACTIVE_INST_LDS.sum() / wc,
           "issuing_vmem": a.SQ_ACTIVE_INST_VMEM.sum() / wc}
    if b is not None and len(b):
        mf = b.SQ_INSTS_MFMA.sum()
        row.update({"valu_insts_per_mfma": (b.SQ_INSTS_VALU.sum() - mf) / max(1.0, mf),
                    "lds_insts_per_mfma": b.SQ_INSTS_LDS.sum() / max(1.0, mf),
                    "vmem_rd_insts_per_mfma": b.SQ_INSTS_VMEM_RD.sum() / max(1.0, mf),
                    "salu_insts_per_mfma": b.SQ_INSTS_SALU.sum() / max(1.0, mf),
                    "mfma_busy_of_sq_busy": b.SQ_VALU_MFMA_BUSY_CYCLES.sum() / max(1.0, 4.0 * b.SQ_BUSY_CYCLES.sum())})
    return {k: (round(float(v), 4) if not isinstance(v, int) else v) for k, v in row.items()}


def main(root, out_json, out_md):
    a = pivot(glob.glob(f"{root}/pmc_stallA/*counter_collection.csv")[0])
    bs = glob.glob(f"{root}/pmc_stallB/*counter_collection.csv")
    b = pivot(bs[0]) if bs else None
    from ddnm_amd import build
    res = {"kernel": KERNEL_RE, "passes": PASSES, "source_digest": build._digest(),
           "units": "fractions of SQ_WAVE_CYCLES (wave-resident quad-cycles, summed over all waves); parked + issue_stall + "
                    "issuing ~= 1; issue_stall_lds is a sub-bucket of issue_stall; issuing_* are sub-buckets of issuing",
           "all_launches": table(a, b), "by_grid": {}}
    for gs in sorted(a.Grid_Size.unique()):
        res["by_grid"][str(int(gs))] = table(a[a.Grid_Size == gs], None if b is None else b[b.Grid_Size == gs])
    json.dump(res, open(out_json, "w"), indent=1)
    keys = ["launches", "parked_waitcnt_barrier", "issue_stall", "issue_stall_lds", "issuing", "issuing_valu_incl_mfma",
            "issuing_lds", "issuing_vmem", "valu_insts_per_mfma", "lds_insts_per_mfma", "vmem_rd_insts_per_mfma"]
    lines = [f"# Stall attribution (rocprofv3 --pmc, two SQ passes): `{KERNEL_RE}`", "", PASSES, "", res["units"], "",
             "| grid (threads) | " + " | ".join(keys) + " |", "|---|" + "---:|" * len(keys)]
    for name, row in [("all", res["all_launches"])] + list(res["by_grid"].items()):
        lines.append(f"| {name} | " + " | ".join(str(row.get(k, "--")) for k in keys) + " |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:4])
