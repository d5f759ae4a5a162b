#!/usr/bin/env python
"""Summarise the three rocprofv3 --pmc passes (MFMA busy / FETCH_SIZE / WRITE_SIZE) of tools/forward_once.py
for the dominant kernel.  FETCH_SIZE is doubled (gfx950 reports 1/2 of wide coalesced reads,
MI355X_MICROARCH.md section HBM); both sizes are in KiB."""
import json
import sys

import pandas as pd

KERNEL = "conv3x3_halo_f32_kernel<2, 2, 2, 2>"


def pivot(path):
    df = pd.read_csv(path)
    df = df[df.Kernel_Name.str.contains(KERNEL, regex=False)]
    return df.pivot_table(index=["Dispatch_Id", "Grid_Size"], columns="Counter_Name", values="Counter_Value",
                          aggfunc="sum").reset_index()


def main(root, out_json, out_md):
    m, f, w = (pivot(f"{root}/pmc_{k}/p_counter_collection.csv") for k in ("mfma", "fetch", "write"))
    n = len(m)
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs issue MFMA
    util = (m.SQ_VALU_MFMA_BUSY_CYCLES / (1024.0 * m.GRBM_GUI_ACTIVE / 8.0))
    fetch_b = f.FETCH_SIZE * 1024.0 * 2.0
    write_b = w.WRITE_SIZE * 1024.0
    res = {
        "kernel": KERNEL, "launches_per_forward": n // 2, "passes": "2 forwards at B=8 per PMC pass",
        "mfma_busy_frac_mean": float(util.mean()), "mfma_busy_frac_weighted": float(
            m.SQ_VALU_MFMA_BUSY_CYCLES.sum() / (1024.0 * m.GRBM_GUI_ACTIVE.sum() / 8.0)),
        "hbm_fetch_bytes_per_launch": float(fetch_b.mean()), "hbm_write_bytes_per_launch": float(write_b.mean()),
        "hbm_bytes_per_launch": float(fetch_b.mean() + write_b.mean()),
        "note": "FETCH_SIZE x2 (gfx950 half-count of 16 B/lane reads), WRITE_SIZE exact (checked: 128->128 @256^2 "
                "launch writes 266240 KiB = output 262144 KiB + GN partials 4096 KiB)",
    }
    json.dump(res, open(out_json, "w"), indent=1)
    lines = ["# PMC passes on the dominant kernel (rocprofv3 --pmc, separate runs)", "",
             "```", json.dumps(res, indent=1), "```", "", "| grid (threads) | launches | MFMA busy | FETCH MB (x2) | WRITE MB |",
             "|---:|---:|---:|---:|---:|"]
    for gs in sorted(m.Grid_Size.unique()):
        mm, ff, ww = m[m.Grid_Size == gs], f[f.Grid_Size == gs], w[w.Grid_Size == gs]
        u = mm.SQ_VALU_MFMA_BUSY_CYCLES.sum() / (1024.0 * mm.GRBM_GUI_ACTIVE.sum() / 8.0)
        lines.append(f"| {gs} | {len(mm)} | {u:.3f} | {ff.FETCH_SIZE.mean() * 2 * 1024 / 1e6:.1f} | "
                     f"{ww.WRITE_SIZE.mean() * 1024 / 1e6:.1f} |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
