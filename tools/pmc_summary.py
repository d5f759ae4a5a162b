#!/usr/bin/env python
"""Summarise the three rocprofv3 --pmc passes (MFMA busy / FETCH_SIZE / WRITE_SIZE) of tools/forward_once.py
for the dominant kernel.  FETCH_SIZE is doubled (gfx950 reports 1/2 of wide coalesced reads,
MI355X_MICROARCH.md section HBM); both sizes are in KiB."""
import json
import os
import sqlite3
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KERNEL = os.environ.get("PMC_KERNEL", "conv3x3_halo_f32_kernel<2, 2, 2, 2>")
KERNEL_RE = os.environ.get("PMC_KERNEL_RE")          # regular expression over kernel names (several template variants)
MARKER = os.environ.get("PMC_AFTER_MARKER")          # only dispatches after the last launch of this kernel
PASSES = os.environ.get("PMC_PASSES", "2 forwards at B=8 per PMC pass")


def pivot(path):
    df = pd.read_csv(path)
    if MARKER:
        hit = df[df.Kernel_Name.str.contains(MARKER, regex=False)]
        if len(hit):
            df = df[df.Dispatch_Id > hit.Dispatch_Id.max()]
    df = df[df.Kernel_Name.str.contains(KERNEL_RE, regex=True) if KERNEL_RE else df.Kernel_Name.str.contains(KERNEL, regex=False)]
    return df.pivot_table(index=["Dispatch_Id", "Grid_Size"], columns="Counter_Name", values="Counter_Value",
                          aggfunc="sum").reset_index()


def rocprof_avg_us(db, kernel):
    """Average duration of `kernel` in a rocprofv3 --kernel-trace --stats database (the number bench.py's
    event-based figure must agree with)."""
    if not db or not os.path.exists(db):
        return None
    import re
    cur = sqlite3.connect(db).cursor()
    t0 = 0
    if MARKER:
        row = cur.execute("select max(end) from kernels where name like ?", (f"%{MARKER}%",)).fetchone()
        t0 = row[0] if row and row[0] else 0
    tot, n = 0.0, 0
    for name, dur in cur.execute("select name, duration from kernels where start > ?", (t0,)):
        if (re.search(KERNEL_RE, name) if KERNEL_RE else kernel in name):
            tot += dur
            n += 1
    return tot / n / 1e3 if n else None        # kernels.duration is in ns


def main(root, out_json, out_md, stats_db=None):
    m, f, w = (pivot(f"{root}/pmc_{k}/p_counter_collection.csv") for k in ("mfma", "fetch", "write"))
    n = len(m)
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs issue MFMA
    util = (m.SQ_VALU_MFMA_BUSY_CYCLES / (1024.0 * m.GRBM_GUI_ACTIVE / 8.0))
    lds = {}
    if "SQ_LDS_BANK_CONFLICT" in m.columns and "SQ_LDS_IDX_ACTIVE" in m.columns:
        lds = {"lds_bank_conflict_frac": float(m.SQ_LDS_BANK_CONFLICT.sum() / max(1.0, m.SQ_LDS_IDX_ACTIVE.sum())),
               "lds_bank_conflict_note": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: extra LDS-array cycles per LDS-array cycle"}
    fetch_b = f.FETCH_SIZE * 1024.0 * 2.0
    write_b = w.WRITE_SIZE * 1024.0
    from ddnm_amd import build
    res = {
        "kernel": KERNEL_RE or KERNEL, "launches_in_pass": n, "passes": PASSES,
        # digest of the sources the profiled binary was built from: bench.py reports these figures only for a loaded
        # library with the same digest (the r01 file went stale the moment a kernel changed)
        "source_digest": build._digest(),
        "rocprof_avg_launch_us": rocprof_avg_us(stats_db, KERNEL),
        "mfma_busy_frac_mean": float(util.mean()), "mfma_busy_frac_weighted": float(
            m.SQ_VALU_MFMA_BUSY_CYCLES.sum() / (1024.0 * m.GRBM_GUI_ACTIVE.sum() / 8.0)),
        "hbm_fetch_bytes_per_launch": float(fetch_b.mean()), "hbm_write_bytes_per_launch": float(write_b.mean()),
        "hbm_bytes_per_launch": float(fetch_b.mean() + write_b.mean()),
        "note": "FETCH_SIZE x2 (gfx950 half-count of 16 B/lane reads), WRITE_SIZE exact (checked: 128->128 @256^2 "
                "launch writes 266240 KiB = output 262144 KiB + GN partials 4096 KiB)",
    }
    res.update(lds)
    # effective shader clock of the same launches: GRBM_GUI_ACTIVE / 8 XCDs / launch duration (kernel trace of the MFMA pass)
    try:
        import glob
        kt = pd.read_csv(glob.glob(f"{root}/pmc_mfma/*kernel_trace.csv")[0])[["Dispatch_Id", "Start_Timestamp", "End_Timestamp"]]
        mk = m.merge(kt, on="Dispatch_Id")
        ns = (mk.End_Timestamp - mk.Start_Timestamp).astype(float)
        res["effective_clock_ghz"] = float((mk.GRBM_GUI_ACTIVE / 8.0).sum() / ns.sum())
        res["effective_clock_note"] = "sum(GRBM_GUI_ACTIVE / 8) / sum(launch duration) over the launches of the MFMA pass; " \
                                      "2.4 GHz nominal -- lower = power limiting (MI355X_MICROARCH.md)"
    except Exception as e:      # noqa: BLE001
        res["effective_clock_ghz"] = None
        res["effective_clock_note"] = f"not available: {e!r}"
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
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
