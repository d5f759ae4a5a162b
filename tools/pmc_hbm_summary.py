#!/usr/bin/env python
"""Per-kernel HBM figures of the HBM-bound fringe: merges the FETCH_SIZE and WRITE_SIZE PMC passes
and the kernel-trace durations of tools/hbm_kernels.py into profiles/rNN_hbm_kernels.json:

    {kernel, launches, us (rocprofv3 average), bytes_pmc (FETCH_SIZE x2 + WRITE_SIZE, the guide's gfx950 correction),
     bytes_algorithmic (every operand once), frac_of_6.29TBps = bytes_algorithmic / (us * 6.29 TB/s),
     frac_pmc = bytes_pmc / (us * 6.29 TB/s)}

    python tools/pmc_hbm_summary.py <root with pmc_fetch/ pmc_write/ trace/> <algorithmic.json> <out.json> <out.md>"""
import glob
import json
import os
import re
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_TBPS = 6.29
MARKER = "finalize_psnr"


def after_marker(df, first_n=1):
    """Dispatches after the FIRST marker launch (the target launches finalize_psnr again inside its loop)."""
    hit = df[df.Kernel_Name.str.contains(MARKER, regex=False)]
    return df[df.Dispatch_Id > hit.Dispatch_Id.min()] if len(hit) else df


def counters(path, name):
    df = after_marker(pd.read_csv(path))
    df = df[df.Counter_Name == name]
    return df.groupby("Kernel_Name").Counter_Value.agg(["sum", "count"])


def main(root, alg_json, out_json, out_md):
    f = counters(glob.glob(f"{root}/pmc_fetch/*counter_collection.csv")[0], "FETCH_SIZE")
    w = counters(glob.glob(f"{root}/pmc_write/*counter_collection.csv")[0], "WRITE_SIZE")
    kt = pd.read_csv(glob.glob(f"{root}/trace/*kernel_trace.csv")[0])
    kt = after_marker(kt)
    kt["us"] = (kt.End_Timestamp - kt.Start_Timestamp) / 1e3
    dur = kt.groupby("Kernel_Name").us.agg(["mean", "count"])
    alg = json.load(open(alg_json))
    from ddnm_amd import build
    rows = []
    for kname in dur.index:
        key = next((k for k in alg if k in kname or k.replace("IDF16", "IDF16_") in kname), None)
        if key is None:
            continue
        us = float(dur.loc[kname, "mean"])
        fb = float(f.loc[kname, "sum"] / f.loc[kname, "count"]) * 1024.0 * 2.0 if kname in f.index else None
        wb = float(w.loc[kname, "sum"] / w.loc[kname, "count"]) * 1024.0 if kname in w.index else None
        pmc = None if fb is None or wb is None else fb + wb
        a = alg[key]
        short = re.sub(r"\(.*", "", kname)
        rows.append({"kernel": short, "launches": int(dur.loc[kname, "count"]), "us": round(us, 2),
                     "bytes_algorithmic": a, "bytes_pmc": pmc,
                     "frac_of_6.29TBps": None if a is None else round(a / (us * 1e-6) / (HBM_TBPS * 1e12), 4),
                     "frac_pmc": None if pmc is None else round(pmc / (us * 1e-6) / (HBM_TBPS * 1e12), 4)})
    rows.sort(key=lambda r: -r["us"] * r["launches"])
    res = {"source_digest": build._digest(), "hbm_achievable_TBps": HBM_TBPS,
           "passes": "tools/hbm_kernels.py (B = 8, 256 x 256; GroupNorm backward at 128 channels): rocprofv3 --kernel-trace, "
                     "--pmc FETCH_SIZE, --pmc WRITE_SIZE as separate runs; FETCH_SIZE x2 (gfx950), WRITE_SIZE exact",
           "kernels": rows}
    json.dump(res, open(out_json, "w"), indent=1)
    lines = ["# HBM-bound kernels of the hot path: achieved fraction of 6.29 TB/s", "", res["passes"], "",
             "| kernel | launches | avg us | algorithmic MB | PMC MB | algorithmic / (t x 6.29 TB/s) | PMC / (t x 6.29 TB/s) |",
             "|---|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        fm = lambda v, s=1e6: "--" if v is None else f"{v / s:.1f}"      # noqa: E731
        lines.append(f"| `{r['kernel']}` | {r['launches']} | {r['us']:.1f} | {fm(r['bytes_algorithmic'])} | {fm(r['bytes_pmc'])} | "
                     f"{'--' if r['frac_of_6.29TBps'] is None else r['frac_of_6.29TBps']} | "
                     f"{'--' if r['frac_pmc'] is None else r['frac_pmc']} |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:5])
