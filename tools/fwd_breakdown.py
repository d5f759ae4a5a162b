#!/usr/bin/env python
"""Per-forward kernel breakdown from a rocprofv3 --kernel-trace database of tools/adm_fwd.py: the launches between the
last two `nchw_to_nhwc_h16` dispatches (= one warmed-up forward), grouped by kernel and grid."""
import glob
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/prof_adm16/**/*.db", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "nchw_to_nhwc_h16"
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = con.execute(f"select d.start, d.end, s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {kd} d "
                   f"join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[2]]
a, b = idx[-2], idx[-1]
tot = defaultdict(lambda: [0, 0.0])
for r in rows[a:b]:
    key = (r[2][:46], r[3] // r[5], r[4])
    tot[key][0] += 1
    tot[key][1] += (r[1] - r[0]) / 1e3
busy = sum(v[1] for v in tot.values())
print(f"{b - a} launches, busy {busy / 1e3:.3f} ms, span {(rows[b][0] - rows[a][0]) / 1e6:.3f} ms")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:48s} wg {k[1]:6d} x {k[2]:2d}  n={v[0]:3d}  total {v[1]:8.1f} us  avg {v[1] / v[0]:7.1f}")
