#!/usr/bin/env python
"""Per-launch timeline of the LAST forward in a rocprofv3 kernel trace (after the marker): name, grid, duration, idle gap
in front of the launch.  Development tool: shows where a forward's time goes in launch order."""
import sqlite3
import sys


def main(db, marker="finalize_psnr", forwards=1):
    cur = sqlite3.connect(db).cursor()
    t0 = cur.execute("select max(end) from kernels where name like ?", (f"%{marker}%",)).fetchone()[0] or 0
    rows = list(cur.execute("select name, grid_x / workgroup_x, grid_y, start, end from kernels where start > ? order by start", (t0,)))
    n = len(rows) // forwards
    rows = rows[-n:]
    prev = None
    tot = gap = 0.0
    for name, gx, gy, s, e in rows:
        g = (s - prev) / 1e3 if prev is not None else 0.0
        short = name.replace("void ", "").split("(")[0][:48]
        print(f"{short:48s} {gx:6d}x{gy:<3d} {(e - s) / 1e3:9.1f} us   gap {g:7.1f}")
        tot += (e - s) / 1e3
        gap += max(g, 0.0)
        prev = e
    print(f"# {len(rows)} launches, kernel time {tot / 1e3:.3f} ms, gaps {gap / 1e3:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], forwards=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
