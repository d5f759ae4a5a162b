#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as markdown.

    python tools/prof_summary.py <results.db> [out.md] [--after-marker finalize_psnr] [--forwards N]

`--after-marker S`: only the dispatches that START after the end of the LAST dispatch whose kernel name contains S
(tools/adm_fwd.py launches `finalize_psnr_kernel` between its set-up / warm-up and the profiled forwards), so the summary
holds the forward's own launches; with `--forwards N` it adds launches and idle time per forward (window span minus the
sum of kernel durations = launch bubbles)."""
import sqlite3
import sys


def main(db, out=None, top=14, marker=None, forwards=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    t0 = 0
    if marker:
        row = cur.execute("select max(end) from kernels where name like ?", (f"%{marker}%",)).fetchone()
        if row and row[0]:
            t0 = row[0]
    rows = list(cur.execute("select name, count(*), sum(duration) / 1e3, avg(duration) / 1e3 from kernels where start > ? "
                            "group by name order by sum(duration) desc", (t0,)))
    total = sum(r[2] for r in rows)
    ncalls = sum(r[1] for r in rows)
    title = db.split('/')[-2]
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({title})", ""]
    if marker:
        lines += [f"Window: dispatches after the last `{marker}` launch (set-up and warm-up excluded).", ""]
    lines += [f"total kernel time: {total / 1e3:.1f} ms over {ncalls} dispatches (durations in us)", ""]
    if forwards:
        span = cur.execute("select min(start), max(end) from kernels where start > ?", (t0,)).fetchone()
        wall = (span[1] - span[0]) / 1e3
        lines += [f"per forward ({forwards} forwards in the window): {ncalls / forwards:.0f} launches, "
                  f"{total / forwards / 1e3:.3f} ms of kernel time, {wall / forwards / 1e3:.3f} ms first-start-to-last-end "
                  f"({(wall - total) / forwards / 1e3:.3f} ms idle between kernels)", ""]
    lines += ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg in rows[:top]:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {calls} | {tot / 1e3:.2f} | {avg:.2f} | {100 * tot / total:.2f} |")
    rest = rows[top:]
    if rest:
        lines.append(f"| (other {len(rest)} kernels) | {sum(r[1] for r in rest)} | {sum(r[2] for r in rest) / 1e3:.2f} | | "
                     f"{100 * sum(r[2] for r in rest) / total:.2f} |")
    lines += ["", "## convolution dispatches by launch shape", "",
              "| kernel | workgroups x ksplit | LDS B | arch VGPR | calls | avg us | min us | max us |",
              "|---|---:|---:|---:|---:|---:|---:|---:|"]
    q = """select name, grid_x/workgroup_x, grid_y, lds_size, vgpr_count, count(*), avg(duration), min(duration), max(duration)
           from kernels where name like '%conv%' and start > ? group by name, grid_x, grid_y order by sum(duration) desc limit 28"""
    for name, gx, gy, lds, vg, n, avg, mn, mx in cur.execute(q, (t0,)):
        short = name.replace("(ConvArgs)", "").replace("void ", "")
        lines.append(f"| `{short}` | {gx} x {gy} | {lds} | {vg} | {n} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    argv = sys.argv[1:]
    marker = forwards = None
    if "--after-marker" in argv:
        i = argv.index("--after-marker")
        marker = argv[i + 1]
        del argv[i:i + 2]
    if "--forwards" in argv:
        i = argv.index("--forwards")
        forwards = int(argv[i + 1])
        del argv[i:i + 2]
    main(argv[0], argv[1] if len(argv) > 1 else None, marker=marker, forwards=forwards)
