#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as markdown."""
import sqlite3
import sys


def main(db, out=None, top=14):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({db.split('/')[-2]})", "",
             f"total kernel time: {total / 1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches (top_kernels durations are in us)", "",
             "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows[:top]:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {calls} | {tot / 1e3:.2f} | {avg:.2f} | {pct:.2f} |")
    rest = rows[top:]
    if rest:
        lines.append(f"| (other {len(rest)} kernels) | {sum(r[1] for r in rest)} | {sum(r[2] for r in rest) / 1e3:.2f} | | "
                     f"{sum(r[4] for r in rest):.2f} |")
    lines += ["", "## convolution dispatches by launch shape", "",
              "| kernel | workgroups x ksplit | LDS B | arch VGPR | calls | avg us | min us | max us |",
              "|---|---:|---:|---:|---:|---:|---:|---:|"]
    q = """select name, grid_x/workgroup_x, grid_y, lds_size, vgpr_count, count(*), avg(duration), min(duration), max(duration)
           from kernels where name like '%conv%' group by name, grid_x, grid_y order by sum(duration) desc limit 24"""
    for name, gx, gy, lds, vg, n, avg, mn, mx in cur.execute(q):
        short = name.replace("(ConvArgs)", "").replace("void ", "")
        lines.append(f"| `{short}` | {gx} x {gy} | {lds} | {vg} | {n} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
