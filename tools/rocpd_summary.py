#!/usr/bin/env python
"""Markdown summaries of a rocprofv3 result database (rocpd sqlite, ROCm 7):
    rocpd_summary.py kernels  <results.db> [--last-frames K --frame-kernel NAME]   per-kernel time table (whole run, or the region
                                                                                   spanned by the last K launches of NAME), plus
                                                                                   the same per stream
    rocpd_summary.py counters <results.db> [<results.db> ...]                      per (kernel, counter) averages of --pmc passes
Kernel names are shortened (anonymous namespace, argument lists)."""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name[:70]


def kernels(db_path, last_frames=0, frame_kernel="k_intracam"):
    cur = sqlite3.connect(db_path).cursor()
    where = ""
    if last_frames:
        rows = cur.execute(f"select start from kernels where name like '%{frame_kernel}%' order by start").fetchall()
        if len(rows) > last_frames:
            where = f" where start >= {rows[-last_frames][0]} and start <= {rows[-1][0]}"
            print(f"# region: the last {last_frames} launches of {frame_kernel} ({(rows[-1][0] - rows[-last_frames][0]) / 1e6:.1f} ms under the profiler)\n")
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels" + where +
                       " group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    print("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {short(r[0])} | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {100 * r[2] / tot:.1f} |")
    print("\n## per stream (busy time of the stream's kernels; the first kernel names it)\n")
    srows = cur.execute("select stream_id, count(*), sum(end-start)/1e3, min(start), max(end) from kernels" + where + " group by stream_id order by 3 desc").fetchall()
    print("| stream | kernels | busy us | span us | busiest kernels |\n|---|---|---|---|---|")
    for sid, n, busy, t0, t1 in srows:
        if n < 20:
            continue
        top = cur.execute("select name, sum(end-start) from kernels" + (where + " and " if where else " where ") + f"stream_id={sid} group by name order by 2 desc limit 3").fetchall()
        print(f"| {sid} | {n} | {busy:.0f} | {(t1 - t0) / 1e3:.0f} | {', '.join(short(t[0]) for t in top)} |")


def counters(paths):
    print("| kernel | counter | dispatches | avg per dispatch | total |\n|---|---|---|---|---|")
    out = []
    for p in paths:
        cur = sqlite3.connect(p).cursor()
        out += cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    for r in sorted(out, key=lambda r: -r[4]):
        if r[2] < 3 or r[0].startswith("void at::") or "rocclr" in r[0]:
            continue
        print(f"| {short(r[0])} | {r[1]} | {r[2]} | {r[3]:.3f} | {r[4]:.1f} |")


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    if sys.argv[1] == "kernels":
        lf = int(sys.argv[sys.argv.index("--last-frames") + 1]) if "--last-frames" in sys.argv else 0
        fk = sys.argv[sys.argv.index("--frame-kernel") + 1] if "--frame-kernel" in sys.argv else "k_intracam"
        kernels(sys.argv[2], lf, fk)
    else:
        counters([a for a in sys.argv[2:] if a.endswith(".db")])
