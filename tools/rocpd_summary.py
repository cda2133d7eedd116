#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into the --stats style table:
per-kernel calls, total / average / min / max duration and share.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d = (e - s) / 1000.0
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {name} | {a[0]} | {a[1]:.1f} | {a[1]/a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100*a[1]/total:.1f} |")
    txt = "\n".join(lines) + f"\n\ntotal kernel time: {total:.1f} us over {len(rows)} dispatches\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
