#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database.  Usage: rocpd_pmc.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, counter_name, counter_value from pmc_events").fetchall()
    agg = {}
    for name, cn, v in rows:
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg.setdefault((name, cn), [0, 0.0])
        a[0] += 1
        a[1] += v
    lines = ["| kernel | counter | dispatches | avg per dispatch | total |", "|---|---|---|---|---|"]
    for (name, cn), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {name} | {cn} | {a[0]} | {a[1]/a[0]:.3f} | {a[1]:.1f} |")
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
