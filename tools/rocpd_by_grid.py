#!/usr/bin/env python
"""Per-(kernel, grid size) duration summary of a rocprofv3 rocpd database: tells the joint BA's launches from the
inter-camera solve's, and active launches from early-exit no-ops.  Usage: rocpd_by_grid.py results.db [substr ...]"""
import collections
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2:] or ["k_"]
agg = collections.defaultdict(list)
for name, s, e, gx, lds in db.execute("select name,start,end,grid_x,lds_size from kernels"):
    nm = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if any(k in nm for k in want):
        agg[(nm, gx, lds)].append((e - s) / 1000)
print("| kernel | grid.x | LDS | calls | avg us | median | p10 | p90 | active (>2 us): n, avg |")
print("|---|---|---|---|---|---|---|---|---|")
for k, v in sorted(agg.items()):
    v = np.array(v)
    a = v[v > 2]
    print(f"| {k[0]} | {k[1]} | {k[2]} | {len(v)} | {v.mean():.1f} | {np.median(v):.1f} | {np.percentile(v, 10):.1f} | "
          f"{np.percentile(v, 90):.1f} | {len(a)}, {a.mean() if len(a) else 0:.1f} |")
