#!/usr/bin/env python
"""Diagnostic: for a rocprofv3 --kernel-trace --hip-trace database, the host submit time of each KLT kernel vs its start."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rcols = [r[1] for r in cur.execute("pragma table_info(regions)")]
print("regions cols:", rcols)
ks = cur.execute("select name, queue_id, start, end, corr_id, stack_id from kernels order by start").fetchall()
rs = cur.execute("select * from regions").fetchall()
ci = {c: i for i, c in enumerate(rcols)}
key = "corr_id" if "corr_id" in ci else ("stack_id" if "stack_id" in ci else None)
by = {}
for r in rs:
    nm = r[ci["name"]]
    if "Launch" in nm or "launch" in nm:
        by.setdefault(r[ci[key]], r)
t0 = ks[0][2]
trk = [k for k in ks if "track_gain" in k[0]]
tstart = trk[len(trk) // 2][2]
n = 0
for k in ks:
    if k[2] < tstart or k[1] == 4 and "linearize" not in k[0]:
        continue
    r = by.get(k[4]) or by.get(k[5])
    sub = (r[ci["start"]] - t0) / 1000 if r else float("nan")
    print(f"submit {sub:10.1f}  start {(k[2]-t0)/1000:10.1f}  end {(k[3]-t0)/1000:10.1f}  lag {((k[2]-t0)/1000-sub):8.1f}  q{k[1]} {k[0].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:26]}")
    n += 1
    if n > 70: break
