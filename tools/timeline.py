#!/usr/bin/env python
"""Diagnostic: kernel timeline of a rocprofv3 --kernel-trace database around one key-frame interval: per queue, every
kernel's start / duration / gap to the previous kernel of the same queue.  Usage: timeline.py results.db [window_us] [skip]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
ks = cur.execute("select name, queue_id, start, end from kernels order by start").fetchall()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 3000.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:22]
t_first, t_last = ks[0][2], ks[-1][3]
t0 = t_first + (t_last - t_first) * skip
# align on the first joint-BA linearize after t0
for k in ks:
    if k[2] >= t0 and "k_linearize" in k[0] and "seg8" not in k[0]:
        t0 = k[2] - 50_000
        break
last_end = {}
for name, q, s, e in ks:
    if s < t0:
        last_end[q] = e
        continue
    if s > t0 + win * 1000:
        break
    gap = (s - last_end[q]) / 1000 if q in last_end else float("nan")
    print(f"{(s - t0) / 1000:9.1f} us  q{q:<3d} {short(name):22s} dur {(e - s) / 1000:7.1f}  gap {gap:7.1f}")
    last_end[q] = e
