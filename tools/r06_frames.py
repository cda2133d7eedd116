#!/usr/bin/env python
"""Diagnostic: every kernel of a few consecutive frames from a rocprofv3 --kernel-trace database, per stream, with its duration and the
gap to the previous kernel of the same stream.  Frames are delimited by the launches of k_intracam.
Usage: r06_frames.py results.db [first frame (counted from the end, default 120)] [frames (default 5)]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 120
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ks = cur.execute("select name, stream_id, start, end from kernels order by start").fetchall()


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", n)[:34]


marks = [k[2] for k in ks if "k_intracam" in k[0]]
t0, t1 = marks[-back], marks[-back + nfr]
# start at the tracker launch that feeds the first frame
last_end = {}
pose_stream = next(k[1] for k in ks if "k_intracam" in k[0])
print(f"# {nfr} frames, {(t1 - t0) / 1e3 / nfr:.1f} us per frame between k_intracam launches; pose stream = {pose_stream}")
busy = {}
for name, q, s, e in ks:
    if s < t0 - 300_000:
        last_end[q] = e
        continue
    if s > t1:
        break
    gap = (s - last_end[q]) / 1000 if q in last_end else float("nan")
    if s >= t0 - 200_000:
        print(f"{(s - t0) / 1000:9.1f} us  s{q:<3d} {short(name):34s} dur {(e - s) / 1000:7.1f}  gap {gap:7.1f}")
    if s >= t0:
        busy[q] = busy.get(q, 0.0) + (e - s) / 1000
    last_end[q] = e
print("# busy per stream per frame (us):", {q: round(v / nfr, 1) for q, v in busy.items()})
