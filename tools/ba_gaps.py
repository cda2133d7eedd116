#!/usr/bin/env python
"""Diagnostic: in a rocprofv3 --kernel-trace database of bench.py, per BA queue the time in kernels vs the gaps in front of
each kind of kernel (queue empty or kernel waiting for resources).  Usage: ba_gaps.py results.db"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
ks = cur.execute("select name, queue_id, start, end, 0, 0 from kernels order by start").fetchall()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
t0 = ks[0][2] + (ks[-1][3] - ks[0][2]) * 0.4
byq = collections.defaultdict(list)
for name, q, s, e, gx, lds in ks:
    if s >= t0:
        byq[q].append((short(name), s, e, gx, lds))
for q, lst in byq.items():
    names = collections.Counter(n for n, *_ in lst)
    if "k_solve_blocked" not in names:
        continue
    sb = [(e - s) / 1000 for n, s, e, *_ in lst if n == "k_solve_blocked" and (e - s) > 3000]   # (skip the no-op launches)
    kind = "joint (order 144)" if sb and sorted(sb)[len(sb) // 2] > 17 else "inter-camera (order 48)"
    dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
    prev_end = None
    for n, s, e, gx, lds in lst:
        dur[n] += (e - s) / 1000
        cnt[n] += 1
        if prev_end is not None and (s - prev_end) < 400_000:   # ignore the idle time between solves
            gap[n] += max(0, s - prev_end) / 1000
        prev_end = e
    steps = cnt["k_solve_blocked"]
    print(f"queue {q}: {kind}, {steps} LM steps")
    tot_d = tot_g = 0
    for n in sorted(dur, key=lambda k: -dur[k]):
        if cnt[n] < steps // 4:
            continue
        print(f"   {n:22s} x{cnt[n]:5d}  avg dur {dur[n]/cnt[n]:7.1f} us   avg gap in front {gap[n]/cnt[n]:7.1f} us")
        tot_d += dur[n]; tot_g += gap[n]
    print(f"   per LM step: {tot_d/steps:.1f} us in kernels + {tot_g/steps:.1f} us of gaps")
