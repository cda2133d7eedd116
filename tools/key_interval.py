#!/usr/bin/env python
"""Diagnostic: in a rocprofv3 --kernel-trace database of bench.py, what a key-frame interval looks like per queue: for the joint
BA's queue every solve from its first parse kernel (k_win_count) to the last kernel before the next idle period -- span, time in
kernels, number of dispatches -- and the kernel sequence of one solve; per-frame busy time of the tracker and pose queues.
Usage: key_interval.py results.db"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
ks = cur.execute("select name, queue_id, start, end from kernels order by start").fetchall()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]  # noqa: E731
t_lo = ks[0][2] + (ks[-1][3] - ks[0][2]) * 0.45
t_hi = ks[0][2] + (ks[-1][3] - ks[0][2]) * 0.95
byq = collections.defaultdict(list)
for name, q, s, e in ks:
    if t_lo <= s <= t_hi:
        byq[q].append((short(name), s, e))
for q, lst in sorted(byq.items()):
    names = collections.Counter(n for n, *_ in lst)
    busy = sum(e - s for _, s, e in lst) / 1000
    span = (lst[-1][2] - lst[0][1]) / 1000
    top = ", ".join(f"{n} x{c}" for n, c in names.most_common(4))
    print(f"queue {q}: {len(lst)} dispatches, busy {busy:.0f} us of {span:.0f} us ({100 * busy / span:.0f} %): {top}")
    if "k_track_rows_fused" in names:
        nfr = names["k_tail_nonmax_level0"] or 1
        print(f"   tracker queue: {busy / nfr:.1f} us busy per frame, {span / nfr:.1f} us per frame")
    if "k_handback" in names:
        nfr = names["k_handback"]
        print(f"   pose queue: {busy / nfr:.1f} us busy per frame, {span / nfr:.1f} us per frame")
        per = collections.Counter()
        for n, s, e in lst:
            per[n] += (e - s) / 1000
        print("   " + ", ".join(f"{n} {v / nfr:.1f}" for n, v in per.most_common(8)))
    if "k_solve_blocked" in names:
        # solves: the joint BA's start with the window parse (k_win_count); otherwise split at idle periods > 60 us
        if "k_win_count" in names:
            idx = [k for k, it in enumerate(lst) if it[0] == "k_win_count"]
            solves = [lst[a:b] for a, b in zip(idx, idx[1:])]
        else:
            solves, curs = [], [lst[0]]
            for prev, it in zip(lst, lst[1:]):
                if it[1] - prev[2] > 60_000:
                    solves.append(curs)
                    curs = []
                curs.append(it)
            solves.append(curs)
        solves = [sv for sv in solves if len(sv) > 10]
        if not solves:
            continue
        period = [(b[0][1] - a[0][1]) / 1000 for a, b in zip(solves, solves[1:])]
        sp = [(sv[-1][2] - sv[0][1]) / 1000 for sv in solves]
        bz = [sum(e - s for _, s, e in sv) / 1000 for sv in solves]
        med = lambda v: sorted(v)[len(v) // 2] if v else 0  # noqa: E731
        print(f"   {len(solves)} solves: start-to-start median {med(period):.0f} us; first-to-last-kernel span median {med(sp):.0f} us "
              f"(min {min(sp):.0f}, max {max(sp):.0f}), in kernels median {med(bz):.0f} us, dispatches median {med([len(sv) for sv in solves])}")
        order = sorted(range(len(solves)), key=lambda k: sp[k])
        sv = solves[order[len(order) // 2]]
        t0 = sv[0][1]
        print("   the solve of median span:")
        prev_e = t0
        for n, s, e in sv:
            print(f"     {(s - t0) / 1000:8.1f} us  {n:24s} dur {(e - s) / 1000:6.1f}  gap {(s - prev_e) / 1000:6.1f}")
            prev_e = e
