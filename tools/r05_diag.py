"""diagnostic: which (size, cameras, placement) combinations of the persistent tracker stay co-resident"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coslam_amd
from coslam_amd.synth import Scene

dev = torch.device("cuda:0")
def run(W, H, FW, FH, n, placed, nf=6):
    sc = Scene(min(n, 8), W, H, 6000 if W < 1000 else 14000, seed=5)
    frames = [[torch.from_numpy(sc.render(c % 8, f)).to(dev) for f in range(3)] for c in range(n)]
    cfg = coslam_amd.KLT_SequenceTrackerConfig(nIterations=10, nLevels=4, levelSkip=1, windowWidth=7, trackWithGain=1, minCornerness=3000.0,
                                                convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=4 if W < 1000 else 8)
    ts = []
    for c in range(n):
        t = coslam_amd.KLT_SequenceTracker(cfg, 0)
        t.allocate(W, H, 4, FW, FH)
        t.set_xcd_placement(placed)
        t.set_profile(True) if hasattr(t, "set_profile") else None
        ts.append(t)
    g = coslam_amd.KLT_TrackerGroup(ts)
    s = torch.cuda.Stream()
    g.set_stream(s.cuda_stream)
    d = [torch.zeros(FW * FH * 5, dtype=torch.int32, device=dev) for _ in range(n)]
    cn = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(n)]
    try:
        g.detect_dev([frames[c][0].data_ptr() for c in range(n)], [x.data_ptr() for x in d], [x.data_ptr() for x in cn]); g.advanceFrame()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.synchronize(); e0.record(s)
        for f in range(1, nf):
            g.redetect_dev([frames[c][f % 3].data_ptr() for c in range(n)], [x.data_ptr() for x in d], [x.data_ptr() for x in cn]); g.advanceFrame()
        e1.record(s); g.synchronize()
        res = f"ok {e0.elapsed_time(e1) * 1e3 / (nf - 1):8.1f} us/frame"
    except Exception as ex:
        res = "FAILED " + str(ex)[:80]
    print(f"{W}x{H} {FW * FH} slots x {n} cams placed={int(placed)}: {res}", flush=True)
    try:
        g.close()
        for t in ts: t.close()
    except Exception:
        pass

import sys
# how many workgroups of the 7 x 7 tracker one XCD really holds: 8 cameras placed one per XCD, slots = 32 x workgroups per camera
for wg in (63, 64, 65, 66, 72, 80, 96):
    run(640, 480, wg, 32, 8, True)
run(640, 480, 96, 32, 8, False)
for n in (2, 4):
    for placed in (False, True):
        run(1920, 1080, 100, 50, n, placed)
for placed in (False, True):
    run(640, 480, 50, 40, 13, placed)
