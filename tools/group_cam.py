#!/usr/bin/env python
"""Diagnostic: KLT stage of N cameras on ONE MI355X driven as a camera group (cs_klt_group_*): frames/s where one frame =
every camera's image consumed, tracker-kernel time (HIP events) and the per-wave cycle breakdown of the persistent kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coslam_amd
from coslam_amd.synth import Scene

dev = torch.device("cuda:0")
W, H, L, FW, FH = 640, 480, 4, 50, 40
NF = 12
CFG5 = os.environ.get("GROUP_CAM_CFG5") == "1"   # BASELINE.json configs[4]'s KLT stage: 4 cameras 1920 x 1080 x 5000 slots
if CFG5:
    W, H, L, FW, FH, NF = 1920, 1080, 4, 100, 50, 4


def cfg():
    return coslam_amd.KLT_SequenceTrackerConfig(nIterations=10, nLevels=L, levelSkip=1, windowWidth=7, trackWithGain=1,
                                                minCornerness=3000.0, convergenceThreshold=1.0, SSD_Threshold=20000.0,
                                                minDistance=8 if CFG5 else 4)


_frames = {}


def frames_of(cam):
    if cam not in _frames:
        sc = Scene(8, W, H, 12000 if CFG5 else 7000, seed=0xC051A + (5 if CFG5 else 4), sigma=1.0)
        _frames[cam] = torch.from_numpy(np.stack([sc.render(cam % 8, f) for f in range(NF)])).to(dev)   # (cameras >= 8 reuse views)
    return _frames[cam]


order = list(range(NF)) + list(range(NF - 2, 0, -1))


def run(n_cams, n_frames=80, prefetch=True, fused=1, probe=False, profile=True):
    stream = torch.cuda.Stream(device=dev)
    ts = []
    for c in range(n_cams):
        t = coslam_amd.KLT_SequenceTracker(cfg(), 0)
        t.allocate(W, H, L, FW, FH)
        t.set_fused(fused)
        t.set_xcd_placement(os.environ.get("KLT_XCD", "1") != "0")   # (the default placement: a camera per XCD; KLT_XCD=0: cameras as grid rows)
        ts.append(t)
    grp = coslam_amd.KLT_TrackerGroup(ts)
    grp.set_stream(stream.cuda_stream)
    fr = [frames_of(c) for c in range(n_cams)]
    dests = [torch.zeros(FW * FH * 5, dtype=torch.int32, device=dev) for _ in range(n_cams)]
    cnts = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(n_cams)]
    dp, cp = [d.data_ptr() for d in dests], [c.data_ptr() for c in cnts]
    grp.detect_dev([f[0].data_ptr() for f in fr], dp, cp)
    grp.advanceFrame()
    if probe:
        for t in ts:
            t.debug_probe(True)

    def frame(i):
        a, b = order[(i + 1) % len(order)], order[(i + 2) % len(order)]
        if prefetch:
            grp.prefetch_dev([f[b].data_ptr() for f in fr])
        grp.redetect_dev([f[a].data_ptr() for f in fr], dp, cp)
        grp.advanceFrame()

    for i in range(10):
        frame(i)
    grp.synchronize()
    if profile:
        ts[0].set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_frames):
        frame(10 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ts[0].get_profile() if profile else None
    grp.synchronize()
    live = [int((d.cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum()) for d in dests]
    pr = [t.debug_probe(False, read=True) for t in ts] if probe else None
    grp.close()
    for t in ts:
        t.close()
    trk_us = prof["tracker_us_total"] / max(prof["frames"], 1) if prof else float("nan")
    print(f"{n_cams} cameras, fused={fused}{' + prefetch' if prefetch else ''}: {n_frames / dt:8.0f} frames/s = "
          f"{n_cams * n_frames / dt:8.0f} camera-frames/s ({1e6 * dt / n_frames:7.1f} us per frame), tracker stage "
          f"{trk_us:7.1f} us, live {min(live)}..{max(live)}", flush=True)
    return pr


def probe_report(n):
    prs = run(n, probe=True, n_frames=30)
    nw = (FW * FH + 7) // 8
    prs = [p.astype(np.float64)[:nw] for p in prs]
    pr = prs[0]
    print(f"  per-wave cycles of camera 0 (last frame, {n} cameras), mean / p10 / p90 over {len(pr)} waves; 40 passes")
    for i, nm in enumerate(["sampling", "folds+solve prep", "hand-off wait", "finish+publish", "polls", "total"]):
        v = pr[:, i]
        print(f"    {nm:18s} {v.mean():10.0f} {np.percentile(v, 10):10.0f} {np.percentile(v, 90):10.0f}")
    print(f"    patch loads        {pr[:, 7].mean():10.1f}")
    t0 = min(p[:, 6].min() for p in prs)
    for c, p in enumerate(prs):
        st, en = p[:, 6] - t0, p[:, 6] + p[:, 5] - t0
        print(f"    camera {c}: waves start {st.min():9.0f}..{st.max():9.0f}, end {en.min():9.0f}..{en.max():9.0f} (s_memtime ticks)")


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    if CFG5:
        for n in (1, 2, 4):
            run(n, n_frames=40)
        run(4, n_frames=10, fused=0)
        sys.exit(0)
    for n in ((1, 8) if quick else (1, 2, 4, 8, 12, 16)):   # beyond the co-residency capacity the span is tracked in two launches
        run(n)
    if not quick:
        run(8, prefetch=False)
        run(8, fused=0, n_frames=20)
        run(1, fused=0, n_frames=20)
    for n in (1, 8):
        probe_report(n)
