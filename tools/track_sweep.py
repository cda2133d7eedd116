#!/usr/bin/env python
"""Diagnostic: tracker-stage time (HIP events), bit-identity of the persistent gain tracker's hand-off variants and
the per-wave cycle breakdown of a pass (texel wait / arithmetic / hand-off wait / solve + publish)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd

dev = torch.device("cuda:0")
sc, frames, *_ = bench.build_inputs(0, 1, 0xC051A + 2)
order = bench.frame_order(bench.N_FRAMES)
d_frames = torch.from_numpy(frames).to(dev)
d_dest = torch.zeros(2000 * 5, dtype=torch.int32, device=dev); d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream

def canon(dest_words):
    a = dest_words.cpu().numpy().view(coslam_amd.KLT_TrackedFeature).copy()
    dead = a["status"] < 0
    a["pos"][dead] = 0; a["gain"][dead] = 0
    return a.tobytes()

def run(fused=1, n=60, probe=False):
    trk = coslam_amd.KLT_SequenceTracker(bench.klt_config(), 0)
    trk.allocate(640, 480, 4, 50, 40); trk.set_stream(stream); trk.set_fused(fused)
    trk.detect_dev(d_frames[0].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
    trk.set_profiling(True)
    if probe: trk.debug_probe(True)
    h = hashlib.sha1()
    for i in range(n):
        trk.redetect_dev(d_frames[order[(i + 1) % len(order)]].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr())
        trk.advanceFrame()
        if i < 12:
            trk.synchronize()
            h.update(canon(d_dest))
    prof = trk.get_profile()
    trk.synchronize()
    pr = trk.debug_probe(False, read=True) if probe else None
    trk.close()
    return prof["tracker_us_total"] / prof["frames"], h.hexdigest()[:12], pr

for fused in (1, 1, 0):
    us, hh, _ = run(fused)
    print(f"fused={fused}: tracker {us:7.1f} us/frame  dest-hash {hh}", flush=True)
us, hh, pr = run(1, probe=True)
print(f"probe run: tracker {us:.1f} us/frame  hash {hh}")
pr = pr.astype(np.float64)
tot = pr[:, 5]
print("per-wave cycles (last frame), mean / p10 / p90 over 2000 waves; 40 passes")
pr[:, 6] = pr[:, 7]
for i, nm in enumerate(["texel wait", "arithmetic", "hand-off wait", "solve+publish", "polls", "total", "patch loads"]):
    v = pr[:, i]
    print(f"  {nm:14s} {v.mean():10.0f} {np.percentile(v,10):10.0f} {np.percentile(v,90):10.0f}")
