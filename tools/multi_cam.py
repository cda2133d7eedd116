#!/usr/bin/env python
"""Diagnostic: KLT throughput with several cameras on ONE MI355X (cfg3 / the 8-camera case on one GPU): camera-frames/s
of the redetect loop with the cameras spread over `S` streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd
dev = torch.device("cuda:0")
order = bench.frame_order(bench.N_FRAMES)
def run(n_cams, n_streams, frames_per_cam=60, prefetch=False):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    cams = []
    for c in range(n_cams):
        sc, frames, *_ = bench.build_inputs(c % 3, 3, 0xC051A + 3)
        t = coslam_amd.KLT_SequenceTracker(bench.klt_config(), 0)
        t.allocate(640, 480, 4, 50, 40)
        t.set_stream(streams[c % n_streams].cuda_stream)
        t.set_concurrent_handles(n_streams)
        d_frames = torch.from_numpy(frames).to(dev)
        dest = torch.zeros(2000 * 5, dtype=torch.int32, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        t.detect_dev(d_frames[0].data_ptr(), dest.data_ptr(), cnt.data_ptr()); t.advanceFrame()
        cams.append((t, d_frames, dest, cnt))
    def frame(i):
        for (t, d_frames, dest, cnt) in cams:
            if prefetch: t.prefetch_dev(d_frames[order[(i + 2) % len(order)]].data_ptr())
            t.redetect_dev(d_frames[order[(i + 1) % len(order)]].data_ptr(), dest.data_ptr(), cnt.data_ptr()); t.advanceFrame()
    for i in range(10): frame(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(frames_per_cam): frame(10 + i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    live = [int((c[2].cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum()) for c in cams]
    for c in cams: c[0].synchronize(); c[0].close()
    print(f"{n_cams} cameras on {n_streams} stream(s){' + prefetch' if prefetch else ''}: {n_cams * frames_per_cam / dt:8.0f} camera-frames/s ({1e6 * dt / frames_per_cam:7.1f} us per frame set), live {min(live)}..{max(live)}", flush=True)
for n_cams, n_streams in ((1, 1), (2, 2), (3, 3), (4, 2), (6, 3), (8, 2), (8, 3), (8, 4), (8, 8)):
    run(n_cams, n_streams)
for n_cams, n_streams in ((1, 1), (2, 2), (3, 3), (8, 2), (8, 3)):
    run(n_cams, n_streams, prefetch=True)
