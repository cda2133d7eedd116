#!/usr/bin/env python
"""KLT-only frame time with and without cs_klt_prefetch_dev (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd

dev = torch.device("cuda:0")
sc, frames, Ms, ms, R0, t0, ba = bench.build_inputs(0, 1, 0xC051A + 2)
order = bench.frame_order(bench.N_FRAMES)
d_frames = torch.from_numpy(frames).to(dev)
ptrs = [d_frames[f].data_ptr() for f in range(bench.N_FRAMES)]
d_dest = torch.zeros(2000 * 5, dtype=torch.int32, device=dev); d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for pf in (0, 1, 0, 1):
    trk = coslam_amd.KLT_SequenceTracker(bench.klt_config(), 0)
    trk.allocate(640, 480, 4, 50, 40); trk.set_stream(stream)
    trk.detect_dev(ptrs[order[0]], d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
    def step(i):
        if pf: trk.prefetch_dev(ptrs[order[(i + 2) % len(order)]])
        trk.redetect_dev(ptrs[order[(i + 1) % len(order)]], d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
    for i in range(30): step(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(30, 330): step(i)
    torch.cuda.synchronize()
    print(f"prefetch={pf}: {(time.perf_counter() - t) / 300 * 1e6:.1f} us/frame, counts {d_counts.cpu().numpy()}")
    trk.close()
