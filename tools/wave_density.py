#!/usr/bin/env python
"""Diagnostic: persistent tracker time vs the number of feature slots (= waves) on the whole chip and on a 192-CU
partition -- how much of the pass time is the number of co-resident waves per SIMD."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd

dev = torch.device("cuda:0")
sc, frames, *_ = bench.build_inputs(0, 1, 0xC051A + 2)
order = bench.frame_order(bench.N_FRAMES)
d_frames = torch.from_numpy(frames).to(dev)
L = coslam_amd.lib(); L.cs_stream_create_cu_range.restype = C.c_void_p
streams = {"256 CUs": (torch.cuda.current_stream().cuda_stream, 256), "192 CUs": (L.cs_stream_create_cu_range(0, 0, 192), 192)}
for name, (stream, ncu) in streams.items():
    for fw, fh in ((50, 40), (40, 38), (40, 25), (32, 24), (25, 20), (16, 16)):
        N = fw * fh
        d_dest = torch.zeros(N * 5, dtype=torch.int32, device=dev); d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
        cfg = bench.klt_config(); cfg.minCornerness = 500.0   # enough corners to fill every grid
        trk = coslam_amd.KLT_SequenceTracker(cfg, 0)
        trk.allocate(640, 480, 4, fw, fh); trk.set_stream(stream); trk.set_cu_count(ncu)
        trk.detect_dev(d_frames[0].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
        trk.set_profiling(True)
        for i in range(60):
            trk.redetect_dev(d_frames[order[(i + 1) % len(order)]].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr())
            trk.advanceFrame()
        prof = trk.get_profile(); trk.synchronize()
        live = int((d_dest.cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum())
        waves_per_simd = N / (ncu * 4)
        print(f"{name}: {N:5d} slots ({live} live) = {waves_per_simd:.2f} waves/SIMD: tracker {prof['tracker_us_total'] / prof['frames']:6.1f} us")
        trk.close()
