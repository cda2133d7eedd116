#!/usr/bin/env python
"""SURVEY 8(d) secondary row: the KLT stage with the reference's own parameter set (CoSLAM's overrides of
KLT_SequenceTrackerConfig, src/app/SL_SingleSLAM.cpp:291-298, SL_GlobParam.cpp:28-34: nLevels 6, levelSkip 2,
windowWidth 6, 12 iterations, minDistance 8, convergence 1.0, SSD 20000, with gain) next to the bench's cfg2 set."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd

dev = torch.device("cuda:0")
sc, frames, *_ = bench.build_inputs(0, 1, 0xC051A + 2)
order = bench.frame_order(bench.N_FRAMES)
d_frames = torch.from_numpy(frames).to(dev)
ptrs = [d_frames[f].data_ptr() for f in range(bench.N_FRAMES)]
stream = torch.cuda.current_stream().cuda_stream
ref = coslam_amd.KLT_SequenceTrackerConfig(nIterations=12, nLevels=6, levelSkip=2, windowWidth=6, trackWithGain=1,
                                           minCornerness=3000.0, convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=8)
for name, cfg, L in (("cfg2 (4 levels, skip 1, 10 it, 7x7, minDistance 5)", bench.klt_config(), 4),
                     ("reference defaults (6 levels, skip 2, 12 it, windowWidth 6, minDistance 8)", ref, 6)):
    for pf in (0, 1):
        d_dest = torch.zeros(2000 * 5, dtype=torch.int32, device=dev); d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
        trk = coslam_amd.KLT_SequenceTracker(cfg, 0)
        trk.allocate(640, 480, L, 50, 40); trk.set_stream(stream)
        trk.detect_dev(ptrs[order[0]], d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
        def step(i):
            if pf: trk.prefetch_dev(ptrs[order[(i + 2) % len(order)]])
            trk.redetect_dev(ptrs[order[(i + 1) % len(order)]], d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
        for i in range(30): step(i)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(30, 330): step(i)
        torch.cuda.synchronize(); us = (time.perf_counter() - t) / 300 * 1e6
        trk.set_profiling(True)
        for i in range(330, 370): step(i)
        prof = trk.get_profile(); trk.synchronize(); trk.set_profiling(False)
        live = int((d_dest.cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum())
        print(f"{name}, prefetch={pf}: {us:6.1f} us/frame ({1e6 / us:6.0f} frames/s), tracker kernel {prof['tracker_us_total'] / prof['frames']:5.1f} us, "
              f"{prof.get('launches_per_frame', 1)} launch(es), live {live}")
        trk.close()
