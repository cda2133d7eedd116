#!/usr/bin/env python
"""Small driver for rocprofv3 --pmc: runs the KLT redetect loop only (the dominant kernel is k_track_gain_fused)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, coslam_amd
dev = torch.device("cuda:0")
sc, frames, *_ = bench.build_inputs(0, 1, 0xC051A + 2)
order = bench.frame_order(bench.N_FRAMES)
d_frames = torch.from_numpy(frames).to(dev)
d_dest = torch.zeros(2000 * 5, dtype=torch.int32, device=dev); d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
trk = coslam_amd.KLT_SequenceTracker(bench.klt_config(), 0)
trk.allocate(640, 480, 4, 50, 40); trk.set_stream(torch.cuda.current_stream().cuda_stream)
trk.detect_dev(d_frames[0].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
for i in range(40):
    trk.redetect_dev(d_frames[order[(i + 1) % len(order)]].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
torch.cuda.synchronize()
