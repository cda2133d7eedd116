#!/usr/bin/env python
"""Diagnostic: does work enqueued on stream B after a long chain of small kernels on stream A overlap with it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd
from coslam_amd.ba import BAWorkspace
dev = torch.device("cuda:0")
sc, frames, Ms, ms, R0, t0, ba = bench.build_inputs(0, 1, 0xC051A + 2)
P = len(ba["pts0"]); obs_pt = np.asarray(ba["obs_pt"]); o = np.argsort(obs_pt, kind="stable")
ptr = np.zeros(P + 1, np.int32); np.add.at(ptr, obs_pt + 1, 1); ptr = np.cumsum(ptr).astype(np.int32)
ws = BAWorkspace(0); ws.upload(ba["Ks"], ba["Rs0"], ba["ts0"], ba["pts0"], ptr, ba["obs_cam"][o], ba["obs_xy"][o])
d_baR = torch.from_numpy(ba["Rs0"].reshape(-1).copy()).to(dev); d_baT = torch.from_numpy(ba["ts0"].reshape(-1).copy()).to(dev)
d_baM = torch.from_numpy(ba["pts0"].reshape(-1).copy()).to(dev)
A = torch.cuda.Stream(device=dev); B = torch.cuda.Stream(device=dev)
x = torch.zeros(1 << 16, device=dev)
order = bench.frame_order(bench.N_FRAMES)
d_frames = torch.from_numpy(frames).to(dev)
d_dest = torch.zeros(2000 * 5, dtype=torch.int32, device=dev); d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
trk = coslam_amd.KLT_SequenceTracker(bench.klt_config(), 0)
trk.allocate(640, 480, 4, 50, 40); trk.set_stream(B.cuda_stream)
trk.detect_dev(d_frames[0].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
torch.cuda.synchronize()

def chainA(kind):
    if kind == "ba":
        ws.solve_dev(A.cuda_stream, d_baR.data_ptr(), d_baT.data_ptr(), d_baM.data_ptr(), 2, 2, 6.0, 2, 10)
    else:
        with torch.cuda.stream(A):
            for _ in range(150):
                x.add_(1.0)

def workB(kind, n):
    if kind == "klt":
        for i in range(n):
            trk.redetect_dev(d_frames[order[(i + 1) % len(order)]].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
    else:
        with torch.cuda.stream(B):
            for _ in range(n * 5):
                x.mul_(1.0)

for ka in ("ba", "torch"):
    for kb in ("klt", "torch"):
        for rep in range(3):
            torch.cuda.synchronize()
            eA0, eA1, eB0, eB1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            eA0.record(A); chainA(ka); eA1.record(A)
            eB0.record(B); workB(kb, 3); eB1.record(B)
            torch.cuda.synchronize()
            print(f"A={ka:5s} B={kb:5s}: A took {eA0.elapsed_time(eA1)*1e3:7.1f} us; B started {eA0.elapsed_time(eB0)*1e3:7.1f} us after A's start, "
                  f"B took {eB0.elapsed_time(eB1)*1e3:7.1f} us")
