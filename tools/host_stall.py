#!/usr/bin/env python
"""Diagnostic: host time of ba_ws.solve_dev / redetect_dev calls in the bench loop."""
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
s = torch.cuda.Stream(device=dev)
for rep in range(6):
    torch.cuda.synchronize()
    t = time.perf_counter()
    ws.solve_dev(s.cuda_stream, d_baR.data_ptr(), d_baT.data_ptr(), d_baM.data_ptr(), 2, 2, 6.0, 2, 10)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"graphs={os.environ.get('COSLAM_BA_GRAPHS','1')} call {rep}: host {1e6*(t1-t):.1f} us, total {1e6*(t2-t):.1f} us")
