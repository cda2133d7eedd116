#!/usr/bin/env python
"""Per-component GPU time of the bench step (diagnostic, not the benchmark)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd
from coslam_amd.ba import BAWorkspace
from coslam_amd.pose import IntraCamPoseOption, intraCamEstimate_batch_dev

dev = torch.device("cuda:0")
sc, frames, Ms, ms, R0, t0, ba = bench.build_inputs(0, 1, 0xC051A + 2)
order = bench.frame_order(bench.N_FRAMES)
d_frames = torch.from_numpy(frames).to(dev)
d_K = torch.from_numpy(sc.K.ravel().copy()).to(dev)
d_Ms, d_ms = torch.from_numpy(Ms).to(dev), torch.from_numpy(ms).to(dev)
d_R0, d_t0 = torch.from_numpy(R0).to(dev), torch.from_numpy(t0).to(dev)
d_npts = torch.full((1,), 192, dtype=torch.int32, device=dev)
d_Ropt = torch.zeros(9, dtype=torch.float64, device=dev); d_topt = torch.zeros(3, dtype=torch.float64, device=dev)
opt0 = IntraCamPoseOption()
d_opt0 = torch.from_numpy(np.frombuffer(bytes(opt0), dtype=np.uint8).copy()).to(dev); d_opt = torch.zeros_like(d_opt0)
d_ok = torch.zeros(1, dtype=torch.int32, device=dev)
d_dest = torch.zeros(2000 * 5, dtype=torch.int32, device=dev); d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream

def timeit(fn, n=300, warm=30):
    for i in range(warm): fn(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(warm + i)
    t_host = time.perf_counter() - t
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6, t_host / n * 1e6

for fused in (1, 0):
    for graphs in (1, 0):
        trk = coslam_amd.KLT_SequenceTracker(bench.klt_config(), 0)
        trk.allocate(640, 480, 4, 50, 40); trk.set_stream(stream); trk.set_fused(fused); trk.enable_graphs(graphs)
        trk.detect_dev(d_frames[0].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
        ptrs = [d_frames[f].data_ptr() for f in range(bench.N_FRAMES)]
        dp, cp = d_dest.data_ptr(), d_counts.data_ptr()
        def klt(i):
            trk.redetect_dev(ptrs[order[(i + 1) % len(order)]], dp, cp); trk.advanceFrame()
        print(f"KLT redetect fused={fused} graphs={graphs}: %.1f us/frame (host enqueue %.1f us)" % timeit(klt))
        trk.close()

pK, pn = d_K.data_ptr(), d_npts.data_ptr()
def pose(i):
    f = order[i % len(order)]
    d_opt.copy_(d_opt0, non_blocking=True)
    intraCamEstimate_batch_dev(stream, 1, 192, pK, d_R0[f].data_ptr(), d_t0[f].data_ptr(), pn, 0, d_Ms[f].data_ptr(),
                               d_ms[f].data_ptr(), 10.0, d_Ropt.data_ptr(), d_topt.data_ptr(), d_opt.data_ptr(), d_ok.data_ptr())
print("pose: %.1f us/call (host %.1f us)" % timeit(pose))

P = len(ba["pts0"]); obs_pt = np.asarray(ba["obs_pt"]); o = np.argsort(obs_pt, kind="stable")
ptr = np.zeros(P + 1, np.int32); np.add.at(ptr, obs_pt + 1, 1); ptr = np.cumsum(ptr).astype(np.int32)
ws = BAWorkspace(0); ws.upload(ba["Ks"], ba["Rs0"], ba["ts0"], ba["pts0"], ptr, ba["obs_cam"][o], ba["obs_xy"][o])
d_baR = torch.from_numpy(ba["Rs0"].reshape(-1).copy()).to(dev); d_baT = torch.from_numpy(ba["ts0"].reshape(-1).copy()).to(dev)
d_baM = torch.from_numpy(ba["pts0"].reshape(-1).copy()).to(dev)
def bafn(i):
    ws.solve_dev(stream, d_baR.data_ptr(), d_baT.data_ptr(), d_baM.data_ptr(), 2, 2, 6.0, 2, 10)
print("BA (2,10): %.1f us/call (host %.1f us)" % timeit(bafn, 50, 5))
print(ws.download()[4].nIterTotal, "LM iterations")

# PCIe-inclusive: the reference-shaped host entry point (image upload + dest read-back + sync every frame)
trk = coslam_amd.KLT_SequenceTracker(bench.klt_config(), 0)
trk.allocate(640, 480, 4, 50, 40)
trk.detect(frames[0]); trk.advanceFrame()
def host(i):
    trk.redetect(frames[order[(i + 1) % len(order)]]); trk.advanceFrame()
print("KLT redetect through the host-pointer API (H2D image + D2H dest + sync): %.1f us/frame (host %.1f)" % timeit(host, 200, 20))

# the reference-shaped host entry points of the pose solve and the BA (host arrays in, host arrays out, synchronous)
from coslam_amd.pose import intraCamEstimate
def pose_host(i):
    f = order[i % len(order)]
    intraCamEstimate(sc.K, R0[f].reshape(3, 3), t0[f], 192, None, Ms[f], ms[f], 10.0)
print("intraCamEstimate through the host-pointer API: %.1f us/call (host %.1f)" % timeit(pose_host, 200, 20))
