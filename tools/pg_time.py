#!/usr/bin/env python
"""Stand-alone timing of the pose-graph relaxation (cs_posegraph_relax_dev + edges) at the benchmark's shape (8 cameras, key
frame every 5th frame, 5 key frames in the window) and at longer key-frame intervals; the oracle's dense QR on one host core
beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coslam_amd, oracle
from coslam_amd.synth import make_pose_graphs

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for kw in (dict(n_cams=8, n_frames=21, key_every=5), dict(n_cams=8, n_frames=121, key_every=30)):
    pg = make_pose_graphs(seed=1, **kw)
    h = coslam_amd.PoseGraphs(pg["graphs"])
    d = {k: T(pg[k]) for k in ("nodeR0", "nodeT0", "nodeR", "nodeT", "edgeR", "edgeT")}
    nR, nT = torch.zeros_like(d["nodeR"]), torch.zeros_like(d["nodeT"])
    s = torch.cuda.current_stream().cuda_stream
    def run():
        h.edges_dev(s, d["nodeR0"].data_ptr(), d["nodeT0"].data_ptr(), d["edgeR"].data_ptr(), d["edgeT"].data_ptr())
        h.relax_dev(s, d["nodeR"].data_ptr(), d["nodeT"].data_ptr(), d["edgeR"].data_ptr(), d["edgeT"].data_ptr(), nR.data_ptr(), nT.data_ptr())
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    h.status(s)
    t0 = time.perf_counter()
    for g, (fixed, id1, id2) in enumerate(pg["graphs"]):
        ns, es = slice(pg["node_ptr"][g], pg["node_ptr"][g + 1]), slice(pg["edge_ptr"][g], pg["edge_ptr"][g + 1])
        oracle.posegraph_relax(fixed, pg["nodeR"][ns], pg["nodeT"][ns], id1, id2, pg["edgeR"][es], pg["edgeT"][es])
        if kw["n_frames"] > 200: break
    cpu = (time.perf_counter() - t0) * (1 if kw["n_frames"] <= 200 else kw["n_cams"])
    print(f"{kw}: {h.counts()}  edges + relaxation {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per call (back to back); "
          f"oracle dense QR, 1 host core: {cpu * 1e3:.1f} ms")
