#!/usr/bin/env python
"""Stand-alone timing of the headline's two key-frame solves (joint local BA, inter-camera solve): whole solve through
cs_ba_solve_dev (one graph) and per-kernel averages with eager launches under HIP events are left to rocprofv3; this
prints microseconds per solve and per LM step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd

dev = torch.device("cuda:0")
ts_ = torch.cuda.Stream(device=dev)
s = ts_.cuda_stream
sc = bench.build_scene()
joint, ic = bench.build_ba_problems(sc)


def run(pr, ncon, npcon, maxIter, inner, name):
    ptr, cam, xy = bench.csr(pr)
    ws = coslam_amd.BAWorkspace(0)
    ws.upload(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy)
    d = [torch.from_numpy(pr[k].reshape(-1).copy()).to(dev) for k in ("Rs0", "ts0", "pts0")]
    for _ in range(3):
        ws.solve_dev(s, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), ncon, npcon, 6.0, maxIter, inner)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ws.solve_dev(s, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), ncon, npcon, 6.0, maxIter, inner)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    _, _, _, _, st = ws.download()
    print(f"{name}: {us:.0f} us per solve, {st.nIterTotal} LM steps -> {us / max(st.nIterTotal, 1):.1f} us per step; cost {st.cost0:.1f} -> {st.cost:.3f}")
    ws.close()


run(joint, joint["n_cams_con"], joint["n_pts_con"], 2, 10, "joint local BA (order 144)")
run(ic, 0, ic["n_static"], 3, 40, "inter-camera solve (order 48)")
