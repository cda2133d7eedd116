#!/usr/bin/env python
"""Driver for rocprofv3 passes over the BA kernels: the headline's joint local BA (C = 40, order 144) and inter-camera
solve (order 48), each solved a few times through cs_ba_solve_dev (whole schedule enqueued, so every kernel shows), and --
PMC_CFG5=1 -- the cfg5-shaped problem (120 cameras, order 720)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd
from coslam_amd.synth import make_ba_problem

dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
coslam_amd.debug_set("ba_graphs", 0)   # eager launches: per-kernel records keep their names under the profiler


def solve(pr, ncon, npcon, maxIter, inner, reps):
    ptr, cam, xy = bench.csr(pr)
    ws = coslam_amd.BAWorkspace(0)
    ws.upload(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy)
    d = [torch.from_numpy(pr[k].reshape(-1).copy()).to(dev) for k in ("Rs0", "ts0", "pts0")]
    for _ in range(reps):
        ws.solve_dev(s, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), ncon, npcon, 6.0, maxIter, inner)
    torch.cuda.synchronize()
    _, _, _, _, st = ws.download()
    print(f"C={len(pr['Rs0'])} P={len(pr['pts0'])} obs={len(pr['obs_cam'])}: {st.nIterTotal} LM steps, cost {st.cost0:.1f} -> {st.cost:.1f}")
    ws.close()


if os.environ.get("PMC_CFG5"):
    pr = make_ba_problem(n_cams=120, n_pts=5000, visibility=1.0, seed=12, W=1920, H=1080, n_cams_con=8, n_pts_con=2)
    solve(pr, 8, 2, 1, 5, 2)
else:
    sc = bench.build_scene()
    joint, ic = bench.build_ba_problems(sc)
    solve(joint, joint["n_cams_con"], joint["n_pts_con"], 2, 10, 3)
    solve(ic, 0, ic["n_static"], 3, 40, 3)
