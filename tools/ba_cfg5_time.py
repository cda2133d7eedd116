#!/usr/bin/env python
"""BASELINE cfg5's bundle adjustment (120 poses of which 8 fixed, 5000 points, every point in every key frame: 600 k
measurements, reduced system of order 672) stand-alone: microseconds per LM step with the Schur sum on the f64 matrix cores
(default) and with the pair-per-workgroup kernel (cs_debug_set ba_syrk = 0).  Under rocprofv3 --kernel-trace --stats it gives the
per-kernel table of profiles/r02_ba_cfg5_kernel_stats.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coslam_amd, oracle
from coslam_amd.synth import make_ba_problem

dev = torch.device("cuda:0")
pr = make_ba_problem(n_cams=120, n_pts=5000, W=1920, H=1080, noise=0.3, outlier_frac=0.01, outlier_mag=40.0, n_cams_con=8, n_pts_con=2, seed=55)
ptr, cam, xy, _ = oracle.csr_by_point(len(pr["pts0"]), pr["obs_pt"], pr["obs_cam"], pr["obs_xy"])
modes = sys.argv[1:] or ["1", "0"]
for mode in modes:
    coslam_amd.debug_set("ba_syrk", int(mode) if mode != "1" else -1)
    ws = coslam_amd.BAWorkspace(0)
    ws.upload(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy)
    d = [torch.from_numpy(pr[k].reshape(-1).copy()).to(dev) for k in ("Rs0", "ts0", "pts0")]
    s = torch.cuda.current_stream().cuda_stream
    ws.solve_dev(s, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), 8, 2, 6.0, 2, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ws.solve_dev(s, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), 8, 2, 6.0, 2, 5)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 3 * 1e6
    R, T, M, out, st = ws.download()
    print(f"ba_syrk={mode}: {us:.0f} us per solve, {st.nIterTotal} LM steps -> {us / max(st.nIterTotal, 1):.0f} us per step; "
          f"cost {st.cost0:.1f} -> {st.cost:.3f}, {int(out.sum())} outliers; |R| checksum {np.abs(R).sum():.12f}")
    ws.close()
