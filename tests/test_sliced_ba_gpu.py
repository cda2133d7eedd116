"""Collective 2 on one MI355X: the rank-local HIP phases (cs_ba_dist_*) driven by the same schedule as
the CPU tests -- as a single rank, and as two / three emulated ranks whose buffers are summed on the device in place of
the RCCL all-reduce -- must reproduce the oracle and the single-process solver."""
import numpy as np
import pytest
import torch

import coslam_amd
import oracle
from coslam_amd.ba import BAWorkspace
from coslam_amd.multicam import HipSlicedBA, bundle_adjust_sliced, point_slice, run_sliced_ba
from coslam_amd.synth import make_ba_problem

pytestmark = pytest.mark.gpu


def setup(kw):
    pr = make_ba_problem(**kw)
    P = len(pr["pts0"])
    ptr, cam, xy, _ = oracle.csr_by_point(P, pr["obs_pt"], pr["obs_cam"], pr["obs_xy"])
    return pr, ptr, cam, xy


def compare(Rs, Ts, pts, out, st, ref):
    R_o, T_o, M_o, out_o, st_o = ref
    assert np.array_equal(out, out_o)
    assert st.nOuter == st_o.nOuter
    sane = np.linalg.norm(M_o, axis=1) < 1e3
    scale = max(1.0, np.abs(M_o[sane]).max())
    assert np.max(np.abs(Rs.reshape(-1, 9) - R_o.reshape(-1, 9))) < 1e-6
    assert np.max(np.abs(Ts - T_o)) < 1e-6 * scale
    assert np.max(np.abs(pts[sane] - M_o[sane])) < 1e-6 * scale
    assert abs(st.cost - st_o.cost) <= 1e-7 * max(1.0, st_o.cost)


CASES = [
    (dict(n_cams=5, n_pts=300, n_cams_con=2, n_pts_con=2, seed=31), 2, 2, 2, 10),
    (dict(n_cams=8, n_pts=400, n_cams_con=0, n_pts_con=340, seed=32, visibility=0.9), 0, 340, 3, 12),   # cfg4 inter-camera solve
    (dict(n_cams=24, n_pts=500, n_cams_con=4, n_pts_con=2, seed=33, visibility=0.5), 4, 2, 2, 6),        # order 120
]


@pytest.mark.parametrize("kw,ncon,npcon,maxIter,inner", CASES)
@pytest.mark.parametrize("world", [1, 2, 3])
def test_sliced_ba_on_device_matches_oracle(hip, kw, ncon, npcon, maxIter, inner, world):
    pr, ptr, cam, xy = setup(kw)
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    d_R = torch.from_numpy(pr["Rs0"].reshape(-1).copy()).to(dev)
    d_T = torch.from_numpy(pr["ts0"].reshape(-1).copy()).to(dev)
    d_M = torch.from_numpy(pr["pts0"].reshape(-1).copy()).to(dev)
    torch.cuda.synchronize()
    wss, engs = [], []
    for r in range(world):
        ws = BAWorkspace(0)
        ws.upload(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy)
        lo, hi = point_slice(r, world, ws.P)
        engs.append(HipSlicedBA(ws, stream, d_R.data_ptr(), d_T.data_ptr(), d_M.data_ptr(), ncon, npcon, 6.0, inner, lo, hi,
                                r == 0, 0))
        wss.append(ws)

    def reduce_fn(name):
        if world == 1:
            return
        with torch.cuda.stream(stream):
            tot = getattr(engs[0], name).clone()
            for e in engs[1:]:
                tot += getattr(e, name)
            for e in engs:
                getattr(e, name).copy_(tot)

    run_sliced_ba(engs, reduce_fn, maxIter, inner)
    torch.cuda.synchronize()
    ref = oracle.ba_robust(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy, ncon, npcon, 6.0, maxIter, inner)
    for ws in wss:
        Rs, Ts, pts, out, st = ws.download()
        compare(Rs, Ts, pts, out, st, ref)
        ws.close()


def test_single_rank_entry_point_equals_the_monolithic_solver(hip):
    pr, ptr, cam, xy = setup(CASES[0][0])
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    d_R = torch.from_numpy(pr["Rs0"].reshape(-1).copy()).to(dev)
    d_T = torch.from_numpy(pr["ts0"].reshape(-1).copy()).to(dev)
    d_M = torch.from_numpy(pr["pts0"].reshape(-1).copy()).to(dev)
    ws = BAWorkspace(0)
    ws.upload(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy)
    torch.cuda.synchronize()
    bundle_adjust_sliced(ws, stream, d_R.data_ptr(), d_T.data_ptr(), d_M.data_ptr(), 2, 2, 6.0, 2, 10, 0)
    torch.cuda.synchronize()
    Rs, Ts, pts, out, st = ws.download()
    Rm, Tm, Mm = pr["Rs0"].copy(), pr["ts0"].copy(), pr["pts0"].copy()
    out_m, st_m = coslam_amd.bundleAdjustRobust(2, pr["Ks"], Rm, Tm, 2, Mm, (ptr, cam, xy), 6.0, 2, 10)
    assert np.array_equal(out, out_m) and st.nIterTotal == st_m.nIterTotal
    assert np.max(np.abs(Rs.reshape(-1, 9) - Rm.reshape(-1, 9))) < 1e-9 and np.max(np.abs(pts - Mm)) < 1e-8
    ws.close()
