"""InterCamPoseEstimator::addMapPoints built on the device (cs_ba_solve_intercam_async) against what the reference's own function built
from the same records (tests/golden/intercam_golden.npz, tests/cxx/ref_intercam_test.cpp), and the solve behind it against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(sc, g, max_iter=0, inner=0):
    import torch

    from coslam_amd.ba import BAInterCam, BAWorkspace, intercam_cams
    from coslam_amd.multicam import _DevArray

    G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
    nc, N, nMap, frame, W, H, ncb, nrb, ps = (int(v) for v in G("dims"))
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    K = np.array([0.82 * W, 0, W / 2.0, 0, 0.82 * W, H / 2.0, 0, 0, 1.0])
    dK, dxy, dst, ds2m, dsp, dfs = d(K), d(G("xy")), d(G("state")), d(G("slot2map")), d(G("trackSpan")), d(G("isStatic"))
    dM, dfl, dnp, dpf, dR, dT = d(G("mapPts")), d(G("mapFlags")), d(G("newPt")), d(G("pointFeat")), d(G("curR")), d(G("curT"))
    cams = intercam_cams([dict(K=dK.data_ptr(), xy=dxy[c].data_ptr(), state=dst[c].data_ptr(), slot2map=ds2m[c].data_ptr(),
                               trackSpan=dsp[c].data_ptr(), isStatic=dfs[c].data_ptr()) for c in range(nc)])
    ic, ws = BAInterCam(nc, N, ps, nMap), BAWorkspace(0)
    s = torch.cuda.current_stream().cuda_stream
    ic.solve_async(ws, s, cams, W, H, ncb, nrb, dR.data_ptr(), dT.data_ptr(), dM.data_ptr(), dfl.data_ptr(), dnp.data_ptr(), dpf.data_ptr(),
                   6.0, max_iter, inner)
    ws.wait()
    C, P, nObs, nStatic, pm = ic.last_problem()
    ga = lambda p, n, ty: torch.as_tensor(_DevArray(p, n, ty), device=dev).cpu().numpy()   # noqa: E731
    _, optr, ocam, oxy = ws.problem_buffers()
    ws.set_sizes(C, P, nObs)
    Rs, Ts, pts, outl, st = ws.download()
    return dict(C=C, P=P, nObs=nObs, nStatic=nStatic, point_map=ga(pm, P, "<i4"), obs_ptr=ga(optr, P + 1, "<i4"), obs_cam=ga(ocam, nObs, "<i4"),
                obs_xy=ga(oxy, 2 * nObs, "<f8").reshape(nObs, 2), Rs=Rs, Ts=Ts, pts=pts, outlier=outl, stats=st, K=K.reshape(3, 3),
                keep=(ic, ws, dK, dxy, dst, ds2m, dsp, dfs, dM, dfl, dnp, dpf, dR, dT))


def test_device_built_problem_is_the_one_the_reference_builds(hip):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "intercam_golden.npz"))
    for sc in range(int(g["n_scenes"])):
        G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
        r = _build(sc, g)
        n_static, n_dyn, P, n_obs = (int(v) for v in G("counts"))
        assert (r["C"], r["P"], r["nObs"], r["nStatic"]) == (int(G("dims")[0]), P, n_obs, n_static)
        for k in ("obs_ptr", "obs_cam", "obs_xy", "point_map"):
            assert np.array_equal(r[k], G(k)), (sc, k)
        # zero LM steps: the estimate the solve returns is the problem as parsed -- the map's points, the cameras' current poses
        assert np.array_equal(r["pts"], G("pts")) and np.array_equal(r["Rs"].reshape(-1, 9), G("curR")) and np.array_equal(r["Ts"], G("curT"))


def _scene_with_pixels_on_the_projections():
    """golden scene 1 with the mapped features moved onto the projections of their map points under slightly different poses (the golden
    scenes' pixels are unrelated to their map points: they pin the bookkeeping), so that a solve has something to find"""
    g0 = np.load(os.path.join(os.path.dirname(__file__), "golden", "intercam_golden.npz"))
    g = {k: g0[k].copy() for k in g0.files}
    sc = 1
    G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
    nc, N, nMap, frame, W, H, ncb, nrb, ps = (int(v) for v in G("dims"))
    K = np.array([[0.82 * W, 0, W / 2.0], [0, 0.82 * W, H / 2.0], [0, 0, 1.0]])
    rng = np.random.default_rng(4)
    from tests.poseupdate_scene import rodrigues

    for c in range(nc):
        Rt = rodrigues(rng.normal(0, 3e-3, 3)) @ G("curR")[c].reshape(3, 3)   # the pose the pixels were "taken" from
        tt = G("curT")[c] + rng.normal(0, 0.02, 3)
        for s_ in range(N):
            m = G("slot2map")[c][s_]
            if m < 0 or G("state")[c][s_] not in (0, 1):
                continue
            X = Rt @ G("mapPts")[m] + tt
            u = K @ X
            px = u[:2] / u[2] + rng.normal(0, 0.3, 2)
            if not (1 <= px[0] < W - 1 and 1 <= px[1] < H - 1):
                G("state")[c][s_] = -1        # (left the image: the track is gone)
                G("pointFeat")[m, c] = -1
                continue
            G("xy")[c][s_], G("xy")[c][N + s_] = px
    return g, sc


def test_the_solve_behind_the_device_build_matches_the_oracle(hip):
    """bundleAdjustRobust(0, Ks, Rs, Ts, m_numStatic, pts, meas, 6, 3, 40) (SL_InterCamPoseEstimator.cpp:95) on the device-built problem
    against the oracle's restatement of addMapPoints + the oracle's solver.  The golden scenes' pixels are unrelated to their map
    points (they pin the bookkeeping); here the mapped features of scene 1 are moved onto the projections of their map points
    under slightly different poses, so that the solve has something to find."""
    import oracle

    g, sc = _scene_with_pixels_on_the_projections()
    G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
    nc, N, nMap, frame, W, H, ncb, nrb, ps = (int(v) for v in G("dims"))
    want = oracle.intercam_add_map_points(W, H, ncb, nrb, ps, G("xy"), G("state"), G("slot2map"), G("trackSpan"), G("isStatic"), G("mapPts"),
                                          G("mapFlags"), G("newPt"), G("pointFeat"))
    r = _build(sc, g, 3, 40)
    assert r["nStatic"] == want["n_static"] > 100 and r["P"] == len(want["pts"]) > r["nStatic"]
    for k in ("obs_ptr", "obs_cam", "obs_xy", "point_map"):
        assert np.array_equal(r[k], want[k]), k
    Ks = np.repeat(r["K"][None], nc, 0)
    Ro, To, Mo, out_o, st_o = oracle.ba_robust(Ks, G("curR").reshape(nc, 3, 3), G("curT"), want["pts"], want["obs_ptr"], want["obs_cam"], want["obs_xy"],
                                               0, want["n_static"], 6.0, 3, 40)
    assert st_o.nIterTotal == r["stats"].nIterTotal > 3 and np.array_equal(out_o, r["outlier"]) and r["stats"].cost < 0.5 * r["stats"].cost0
    assert np.abs(r["Rs"] - Ro).max() < 1e-6 and np.abs(r["Ts"] - To).max() < 1e-5 and np.abs(r["pts"] - Mo).max() < 1e-5
    assert np.array_equal(r["pts"][:r["nStatic"]], want["pts"][:r["nStatic"]])   # the static points are held


def test_apply_writes_the_poses_back_and_gates_the_map_like_the_reference(hip):
    """cs_ba_intercam_apply_dev (VERDICT r04 missing 5): InterCamPoseEstimator::apply's write-back (reference
    src/app/SL_InterCamPoseEstimator.cpp:100-136) behind the solve of the scene above -- the solved poses ARE the current poses afterwards,
    and the map has gone through the gate of poseUpdate3D's tail under them (the loop at :105-136 is SL_SingleSLAM.cpp:677-706 statement for
    statement; the oracle's restatement of that loop is pinned to the reference's own): inliers re-triangulated, outliers uncertain, the
    features' reprojErr -- bit for bit against oracle.pose_update_gate run with the same poses, camera after camera."""
    import torch

    import oracle
    from coslam_amd.poseupdate import poseupdate_cams

    g, sc = _scene_with_pixels_on_the_projections()
    G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
    nc, N, nMap, frame, W, H, ncb, nrb, ps = (int(v) for v in G("dims"))
    r = _build(sc, g, 3, 40)
    ic, ws, dK, dxy, dst, ds2m, dsp, dfs, dM, dfl, dnp, dpf, dR, dT = r["keep"]
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(9)
    A = rng.normal(0, 0.03, (nMap, 3, 3))
    cov0 = (A @ A.transpose(0, 2, 1) + 1e-5 * np.eye(3)).reshape(nMap, 9)
    d_cov = torch.from_numpy(cov0.copy()).to(dev)
    d_err = torch.zeros((nc, N), dtype=torch.float64, device=dev)
    d_nodes, d_out = torch.zeros(nc, dtype=torch.int32, device=dev), torch.zeros(nc, dtype=torch.int32, device=dev)
    pu = poseupdate_cams([dict(K=dK.data_ptr(), xy=dxy[c].data_ptr(), state=dst[c].data_ptr(), slot2map=ds2m[c].data_ptr(),
                               reprojErr=d_err[c].data_ptr()) for c in range(nc)])
    M0, fl0 = G("mapPts").copy(), G("mapFlags").copy()
    # a few features off their points by 25 px (behind the solve, in front of the gate: outliers for the gate to find)
    xy = G("xy").copy()
    for c in range(nc):
        on = np.nonzero((G("state")[c] >= 0) & (G("slot2map")[c] >= 0))[0][::9]
        xy[c][on] += 25.0
    dxy.copy_(torch.from_numpy(xy))
    ic.apply_dev(ws, s, pu, N, dpf.data_ptr(), nMap, dR.data_ptr(), dT.data_ptr(), dM.data_ptr(), d_cov.data_ptr(), dfl.data_ptr(), float(np.sqrt(10.0)),
                 d_numNodes=d_nodes.data_ptr(), d_numOut=d_out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(dR.cpu().numpy().reshape(nc, 3, 3), r["Rs"]) and np.array_equal(dT.cpu().numpy(), r["Ts"])
    assert np.abs(r["Rs"].reshape(nc, 9) - G("curR")).max() > 1e-4   # (the solve moved them)
    Mw, cw, fw = M0.copy(), cov0.copy(), fl0.copy()
    errs = [np.zeros(N) for _ in range(nc)]
    want = oracle.pose_update_gate([r["K"]] * nc, r["Rs"].reshape(nc, 9), r["Ts"], list(xy), list(G("state")), list(G("slot2map")), Mw, cw, fw, 0,
                                   float(np.sqrt(10.0)), errs)
    assert np.array_equal(dM.cpu().numpy(), Mw) and np.array_equal(d_cov.cpu().numpy(), cw) and np.array_equal(dfl.cpu().numpy(), fw)
    assert np.array_equal(d_err.cpu().numpy(), np.stack(errs))
    assert d_nodes.cpu().numpy().tolist() == [w[0] for w in want] and d_out.cpu().numpy().tolist() == [w[1] for w in want]
    assert sum(w[0] for w in want) > 200 and (Mw != M0).any(axis=1).sum() > 100 and 0 < sum(w[1] for w in want) < sum(w[0] for w in want)
