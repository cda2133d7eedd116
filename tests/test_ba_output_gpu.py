"""RobustBundleRTS::output() on the device (reference src/app/SL_CoSLAMRobustBA.cpp:273-316): cs_ba_output_* -- the worker packs a
window solve's result into a record; cs_ba_output_apply_dev writes the key poses into the pose history / the window ring, the
points into the map (outlier points set false), relaxes the non-key frames over the camera chains and re-triangulates the map
(updateNewPosesPoints).  Checked against the same steps put together from the oracle's pieces, each of which is pinned against
the reference's own code elsewhere (posegraph_golden, update_points_golden)."""
import numpy as np
import pytest

from tests.poseupdate_scene import Scene

pytestmark = pytest.mark.gpu
SIGMA, MAX_EPI = 10.0, 6.0


def _drive(T=27, first_key=2, key_every=5, n_kf=5, hist=40, seed=31, lag_frames=None, key_frames_at=None):
    import torch

    import coslam_amd
    from coslam_amd.ba import BAOutput, BAWindow, BAWorkspace
    from coslam_amd.handback import handback_cams
    from coslam_amd.poseupdate import TrackHistory

    sc = Scene(T=T, seed=seed)
    nC, N, nMap = sc.nC, sc.N, sc.nMap
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    th = TrackHistory(nC, N, hist)
    win = BAWindow(nC, n_kf, N, nMap)
    ws = BAWorkspace(0)
    win.reserve(ws)
    out = BAOutput(nC, n_kf, nMap, n_slots=4)
    out.attach(ws)
    d_K = torch.from_numpy(sc.K.reshape(9).copy()).to(dev)
    d_iK = torch.from_numpy(sc.iK.reshape(9).copy()).to(dev)
    flags0 = sc.flags0.copy()
    d_fl = torch.from_numpy(flags0.copy()).to(dev)
    d_map = torch.from_numpy(sc.map0.copy()).to(dev)
    d_cov = torch.from_numpy(sc.cov0.copy()).to(dev)
    d_stat = [torch.ones(N, dtype=torch.uint8, device=dev) for _ in range(nC)]
    hR, hT, hXY = [[] for _ in range(nC)], [[] for _ in range(nC)], [[] for _ in range(nC)]
    keep, key_frames, kf_recs = [], [], []
    for f in range(T):
        recs = sc.frame(f)
        Rs = np.stack([sc.Re[f][c].reshape(9) for c in range(nC)])
        ts = np.stack([sc.te[f][c] for c in range(nC)])
        cams, hb = [], []
        for c, r in enumerate(recs):
            t_ = {k: torch.from_numpy(v).to(dev) for k, v in r.items()}
            keep.append(t_)
            cams.append(dict(K=d_K.data_ptr(), iK=d_iK.data_ptr(), xy=t_["xy"].data_ptr(), state=t_["state"].data_ptr(),
                             slot2map=t_["slot2map"].data_ptr(), trackSpan=t_["trackSpan"].data_ptr(), isStatic=d_stat[c].data_ptr()))
            hb.append(dict(xy=t_["xy"].data_ptr(), state=t_["state"].data_ptr(), slot2map=t_["slot2map"].data_ptr()))
            hR[c].insert(0, Rs[c].copy()), hT[c].insert(0, ts[c].copy()), hXY[c].insert(0, r["xy"].copy())
        d_R, d_t = torch.from_numpy(Rs).to(dev), torch.from_numpy(ts).to(dev)
        keep += [d_R, d_t]
        th.detect_dynamic_dev(s, cams, d_R.data_ptr(), d_t.data_ptr(), nMap, d_fl.data_ptr(), f, maxEpiErr=MAX_EPI)
        is_key = f in key_frames_at if key_frames_at is not None else (f >= first_key and (f - first_key) % key_every == 0)
        if is_key and len(key_frames) < n_kf:
            win.push_dev(s, handback_cams(hb), d_K.data_ptr(), 1, d_R.data_ptr(), d_t.data_ptr(), f)
            key_frames.append(f)
            kf_recs.append(recs)
            if len(key_frames) == n_kf:
                win.solve_flags_async(ws, s, d_map.data_ptr(), d_fl.data_ptr(), 2 * nC, 2, 6.0, 2, 10)
    torch.cuda.synchronize()
    return dict(sc=sc, th=th, win=win, ws=ws, out=out, cams=cams, recs=recs, key_frames=key_frames, kf_recs=kf_recs, d_K=d_K, d_iK=d_iK,
                d_fl=d_fl, d_map=d_map, d_cov=d_cov, d_stat=d_stat, hR=hR, hT=hT, hXY=hXY, keep=keep, dev=dev, s=s, Rs_last=Rs, ts_last=ts,
                flags0=flags0)


def test_the_worker_packs_every_window_solve_into_a_record(hip):
    import torch

    D = _drive()
    sc, ws, out, win = D["sc"], D["ws"], D["out"], D["win"]
    rec = out.wait(0)
    assert out.packed() == 1
    hd = out.header(rec, D["s"])
    Cw, Pw, Ow, d_pm, kfs = win.last_problem()
    assert hd["ok"] == 1 and hd["seq"] == 0 and hd["nKf"] == 5 and hd["nCams"] == sc.nC
    assert (hd["C"], hd["P"], hd["nObs"]) == (Cw, Pw, Ow) and hd["key_frames"] == D["key_frames"] == kfs and Pw > 100
    ws.set_sizes(Cw, Pw, Ow)
    Rs, Ts, pts, outl, st = ws.download()
    assert st.nIterTotal > 0 and outl.sum() > 0
    from coslam_amd.multicam import _DevArray

    pR, pT, pM, pMap, pOut = out.arrays(rec)
    g = lambda p, n, ty: torch.as_tensor(_DevArray(p, n, ty), device=D["dev"]).cpu().numpy()   # noqa: E731
    assert np.array_equal(g(pR, 9 * Cw, "<f8").reshape(Cw, 3, 3), Rs) and np.array_equal(g(pT, 3 * Cw, "<f8").reshape(Cw, 3), Ts)
    assert np.array_equal(g(pM, 3 * Pw, "<f8").reshape(Pw, 3), pts)
    pm = g(d_pm, Pw, "<i4")
    assert np.array_equal(g(pMap, Pw, "<i4"), pm)
    # only points that are isLocalStatic() took part (the flags form of the request)
    assert not (D["flags0"][pm] & 3).any()
    # a point is an outlier point when any of its measurements is
    optr = g(ws.problem_buffers()[1], Pw + 1, "<i4")
    want = np.array([outl[optr[i]:optr[i + 1]].any() for i in range(Pw)], dtype=np.uint8)
    assert np.array_equal(g(pOut, Pw, "|u1"), want) and want.sum() > 0


@pytest.mark.parametrize("kf_at", [None, [2, 5, 11, 14, 22], [3, 4, 5, 19, 26]])
def test_apply_writes_poses_points_and_flags_back_and_relaxes_the_non_key_frames(hip, kf_at):
    """kf_at: key frames where a decision put them (cs_ba_output_apply_frames_dev; the last case: neighbouring key frames, and the
    newest frame itself a key frame -- no free tail) instead of the fixed cadence."""
    import torch

    import oracle
    from coslam_amd.multicam import _DevArray
    from coslam_amd.poseupdate import poseupdate_cams

    D = _drive(key_frames_at=kf_at)
    sc, th, ws, out, win, s, dev = D["sc"], D["th"], D["ws"], D["out"], D["win"], D["s"], D["dev"]
    nC, N, nMap, T = sc.nC, sc.N, sc.nMap, sc.T
    rec = out.wait(0)
    Cw, Pw, Ow, d_pm, kfs = win.last_problem()
    assert kf_at is None or kfs == kf_at
    ws.set_sizes(Cw, Pw, Ow)
    Rs, Ts, pts, outl, _ = ws.download()
    g = lambda p, n, ty: torch.as_tensor(_DevArray(p, n, ty), device=dev).cpu().numpy()   # noqa: E731
    pm = g(d_pm, Pw, "<i4")
    ptOut = g(out.arrays(rec)[4], Pw, "|u1")
    first_key, key_every, n_kf = kfs[0], 5, 5
    newest = T - 1
    nN = newest - first_key + 1
    pf = Scene.point_feat(D["recs"], nMap)
    d_pf = torch.from_numpy(pf).to(dev)
    d_Rc = torch.from_numpy(D["Rs_last"].copy()).to(dev)
    d_tc = torch.from_numpy(D["ts_last"].copy()).to(dev)
    d_cnt = torch.zeros(3, dtype=torch.int32, device=dev)
    M0, cov0, fl0 = D["d_map"].cpu().numpy().copy(), D["d_cov"].cpu().numpy().copy(), D["d_fl"].cpu().numpy().copy()
    if kf_at is None:
        out.apply_dev(rec, s, th, win, poseupdate_cams(D["cams"]), d_pf.data_ptr(), nMap, D["d_map"].data_ptr(), D["d_cov"].data_ptr(),
                      D["d_fl"].data_ptr(), SIGMA, first_key, key_every, d_Rc.data_ptr(), d_tc.data_ptr(), d_cnt.data_ptr())
    else:   # with the sequence number stated: every key frame's number is held against the record's header
        out.apply_frames_dev(rec, s, th, win, poseupdate_cams(D["cams"]), d_pf.data_ptr(), nMap, D["d_map"].data_ptr(), D["d_cov"].data_ptr(),
                             D["d_fl"].data_ptr(), SIGMA, kfs, d_Rc.data_ptr(), d_tc.data_ptr(), d_cnt.data_ptr(), seq=0)
    node_of = [f - first_key for f in kfs]
    d_nR = torch.zeros((nC, nN, 9), dtype=torch.float64, device=dev)
    d_nT = torch.zeros((nC, nN, 3), dtype=torch.float64, device=dev)
    th.get_span_dev(s, first_key, nN, d_nR.data_ptr(), d_nT.data_ptr())
    torch.cuda.synchronize()
    nR, nT = d_nR.cpu().numpy(), d_nT.cpu().numpy()
    # --- the camera chains: edges from the poses as tracked, the key frames fixed at the adjusted poses, the rest relaxed
    id1, id2 = np.arange(nN - 1), np.arange(1, nN)
    fixed = np.zeros(nN, dtype=np.uint8)
    fixed[node_of] = 1
    moved = 0.0
    for c in range(nC):
        R0 = np.stack([D["hR"][c][newest - f] for f in range(first_key, newest + 1)])
        t0 = np.stack([D["hT"][c][newest - f] for f in range(first_key, newest + 1)])
        eR, eT = oracle.posegraph_edges(R0, t0, id1, id2)
        for j in range(n_kf):
            R0[node_of[j]], t0[node_of[j]] = Rs[j * nC + c].reshape(9), Ts[j * nC + c]
        rc, wR, wT = oracle.posegraph_relax(fixed, R0, t0, id1, id2, eR, eT)
        assert rc == 0
        assert np.abs(nR[c] - wR).max() < 1e-9 and np.abs(nT[c] - wT).max() < 1e-9, (c, np.abs(nR[c] - wR).max(), np.abs(nT[c] - wT).max())
        for j in range(n_kf):   # the adjusted key poses themselves: copied, bit for bit
            assert np.array_equal(nR[c, node_of[j]], Rs[j * nC + c].reshape(9)) and np.array_equal(nT[c, node_of[j]], Ts[j * nC + c])
        moved = max(moved, np.abs(nT[c, -1] - D["hT"][c][0]).max())
        # the newest relaxed pose is the camera's current pose
        assert np.array_equal(d_Rc.cpu().numpy()[c], nR[c, -1]) and np.array_equal(d_tc.cpu().numpy()[c], nT[c, -1])
    assert moved > 1e-6, "the free tail of the chains followed the last key frame (or IS the last key frame)"
    assert out.wait_errors() == 0
    # --- the window's ring holds the adjusted key poses now: a second request parses them as its start
    # --- the map: adjusted points, outlier points false, then updateNewPosesPoints with the relaxed history
    M, cov, fl = M0.copy(), cov0.copy(), fl0.copy()
    M[pm] = pts
    fl[pm[ptOut > 0]] = (fl[pm[ptOut > 0]] & ~np.uint8(1)) | np.uint8(2)
    assert ptOut.sum() > 0
    histR = np.stack([np.stack(D["hR"][c]) for c in range(nC)])
    histT = np.stack([np.stack(D["hT"][c]) for c in range(nC)])
    histXY = np.stack([np.stack(D["hXY"][c]) for c in range(nC)])
    for f in range(first_key, newest + 1):
        histR[:, newest - f], histT[:, newest - f] = nR[:, f - first_key], nT[:, f - first_key]
    span = np.stack([r["trackSpan"] for r in D["recs"]])
    fstat = np.stack([d.cpu().numpy() for d in D["d_stat"]])
    n, ns, nd, _ = oracle.update_new_poses_points([sc.K] * nC, [sc.iK] * nC, histR, histT, histXY, span, fstat, pf, M, cov, fl, SIGMA,
                                                  firstKeyFrame=first_key)
    assert ns > 100
    cnt = d_cnt.cpu().tolist()
    assert cnt[0] == ns and cnt[1] == nd and cnt[2] == int(((fl & 2) != 0).sum() - ((fl0 & 2) != 0).sum())
    assert np.array_equal(D["d_fl"].cpu().numpy(), fl)
    assert np.array_equal(D["d_map"].cpu().numpy(), M) and np.array_equal(D["d_cov"].cpu().numpy(), cov)
    # the adjusted points that updateNewPosesPoints did not touch are the solver's, bit for bit
    assert (np.abs(D["d_map"].cpu().numpy() - M0).max(axis=1) > 0).sum() > 100


def test_a_second_window_starts_from_the_adjusted_key_poses(hip):
    """output() writes through the CamPoseItems the key frames share with the next window (SL_CoSLAMRobustBA.cpp:283-285): after
    cs_ba_output_apply_dev the window's ring holds the adjusted key poses, and a request made afterwards parses them as its start --
    while a request made BEFORE the apply keeps the poses as they stood when it was made."""
    import torch

    from coslam_amd.multicam import _DevArray
    from coslam_amd.poseupdate import poseupdate_cams

    D = _drive()
    sc, th, ws, out, win, s, dev = D["sc"], D["th"], D["ws"], D["out"], D["win"], D["s"], D["dev"]
    nC, nMap = sc.nC, sc.nMap
    rec = out.wait(0)
    Cw, Pw, Ow, _, kfs = win.last_problem()
    ws.set_sizes(Cw, Pw, Ow)
    Rs1, Ts1 = ws.download()[:2]
    d_pf = torch.from_numpy(Scene.point_feat(D["recs"], nMap)).to(dev)
    d_Rc, d_tc = torch.from_numpy(D["Rs_last"].copy()).to(dev), torch.from_numpy(D["ts_last"].copy()).to(dev)
    out.apply_dev(rec, s, th, win, poseupdate_cams(D["cams"]), d_pf.data_ptr(), nMap, D["d_map"].data_ptr(), D["d_cov"].data_ptr(),
                  D["d_fl"].data_ptr(), SIGMA, kfs[0], 5, d_Rc.data_ptr(), d_tc.data_ptr())
    # same window again, zero LM steps: the estimate the solve "returns" is the parse's start
    win.solve_flags_async(ws, s, D["d_map"].data_ptr(), D["d_fl"].data_ptr(), 2 * nC, 2, 6.0, 0, 0)
    rec2 = out.wait(1)
    hd = out.header(rec2, s)
    assert hd["seq"] == 1 and hd["key_frames"] == kfs
    g = lambda p, n, ty: torch.as_tensor(_DevArray(p, n, ty), device=dev).cpu().numpy()   # noqa: E731
    R2 = g(out.arrays(rec2)[0], 9 * Cw, "<f8").reshape(Cw, 3, 3)
    T2 = g(out.arrays(rec2)[1], 3 * Cw, "<f8").reshape(Cw, 3)
    assert np.array_equal(R2, Rs1) and np.array_equal(T2, Ts1)
    assert np.abs(Rs1 - np.stack([np.asarray(sc.Re[f][c]) for f in kfs for c in range(nC)])).max() > 1e-7
