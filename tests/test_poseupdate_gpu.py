"""poseUpdate3D's gate + seqTriangulate loop and detectDynamicFeaturePoints on the device (cs_pose_update3d_dev,
cs_detect_dynamic_dev, cs_pose_update_frame_dev) against the oracle's restatement of the reference's loops
(src/app/SL_SingleSLAM.cpp:672-708, 784-824), frame after frame with the state carried along on both sides."""
import numpy as np
import pytest

from tests.poseupdate_scene import Scene

pytestmark = pytest.mark.gpu

SIGMA = 10.0   # Const::PIXEL_ERR_VAR (src/app/SL_GlobParam.cpp:37)
MAX_EPI = 6.0  # Const::MAX_EPI_ERR (:36)


def _run(mode, largeErr=0, H=32, T=14):
    import torch

    import oracle
    from coslam_amd.poseupdate import TrackHistory, pose_update3d_dev

    sc = Scene(T=T)
    nC, N, nMap = sc.nC, sc.N, sc.nMap
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    # oracle state
    o_map, o_cov, o_fl = sc.map0.copy(), sc.cov0.copy(), sc.flags0.copy()
    o_err = [np.zeros(N) for _ in range(nC)]
    o_stat = [np.ones(N, dtype=np.uint8) for _ in range(nC)]
    hist = [dict(R=[], t=[], xy=[]) for _ in range(nC)]
    # device state
    d_map = torch.from_numpy(sc.map0.copy()).to(dev)
    d_cov = torch.from_numpy(sc.cov0.copy()).to(dev)
    d_fl = torch.from_numpy(sc.flags0.copy()).to(dev)
    d_err = [torch.zeros(N, dtype=torch.float64, device=dev) for _ in range(nC)]
    d_stat = [torch.ones(N, dtype=torch.uint8, device=dev) for _ in range(nC)]
    d_K = torch.from_numpy(sc.K.reshape(9).copy()).to(dev)
    d_iK = torch.from_numpy(sc.iK.reshape(9).copy()).to(dev)
    d_cnt = torch.zeros(3, nC, dtype=torch.int32, device=dev)
    th = TrackHistory(nC, N, H)
    exact = True
    n_dyn_total = n_out_total = n_in_total = 0
    for f in range(sc.T):
        recs = sc.frame(f)
        pf = Scene.point_feat(recs, nMap)
        Rs = np.stack([sc.Re[f][c].reshape(9) for c in range(nC)])
        ts = np.stack([sc.te[f][c] for c in range(nC)])
        # ---- oracle: the cameras one after the other, gate then dynamic test (CoSLAM::parallelPoseUpdate) ----
        res = oracle.pose_update_gate([sc.K] * nC, Rs, ts, [r["xy"] for r in recs], [r["state"] for r in recs],
                                      [r["slot2map"] for r in recs], o_map, o_cov, o_fl, largeErr, SIGMA, o_err)
        o_dyn = []
        for c in range(nC):
            h = hist[c]
            h["R"].insert(0, Rs[c]), h["t"].insert(0, ts[c]), h["xy"].insert(0, recs[c]["xy"].copy())
            del h["R"][H:], h["t"][H:], h["xy"][H:]
            o_dyn.append(oracle.detect_dynamic(sc.iK, np.stack(h["R"]), np.stack(h["t"]), np.stack(h["xy"]), recs[c]["state"],
                                               recs[c]["slot2map"], recs[c]["trackSpan"], o_fl, 20, 5, 3, MAX_EPI, o_stat[c]))
        # ---- device ----
        keep, cams = [], []
        for c, r in enumerate(recs):
            t_ = {k: torch.from_numpy(v).to(dev) for k, v in r.items()}
            keep.append(t_)
            cams.append(dict(K=d_K.data_ptr(), iK=d_iK.data_ptr(), xy=t_["xy"].data_ptr(), state=t_["state"].data_ptr(),
                             slot2map=t_["slot2map"].data_ptr(), trackSpan=t_["trackSpan"].data_ptr(),
                             reprojErr=d_err[c].data_ptr(), isStatic=d_stat[c].data_ptr()))
        d_pf = torch.from_numpy(pf).to(dev)
        d_R, d_t = torch.from_numpy(Rs).to(dev), torch.from_numpy(ts).to(dev)
        if mode == "fused":
            th.pose_update_frame_dev(s, cams, d_pf.data_ptr(), nMap, d_R.data_ptr(), d_t.data_ptr(), d_map.data_ptr(),
                                     d_cov.data_ptr(), d_fl.data_ptr(), largeErr, SIGMA, f, maxEpiErr=MAX_EPI,
                                     d_numNodes=d_cnt[0].data_ptr(), d_numOut=d_cnt[1].data_ptr(), d_numDyn=d_cnt[2].data_ptr())
        else:   # camera by camera (the serial caller) through the two separate entry points
            for c in range(nC):
                pose_update3d_dev(s, cams, N, d_pf.data_ptr(), nMap, d_R.data_ptr(), d_t.data_ptr(), d_map.data_ptr(),
                                  d_cov.data_ptr(), d_fl.data_ptr(), largeErr, SIGMA, d_cnt[0].data_ptr(), d_cnt[1].data_ptr(),
                                  cam0=c, nCamsRun=1)
                th.detect_dynamic_dev(s, cams, d_R.data_ptr(), d_t.data_ptr(), nMap, d_fl.data_ptr(), f, maxEpiErr=MAX_EPI,
                                      d_numDyn=d_cnt[2].data_ptr(), cam0=c, nCamsRun=1)
        torch.cuda.synchronize()
        assert th.frames == min(f + 1, H)
        cnt = d_cnt.cpu().numpy()
        assert cnt[0].tolist() == [r[0] for r in res], f"nodes, frame {f}"
        assert cnt[1].tolist() == [r[1] for r in res], f"numOut, frame {f}"
        assert cnt[2].tolist() == o_dyn, f"numDyn, frame {f}"
        assert np.array_equal(d_fl.cpu().numpy(), o_fl), f"map flags, frame {f}"
        for c in range(nC):
            assert np.array_equal(d_stat[c].cpu().numpy(), o_stat[c]), f"feature types, frame {f} camera {c}"
            g = d_err[c].cpu().numpy()
            np.testing.assert_allclose(g, o_err[c], rtol=1e-11, atol=1e-13)
            exact &= np.array_equal(g, o_err[c])
        gm, gc = d_map.cpu().numpy(), d_cov.cpu().numpy()
        np.testing.assert_allclose(gm, o_map, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(gc, o_cov, rtol=1e-9, atol=1e-16)
        exact &= np.array_equal(gm, o_map) and np.array_equal(gc, o_cov)
        n_dyn_total += sum(o_dyn)
        n_out_total += sum(r[1] for r in res)
        n_in_total += sum(r[0] - r[1] for r in res)
    # the scene exercises every branch
    assert n_dyn_total > 20 and n_out_total > 5 and n_in_total > 500
    assert (o_fl & 4).sum() > (sc.flags0 & 4).sum()          # points were made uncertain
    assert not np.array_equal(o_map, sc.map0)                 # and points were refined
    assert all((st == 0).sum() > 3 for st in o_stat)          # dynamic features in every camera
    th.close()
    return exact


@pytest.mark.parametrize("mode", ["fused", "serial"])
def test_pose_update_gate_and_dynamic_test_match_the_oracle_frame_after_frame(mode):
    exact = _run(mode)
    assert exact, "floating-point outputs agree within tolerance but not bit for bit"


def test_large_err_gate_and_a_short_history_ring():
    # largeErr: the 6.0 gate (SL_SingleSLAM.cpp:673); a ring shorter than the tracks bounds the walk on both sides alike
    _run("fused", largeErr=1, H=6)


def test_history_is_dropped_when_the_frame_numbers_jump():
    import torch

    from coslam_amd.poseupdate import TrackHistory

    sc = Scene(T=3)
    dev = torch.device("cuda:0")
    th = TrackHistory(sc.nC, sc.N, 8)
    d_K = torch.from_numpy(sc.K.reshape(9).copy()).to(dev)
    d_iK = torch.from_numpy(sc.iK.reshape(9).copy()).to(dev)
    d_fl = torch.from_numpy(sc.flags0.copy()).to(dev)
    keep = []
    for frame in (0, 1, 2, 7, 8):
        recs = sc.frame(min(frame, 2))
        cams = []
        for r in recs:
            t_ = {k: torch.from_numpy(v).to(dev) for k, v in r.items()}
            st = torch.ones(sc.N, dtype=torch.uint8, device=dev)
            keep += [t_, st]
            cams.append(dict(K=d_K.data_ptr(), iK=d_iK.data_ptr(), xy=t_["xy"].data_ptr(), state=t_["state"].data_ptr(),
                             slot2map=t_["slot2map"].data_ptr(), trackSpan=t_["trackSpan"].data_ptr(), isStatic=st.data_ptr()))
        Rs = torch.from_numpy(np.stack([sc.Re[0][c].reshape(9) for c in range(sc.nC)])).to(dev)
        ts = torch.from_numpy(np.stack([sc.te[0][c] for c in range(sc.nC)])).to(dev)
        th.detect_dynamic_dev(torch.cuda.current_stream().cuda_stream, cams, Rs.data_ptr(), ts.data_ptr(), sc.nMap, d_fl.data_ptr(), frame)
        torch.cuda.synchronize()
        assert th.frames == {0: 1, 1: 2, 2: 3, 7: 1, 8: 2}[frame]
    th.close()


def test_bad_arguments_are_refused():
    from coslam_amd._lib import CoslamHipError
    from coslam_amd.poseupdate import TrackHistory, pose_update3d_dev

    with pytest.raises(CoslamHipError):
        TrackHistory(3, 100, 100000)
    with pytest.raises(CoslamHipError):
        pose_update3d_dev(0, [dict()], 10, 0, 5, 0, 0, 0, 0, 0, 0, SIGMA)
    th = TrackHistory(2, 16, 4)
    with pytest.raises(CoslamHipError):   # null map pointers
        th.update_new_poses_points_dev(0, [dict(), dict()], 0, 5, 0, 0, 0, SIGMA)
    with pytest.raises(CoslamHipError):   # a history without a frame
        th.update_new_poses_points_dev(0, [dict(), dict()], 0, 0, 0, 0, 0, SIGMA)
    with pytest.raises(CoslamHipError):
        th.set_poses_dev(0, 3, 0, 0, 0, 0)
    th.close()


def test_register_mergability_walks_the_candidates_tracks_like_the_oracle():
    """cs_register_mergability_dev: CoSLAM::staticCheckMergability (reference src/app/SL_CoSLAM.cpp:714-729) for every candidate of
    a registration search in one launch, the tracks' past pixels and the frames' poses from the history ring, against the oracle
    (which reproduces the reference's own function on tests/golden/mergability_golden.npz): every verdict equal."""
    import torch

    import oracle
    from coslam_amd.poseupdate import TrackHistory

    sc = Scene(T=12, seed=11)
    nC, N, nMap, H = sc.nC, sc.N, sc.nMap, 8     # (a ring shorter than the longest tracks: the walk is bounded alike on both sides)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    th = TrackHistory(nC, N, H)
    d_K = torch.from_numpy(sc.K.reshape(9).copy()).to(dev)
    d_iK = torch.from_numpy(sc.iK.reshape(9).copy()).to(dev)
    d_fl = torch.from_numpy(sc.flags0.copy()).to(dev)
    hist = [dict(R=[], t=[], xy=[]) for _ in range(nC)]
    keep = []
    for f in range(sc.T):
        recs = sc.frame(f)
        Rs = np.stack([sc.Re[f][c].reshape(9) for c in range(nC)])
        ts = np.stack([sc.te[f][c] for c in range(nC)])
        cams = []
        for c, r in enumerate(recs):
            t_ = {k: torch.from_numpy(v).to(dev) for k, v in r.items()}
            st = torch.ones(N, dtype=torch.uint8, device=dev)
            keep += [t_, st]
            cams.append(dict(K=d_K.data_ptr(), iK=d_iK.data_ptr(), xy=t_["xy"].data_ptr(), state=t_["state"].data_ptr(),
                             slot2map=t_["slot2map"].data_ptr(), trackSpan=t_["trackSpan"].data_ptr(), isStatic=st.data_ptr()))
            h = hist[c]
            h["R"].insert(0, Rs[c]), h["t"].insert(0, ts[c]), h["xy"].insert(0, r["xy"].copy())
            del h["R"][H:], h["t"][H:], h["xy"][H:]
        d_R, d_t = torch.from_numpy(Rs).to(dev), torch.from_numpy(ts).to(dev)
        th.detect_dynamic_dev(s, cams, d_R.data_ptr(), d_t.data_ptr(), nMap, d_fl.data_ptr(), f)   # (pushes the frame into the ring)
        torch.cuda.synchronize()
    # candidates: for every (map point, camera) the slot that tracks the point there, else (every third pair) some live slot
    rng = np.random.default_rng(4)
    slot = np.full((nMap, nC), -1, dtype=np.int32)
    for c in range(nC):
        live = np.nonzero(recs[c]["state"] >= 0)[0]
        by_pt = {int(sc.slotPt[c, i]): int(i) for i in live}
        for m in range(nMap):
            if m in by_pt:
                slot[m, c] = by_pt[m]
            elif m % 3 == 0:
                slot[m, c] = int(live[rng.integers(len(live))])
    n_true = {}
    for sigma in (1.2, 10.0):
        d_slot = torch.from_numpy(slot).to(dev)
        d_M = torch.from_numpy(sc.map0.copy()).to(dev)
        d_cov = torch.from_numpy(sc.cov0.copy()).to(dev)
        d_out = torch.full((nMap, nC), 77, dtype=torch.uint8, device=dev)
        th.register_mergability_dev(s, cams, nMap, d_M.data_ptr(), d_cov.data_ptr(), d_slot.data_ptr(), sigma, d_out.data_ptr())
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        want = np.full((nMap, nC), 255, dtype=np.uint8)
        for c in range(nC):
            h = hist[c]
            hR, hT, hXY = np.stack(h["R"]), np.stack(h["t"]), np.stack(h["xy"])
            span = recs[c]["trackSpan"]
            for m in range(nMap):
                sl = slot[m, c]
                if sl < 0:
                    continue
                ln = span[N + sl] - span[sl] + 1 if span[sl] >= 0 else 0
                ok = oracle.static_check_mergability(sc.K, hR, hT, hXY, sl, ln, sc.map0[m], sc.cov0[m], sigma)
                # (a track longer than the ring cannot be judged: 2 whatever the held frames say, never attached)
                want[m, c] = 2 if ln > H else (1 if ok else 0)
        assert np.array_equal(got, want), sigma
        n_true[sigma] = (int(((want == 1) | (want == 2)).sum()), int((want == 0).sum()))
        assert (want == 1).sum() > 10 and (want == 2).sum() > 10
    assert n_true[1.2][0] > 50 and n_true[1.2][1] > 200 and n_true[10.0][0] > n_true[1.2][0]   # both verdicts occur, the gate matters
    th.close()


def _feed_long_golden(g, th, f, dev, keep):
    """frame f of tests/golden/mergability_long_golden.npz into the history (cs_detect_dynamic_dev pushes it): slot = track"""
    import torch

    nC, nT, T = g["xy"].shape[:3]
    cams = []
    f1 = g["f1"]
    for c in range(nC):
        xy = np.nan_to_num(g["xy"][c, :, f])                       # [nT][2]
        alive = f >= f1[c]
        rec = dict(xy=np.concatenate([xy[:, 0], xy[:, 1]]), state=np.where(alive, np.where(f == f1[c], 1, 0), -1).astype(np.int32),
                   slot2map=np.full(nT, -1, np.int32),
                   trackSpan=np.concatenate([np.where(alive, f1[c], -1), np.where(alive, f, -1)]).astype(np.int32))
        t_ = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in rec.items()}
        st = torch.ones(nT, dtype=torch.uint8, device=dev)
        keep += [t_, st]
        cams.append(dict(K=keep[0][c].data_ptr(), iK=keep[1][c].data_ptr(), xy=t_["xy"].data_ptr(), state=t_["state"].data_ptr(),
                         slot2map=t_["slot2map"].data_ptr(), trackSpan=t_["trackSpan"].data_ptr(), isStatic=st.data_ptr()))
    d_R = torch.from_numpy(np.ascontiguousarray(g["R"][:, f])).to(dev)
    d_t = torch.from_numpy(np.ascontiguousarray(g["t"][:, f])).to(dev)
    keep += [d_R, d_t]
    s = torch.cuda.current_stream().cuda_stream
    th.detect_dynamic_dev(s, cams, d_R.data_ptr(), d_t.data_ptr(), 1, keep[2].data_ptr(), f, minLen=1 << 30)
    return cams


def test_running_mergability_verdict_equals_the_references_whole_track_walk():
    """cs_register_mergability_running_dev: CoSLAM::staticCheckMergability walks a candidate's WHOLE track
    (reference src/app/SL_CoSLAM.cpp:714-729); the device walks the newest 64 frames every frame and keeps the verdict over the older
    ones per (map point, camera), extended by one term per frame.  On tests/golden/mergability_long_golden.npz -- 144 tracks of up
    to 420 frames judged by the reference's own function compiled in place -- fed frame by frame into a history whose WALK depth is
    64:  (a) the running verdict at the last frame equals the reference's on every track, and at frames along the way the oracle's
    whole-track verdict of that moment;  (b) with a store too short to ever re-walk a tail (128 frames) the same holds, because a
    stable candidate never needs it;  (c) a cold cache at the last frame (512-frame store) walks every tail in full: same verdicts;
    with the short store the long tracks come out 2 (unjudged);  (d) a point moved by less than tolPix keeps its cached tail (tail
    terms of the old point AND window terms of the new one), one moved further has its tail judged again with the new point."""
    import os

    import torch

    import oracle
    from coslam_amd.poseupdate import TrackHistory

    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "mergability_long_golden.npz")))
    nC, nT, T = g["xy"].shape[:3]
    sigma, W = float(g["sigma"]), 64
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    P = nC * nT                                               # point p = camera p // nT, track p % nT; a candidate in its own camera only
    slot = np.full((P, nC), -1, np.int32)
    for c in range(nC):
        slot[c * nT:(c + 1) * nT, c] = np.arange(nT)
    M0, cov0 = g["M"].reshape(P, 3).copy(), g["cov"].reshape(P, 9).copy()
    d_M, d_cov = torch.from_numpy(M0).to(dev), torch.from_numpy(cov0).to(dev)

    def whole_track(f, Mq=None, lo=0, hi=None):
        """the oracle's verdict at frame f over walk depths lo .. hi - 1 of every track alive at f (255: not alive)"""
        out = np.full((P, nC), 255, np.uint8)
        for c in range(nC):
            histR, histT = g["R"][c][:f + 1][::-1], g["t"][c][:f + 1][::-1]
            for k in range(nT):
                if f < g["f1"][c, k]:
                    continue
                L = f - int(g["f1"][c, k]) + 1
                a, b = lo, L if hi is None else min(hi, L)
                if b <= a:
                    out[c * nT + k, c] = 1
                    continue
                hxy = np.ascontiguousarray(np.nan_to_num(g["xy"][c, k][:f + 1][::-1]))
                p = c * nT + k
                out[p, c] = 1 if oracle.static_check_mergability(g["K"][c], histR[a:b], histT[a:b], hxy[a:b], 0, b - a,
                                                                 (M0 if Mq is None else Mq)[p], cov0[p], sigma) else 0
        return out

    finals = {}
    for store in (512, 128):
        th = TrackHistory(nC, nT, W, storeLen=store)
        keep = [torch.from_numpy(g["K"].copy()).to(dev), torch.from_numpy(np.stack([np.linalg.inv(k.reshape(3, 3)).ravel() for k in g["K"]])).to(dev),
                torch.zeros(4, dtype=torch.uint8, device=dev)]
        cache = torch.zeros(th.mergability_cache_bytes(P), dtype=torch.uint8, device=dev)
        cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        d_out = torch.full((P, nC), 77, dtype=torch.uint8, device=dev)
        for f in range(T):
            cams = _feed_long_golden(g, th, f, dev, keep)
            live = (f >= g["f1"]).reshape(P)
            sl = np.where(live[:, None], slot, -1).astype(np.int32)
            d_slot = torch.from_numpy(sl).to(dev)
            th.register_mergability_running_dev(s, cams, P, d_M.data_ptr(), d_cov.data_ptr(), d_slot.data_ptr(), sigma, cache.data_ptr(),
                                                d_out.data_ptr(), tolPix=0.0, d_counts=cnt.data_ptr())
            if f in (70, 150, 229, 300, 377, T - 1):
                torch.cuda.synchronize()
                want = whole_track(f)
                assert np.array_equal(d_out.cpu().numpy(), want), (store, f)
            del keep[3:-2 * nC - 2]                            # (the records of older frames are no longer read)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        for c in range(nC):                                   # (a): the reference's own verdicts
            assert np.array_equal(got[c * nT:(c + 1) * nT, c], g["verdict"][c].astype(np.uint8)), (store, c)
        hits, walks, cuts, terms = cnt.cpu().tolist()
        tail_frames = int(np.clip(T - g["f1"] - W, 0, None).sum())   # frames in which a track was longer than the window
        # every tail was built term by term: no tail walked from scratch, nothing unjudged
        assert cuts == 0 and walks == 0 and hits == tail_frames > 20000
        assert 0 < terms <= tail_frames                                   # (a failed tail stops growing)
        finals[store] = (th, cams, cache, keep)
        # (c) a cold cache at the last frame
        cold = torch.zeros_like(cache)
        cnt.zero_()
        d_cold = torch.full((P, nC), 77, dtype=torch.uint8, device=dev)
        th.register_mergability_running_dev(s, cams, P, d_M.data_ptr(), d_cov.data_ptr(), d_slot.data_ptr(), sigma, cold.data_ptr(),
                                            d_cold.data_ptr(), tolPix=0.0, d_counts=cnt.data_ptr())
        torch.cuda.synchronize()
        gc_, long_ = d_cold.cpu().numpy(), (T - g["f1"] > store).reshape(P)
        win_ok = np.stack([whole_track(T - 1, None, 0, W)[c * nT:(c + 1) * nT, c] for c in range(nC)]).reshape(P) == 1   # (the window decides first)
        has_tail = (T - g["f1"] - W > 16).reshape(P)   # (a tail of up to 16 frames is walked by the candidate's own lanes: not counted as a walk)
        if store == 512:
            assert np.array_equal(gc_, got) and cnt[1].item() == int((has_tail & win_ok).sum()) > 60 and cnt[2].item() == 0
        else:
            for c in range(nC):
                sl_ = slice(c * nT, (c + 1) * nT)
                col, lg, wk = gc_[sl_, c], long_[sl_], win_ok[sl_]
                assert (col[lg & wk] == 2).all() and (col[lg & ~wk] == 0).all() and np.array_equal(col[~lg], got[sl_, c][~lg])
            assert cnt[2].item() == int((long_ & win_ok).sum()) > 40
    # (d) points that move: by a hair (the cached tail stays) and by a lot (the tail is judged again), on the 512-frame store
    th, cams, cache, keep = finals[512]
    rng = np.random.default_rng(5)
    Mh = M0 + rng.normal(0, 2e-4, M0.shape)                   # ~0.01 px
    Mf = M0 + rng.normal(0, 0.25, M0.shape)                   # ~15 px: many verdicts flip
    live_slot = torch.from_numpy(slot).to(dev)
    for Mq, moved_far in ((Mh, False), (Mf, True)):
        c2 = cache.clone()
        cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        d_o = torch.full((P, nC), 77, dtype=torch.uint8, device=dev)
        d_Mq = torch.from_numpy(Mq).to(dev)
        th.register_mergability_running_dev(s, cams, P, d_Mq.data_ptr(), d_cov.data_ptr(), live_slot.data_ptr(), sigma, c2.data_ptr(), d_o.data_ptr(),
                                            tolPix=0.5, d_counts=cnt.data_ptr())
        torch.cuda.synchronize()
        got = d_o.cpu().numpy()
        if moved_far:
            want = whole_track(T - 1, Mq)
            wq = np.stack([whole_track(T - 1, Mq, 0, W)[c * nT:(c + 1) * nT, c] for c in range(nC)]).reshape(P) == 1
            assert cnt[1].item() >= int(((T - g["f1"] - W > 16).reshape(P) & wq).sum()) - 8 > 5    # (a point that happened to move < 0.5 px keeps its tail)
            assert (got != whole_track(T - 1)).sum() > 10
        else:
            tail, win = whole_track(T - 1, M0, W, None), whole_track(T - 1, Mq, 0, W)
            want = np.where(tail == 255, 255, np.minimum(tail, win)).astype(np.uint8)
            assert cnt[1].item() == 0
        assert np.array_equal(got, want), moved_far
    # (e) poses rewritten behind cached tails (an apply whose window reaches back beyond the walk depth: cs_track_history_set_span_dev): a span
    # inside the newest 64 frames leaves the cache alone; a span of TAIL frames makes every cached verdict that covers it stale -- the tails are
    # walked again with the poses as they stand, and the verdict is the whole-track walk's over the rewritten poses
    for f0, n, tail_span in ((T - 40, 30, False), (120, 60, True)):
        R_span, t_span = np.ascontiguousarray(g["R"][:, f0:f0 + n]), np.ascontiguousarray(g["t"][:, f0:f0 + n]).copy()
        t_span[0] += np.array([0.35, -0.2, 0.1])              # camera 0's poses of the span moved: many of its terms fail now
        d_Rs, d_ts = torch.from_numpy(R_span).to(dev), torch.from_numpy(t_span).to(dev)
        th.set_span_dev(s, f0, n, d_Rs.data_ptr(), d_ts.data_ptr())
        t_keep = g["t"].copy()
        g["t"][:, f0:f0 + n] = t_span
        c2 = cache.clone()
        cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        d_o = torch.full((P, nC), 77, dtype=torch.uint8, device=dev)
        th.register_mergability_running_dev(s, cams, P, d_M.data_ptr(), d_cov.data_ptr(), live_slot.data_ptr(), sigma, c2.data_ptr(), d_o.data_ptr(),
                                            tolPix=0.0, d_counts=cnt.data_ptr())
        torch.cuda.synchronize()
        got, want = d_o.cpu().numpy(), whole_track(T - 1)
        if tail_span:
            assert np.array_equal(got, want) and cnt[1].item() > 20 and (want != whole_track_before).sum() > 5, (cnt.tolist(), int((got != want).sum()))
        else:
            assert cnt[1].item() == 0 and np.array_equal(got, want)   # (window frames are judged afresh every call anyway)
            whole_track_before = want
        # the span back as it was (the next round of the loop, and the handles' owners, see the golden poses)
        th.set_span_dev(s, f0, n, d_Rs.data_ptr(), torch.from_numpy(np.ascontiguousarray(t_keep[:, f0:f0 + n])).to(dev).data_ptr())
        g["t"][:] = t_keep
    for v in finals.values():
        v[0].close()


def test_update_new_poses_points_reproduces_the_reference_on_its_golden_scenes():
    """cs_update_new_poses_points_dev against tests/golden/update_points_golden.npz -- what the reference's own
    RobustBundleRTS::updateNewPosesPoints + updateStaticPointPosition / updateDynamicPointPosition (src/app/SL_CoSLAMRobustBA.cpp:
    248-271, src/slam/SL_CoSLAMHelper.cpp:338-394, 455-484, compiled in place) made of six scenes: every point and covariance bit for
    bit (the un-vendored triangulation helpers are our definitions on both sides; the loops are the reference's).  The ring is
    filled frame by frame with placeholder poses and then given the scene's poses through cs_track_history_set_poses_dev, the way
    the adjusted poses arrive after a bundle adjustment."""
    import os

    import torch

    from coslam_amd.poseupdate import TrackHistory

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "update_points_golden.npz"))
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    touched = 0
    for sc in range(int(g["n_scenes"])):
        G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
        hR, hT, hXY, span = G("histR"), G("histT"), G("histXY"), G("trackSpan")
        nC, H = hR.shape[0], hR.shape[1]
        N = hXY.shape[2] // 2
        cur, nMap = int(G("curFrame")), G("M0").shape[0]
        th = TrackHistory(nC, N, H + 3)   # (a ring deeper than the history: the unused entries are never walked)
        d_K = torch.from_numpy(G("K").copy()).to(dev)
        d_iK = torch.from_numpy(G("iK").copy()).to(dev)
        d_fl = torch.from_numpy(G("flags").copy()).to(dev)
        d_span = torch.from_numpy(span.copy()).to(dev)
        d_fstat = torch.from_numpy(G("featStatic").copy()).to(dev)
        d_scratch = torch.ones((nC, N), dtype=torch.uint8, device=dev)
        d_s2m = torch.full((nC, N), -1, dtype=torch.int32, device=dev)
        eye = torch.from_numpy(np.tile(np.eye(3).reshape(9), (nC, 1))).to(dev)
        zero = torch.zeros((nC, 3), dtype=torch.float64, device=dev)
        keep = []
        for j in range(H - 1, -1, -1):   # oldest first; entry j = frame cur - j
            xy = torch.from_numpy(hXY[:, j].copy()).to(dev)
            alive = ((span[:, :N] >= 0) & (span[:, :N] <= cur - j)).astype(np.int32) - 1   # 0 tracked / -1 dead at that frame
            st = torch.from_numpy(alive).to(dev)
            keep += [xy, st]
            cams = [dict(K=d_K[c].data_ptr(), iK=d_iK[c].data_ptr(), xy=xy[c].data_ptr(), state=st[c].data_ptr(),
                         slot2map=d_s2m[c].data_ptr(), trackSpan=d_span[c].data_ptr(), isStatic=d_scratch[c].data_ptr()) for c in range(nC)]
            th.detect_dynamic_dev(s, cams, eye.data_ptr(), zero.data_ptr(), nMap, d_fl.data_ptr(), cur - j, minLen=1 << 30)
        assert th.frames == H
        # the poses: every (camera, frame) of the history plus pairs the ring does not hold (skipped)
        cam_i = np.repeat(np.arange(nC), H).astype(np.int32)
        frm_i = np.tile(cur - np.arange(H), nC).astype(np.int32)
        cam_i = np.concatenate([cam_i, [0, 1, nC, -1]]).astype(np.int32)
        frm_i = np.concatenate([frm_i, [cur - H, cur + 1, cur, cur]]).astype(np.int32)
        Rq = np.concatenate([hR.reshape(-1, 9), np.full((4, 9), 777.0)])
        tq = np.concatenate([hT.reshape(-1, 3), np.full((4, 3), 777.0)])
        d_ci, d_fi = torch.from_numpy(cam_i).to(dev), torch.from_numpy(frm_i).to(dev)
        d_Rq, d_tq = torch.from_numpy(Rq).to(dev), torch.from_numpy(tq).to(dev)
        th.set_poses_dev(s, len(cam_i), d_ci.data_ptr(), d_fi.data_ptr(), d_Rq.data_ptr(), d_tq.data_ptr())
        d_M = torch.from_numpy(G("M0").copy()).to(dev)
        d_cov = torch.from_numpy(G("cov0").copy()).to(dev)
        d_pf = torch.from_numpy(G("pointFeat").copy()).to(dev)
        d_lf = torch.from_numpy(G("lastFrame").copy()).to(dev)
        d_ic = torch.from_numpy(G("isCurrent").copy()).to(dev)
        d_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
        cams = [dict(K=d_K[c].data_ptr(), iK=d_iK[c].data_ptr(), trackSpan=d_span[c].data_ptr(), isStatic=d_fstat[c].data_ptr())
                for c in range(nC)]
        th.update_new_poses_points_dev(s, cams, d_pf.data_ptr(), nMap, d_M.data_ptr(), d_cov.data_ptr(), d_fl.data_ptr(),
                                       float(G("sigma")), d_lastFrame=d_lf.data_ptr(), d_isCurrent=d_ic.data_ptr(),
                                       firstKeyFrame=int(G("firstKey")), d_counts=d_cnt.data_ptr())
        torch.cuda.synchronize()
        M, cov = d_M.cpu().numpy(), d_cov.cpu().numpy()
        n_ref = int((G("M_ref") != G("M0")).any(axis=1).sum())
        assert int(d_cnt.sum()) == n_ref, f"scene {sc}: {d_cnt.tolist()} points re-triangulated, the reference touched {n_ref}"
        dM, dC = np.abs(M - G("M_ref")).max(), np.abs(cov - G("cov_ref")).max()
        assert np.array_equal(M, G("M_ref")) and np.array_equal(cov, G("cov_ref")), f"scene {sc}: max |dM| {dM:.3e}, max |dcov| {dC:.3e}"
        touched += n_ref
        # CoSLAM::refineMapPoint on the same history: the reference's own function (src/app/SL_CoSLAM.cpp:666-713) was run on a copy
        # of every point two cameras see; the selected points bit for bit, the others untouched
        d_M2 = torch.from_numpy(G("M0").copy()).to(dev)
        d_cov2 = torch.from_numpy(G("cov0").copy()).to(dev)
        d_sel = torch.from_numpy(G("refine_select").copy()).to(dev)
        d_n = torch.zeros(1, dtype=torch.int32, device=dev)
        th.refine_map_points_dev(s, cams, d_pf.data_ptr(), nMap, d_M2.data_ptr(), d_cov2.data_ptr(), float(G("sigma")),
                                 d_select=d_sel.data_ptr(), d_count=d_n.data_ptr())
        torch.cuda.synchronize()
        assert int(d_n.item()) == int(G("refine_select").sum())
        M2, cov2 = d_M2.cpu().numpy(), d_cov2.cpu().numpy()
        assert np.array_equal(M2, G("M_refine")) and np.array_equal(cov2, G("cov_refine")), \
            f"scene {sc} refine: max |dM| {np.abs(M2 - G('M_refine')).max():.3e}"
        th.close()
    assert touched > 150


def test_update_new_poses_points_after_a_tracked_sequence_matches_the_oracle():
    """The same launch at the end of a tracked multi-camera sequence (512 slots x 3 cameras, 600 map points, ring shorter than the
    longest tracks), after the poses of the last frames were replaced the way a bundle adjustment + relaxation replaces them:
    against the oracle on the same arrays, bit for bit, with the feature types the dynamic test left behind."""
    import torch

    import oracle
    from coslam_amd.poseupdate import TrackHistory

    sc = Scene(T=12, seed=23)
    nC, N, nMap, H = sc.nC, sc.N, sc.nMap, 8
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    th = TrackHistory(nC, N, H)
    d_K = torch.from_numpy(sc.K.reshape(9).copy()).to(dev)
    d_iK = torch.from_numpy(sc.iK.reshape(9).copy()).to(dev)
    d_fl = torch.from_numpy(sc.flags0.copy()).to(dev)
    d_stat = [torch.ones(N, dtype=torch.uint8, device=dev) for _ in range(nC)]
    o_stat = [np.ones(N, dtype=np.uint8) for _ in range(nC)]
    hist = [dict(R=[], t=[], xy=[]) for _ in range(nC)]
    keep = []
    for f in range(sc.T):
        recs = sc.frame(f)
        Rs = np.stack([sc.Re[f][c].reshape(9) for c in range(nC)])
        ts = np.stack([sc.te[f][c] for c in range(nC)])
        cams = []
        for c, r in enumerate(recs):
            t_ = {k: torch.from_numpy(v).to(dev) for k, v in r.items()}
            keep.append(t_)
            cams.append(dict(K=d_K.data_ptr(), iK=d_iK.data_ptr(), xy=t_["xy"].data_ptr(), state=t_["state"].data_ptr(),
                             slot2map=t_["slot2map"].data_ptr(), trackSpan=t_["trackSpan"].data_ptr(), isStatic=d_stat[c].data_ptr()))
            h = hist[c]
            h["R"].insert(0, Rs[c]), h["t"].insert(0, ts[c]), h["xy"].insert(0, r["xy"].copy())
            del h["R"][H:], h["t"][H:], h["xy"][H:]
            oracle.detect_dynamic(sc.iK, np.stack(h["R"]), np.stack(h["t"]), np.stack(h["xy"]), r["state"], r["slot2map"], r["trackSpan"],
                                  sc.flags0, 20, 5, 3, MAX_EPI, o_stat[c])
        d_R, d_t = torch.from_numpy(Rs).to(dev), torch.from_numpy(ts).to(dev)
        th.detect_dynamic_dev(s, cams, d_R.data_ptr(), d_t.data_ptr(), nMap, d_fl.data_ptr(), f, maxEpiErr=MAX_EPI)
        torch.cuda.synchronize()
    for c in range(nC):
        assert np.array_equal(d_stat[c].cpu().numpy(), o_stat[c])
    # "the adjustment": new poses for the last 6 frames of every camera
    rng = np.random.default_rng(9)
    last = sc.T - 1
    ci, fi, Rn, tn = [], [], [], []
    from tests.poseupdate_scene import rodrigues
    for c in range(nC):
        for j in range(6):
            R = rodrigues(rng.normal(0, 3e-4, 3)) @ hist[c]["R"][j].reshape(3, 3)
            t = hist[c]["t"][j] + rng.normal(0, 2e-3, 3)
            hist[c]["R"][j], hist[c]["t"][j] = R.reshape(9).copy(), t.copy()
            ci.append(c), fi.append(last - j), Rn.append(R.reshape(9)), tn.append(t)
    d_ci, d_fi = torch.from_numpy(np.array(ci, np.int32)).to(dev), torch.from_numpy(np.array(fi, np.int32)).to(dev)
    d_Rn, d_tn = torch.from_numpy(np.array(Rn)).to(dev), torch.from_numpy(np.array(tn)).to(dev)
    th.set_poses_dev(s, len(ci), d_ci.data_ptr(), d_fi.data_ptr(), d_Rn.data_ptr(), d_tn.data_ptr())
    pf = Scene.point_feat(recs, nMap)
    lastFrame = np.where(rng.random(nMap) < 0.1, last - 9, last).astype(np.int32)
    isCur = (rng.random(nMap) < 0.8).astype(np.uint8)
    first_key = last - 8
    o_M, o_cov = sc.map0.copy(), sc.cov0.copy()
    span = np.stack([r["trackSpan"] for r in recs])
    n, ns, nd, chosen = oracle.update_new_poses_points([sc.K] * nC, [sc.iK] * nC, np.stack([np.stack(h["R"]) for h in hist]),
                                                       np.stack([np.stack(h["t"]) for h in hist]), np.stack([np.stack(h["xy"]) for h in hist]),
                                                       span, np.stack(o_stat), pf, o_M, o_cov, sc.flags0, SIGMA, lastFrame=lastFrame,
                                                       isCurrent=isCur, firstKeyFrame=first_key)
    assert ns > 150 and nd > 5 and (chosen >= 1).sum() > 150 and chosen.max() == H - 1
    d_M, d_cov = torch.from_numpy(sc.map0.copy()).to(dev), torch.from_numpy(sc.cov0.copy()).to(dev)
    d_pf, d_lf, d_ic = torch.from_numpy(pf).to(dev), torch.from_numpy(lastFrame).to(dev), torch.from_numpy(isCur).to(dev)
    d_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    th.update_new_poses_points_dev(s, cams, d_pf.data_ptr(), nMap, d_M.data_ptr(), d_cov.data_ptr(), d_fl.data_ptr(), SIGMA,
                                   d_lastFrame=d_lf.data_ptr(), d_isCurrent=d_ic.data_ptr(), firstKeyFrame=first_key,
                                   d_counts=d_cnt.data_ptr())
    torch.cuda.synchronize()
    assert d_cnt.tolist() == [ns, nd]
    M, cov = d_M.cpu().numpy(), d_cov.cpu().numpy()
    dM, dC = np.abs(M - o_M).max(), np.abs(cov - o_cov).max()
    assert np.array_equal(M, o_M) and np.array_equal(cov, o_cov), f"max |dM| {dM:.3e}, max |dcov| {dC:.3e}"
    # NULL lastFrame / isCurrent: every point passes the frame test and counts as current
    o_M2, o_cov2 = sc.map0.copy(), sc.cov0.copy()
    n2, ns2, nd2, _ = oracle.update_new_poses_points([sc.K] * nC, [sc.iK] * nC, np.stack([np.stack(h["R"]) for h in hist]),
                                                     np.stack([np.stack(h["t"]) for h in hist]), np.stack([np.stack(h["xy"]) for h in hist]),
                                                     span, np.stack(o_stat), pf, o_M2, o_cov2, sc.flags0, SIGMA)
    d_M2, d_cov2 = torch.from_numpy(sc.map0.copy()).to(dev), torch.from_numpy(sc.cov0.copy()).to(dev)
    th.update_new_poses_points_dev(s, cams, d_pf.data_ptr(), nMap, d_M2.data_ptr(), d_cov2.data_ptr(), d_fl.data_ptr(), SIGMA,
                                   d_counts=d_cnt.data_ptr())
    torch.cuda.synchronize()
    assert d_cnt.tolist() == [ns2, nd2] and ns2 >= ns and nd2 >= nd
    assert np.array_equal(d_M2.cpu().numpy(), o_M2) and np.array_equal(d_cov2.cpu().numpy(), o_cov2)
    th.close()


def test_map_points_classify_reproduces_the_reference_on_its_golden_scenes():
    """cs_map_points_classify_dev against tests/golden/classify_golden.npz -- what the reference's own CoSLAM::mapPointsClassify
    (src/app/SL_CoSLAM.cpp:418-520) with its helpers (src/slam/SL_CoSLAMHelper.cpp:67-330), compiled in place, left of three scenes
    of 72 map points that walk every branch of the state machine: positions and covariances bit for bit, types, uncertain flags,
    bNewPt, staticFrameNum, the detached features (table and slot2map), the features' types."""
    import os

    import torch

    from coslam_amd.poseupdate import TrackHistory

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "classify_golden.npz"))
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    examined = 0
    for sc in range(int(g["n_scenes"])):
        G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
        hR, hT, hXY, span = G("histR"), G("histT"), G("histXY"), G("trackSpan")
        nC, H = hR.shape[0], hR.shape[1]
        N = hXY.shape[2] // 2
        cur, nMap = int(G("curFrame")), G("M0").shape[0]
        th = TrackHistory(nC, N, H)
        d_K = torch.from_numpy(G("K").copy()).to(dev)
        d_iK = torch.from_numpy(G("iK").copy()).to(dev)
        d_fl0 = torch.zeros(nMap, dtype=torch.uint8, device=dev)
        d_span = torch.from_numpy(span.copy()).to(dev)
        d_scratch = torch.ones((nC, N), dtype=torch.uint8, device=dev)
        d_none = torch.full((nC, N), -1, dtype=torch.int32, device=dev)
        keep = []
        for j in range(H - 1, -1, -1):   # oldest first; entry j = frame cur - j, with that frame's poses
            xy = torch.from_numpy(hXY[:, j].copy()).to(dev)
            st = torch.zeros((nC, N), dtype=torch.int32, device=dev)
            Rj, tj = torch.from_numpy(hR[:, j].copy()).to(dev), torch.from_numpy(hT[:, j].copy()).to(dev)
            keep += [xy, st, Rj, tj]
            cams = [dict(K=d_K[c].data_ptr(), iK=d_iK[c].data_ptr(), xy=xy[c].data_ptr(), state=st[c].data_ptr(),
                         slot2map=d_none[c].data_ptr(), trackSpan=d_span[c].data_ptr(), isStatic=d_scratch[c].data_ptr()) for c in range(nC)]
            th.detect_dynamic_dev(s, cams, Rj.data_ptr(), tj.data_ptr(), nMap, d_fl0.data_ptr(), cur - j, minLen=1 << 30)
        assert th.frames == H
        pf = G("pointFeat")
        s2m = np.where(pf.T >= 0, np.arange(N, dtype=np.int32)[None, :], -1).astype(np.int32)   # slot = point index in these scenes
        d_s2m = torch.from_numpy(np.ascontiguousarray(s2m)).to(dev)
        d_fstat = torch.from_numpy(G("featStatic").copy()).to(dev)
        d_M, d_cov = torch.from_numpy(G("M0").copy()).to(dev), torch.from_numpy(G("cov0").copy()).to(dev)
        d_fl, d_new = torch.from_numpy(G("flags").copy()).to(dev), torch.from_numpy(G("newPt").copy()).to(dev)
        d_sfn, d_first = torch.from_numpy(G("staticFrameNum").copy()).to(dev), torch.from_numpy(G("firstFrame").copy()).to(dev)
        d_pf = torch.from_numpy(pf.copy()).to(dev)
        d_ff, d_f1 = torch.from_numpy(G("featFrame").copy()).to(dev), torch.from_numpy(G("featFirst").copy()).to(dev)
        d_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
        cams = [dict(K=d_K[c].data_ptr(), iK=d_iK[c].data_ptr(), trackSpan=d_span[c].data_ptr(), isStatic=d_fstat[c].data_ptr(),
                     slot2map=d_s2m[c].data_ptr()) for c in range(nC)]
        th.map_points_classify_dev(s, cams, d_pf.data_ptr(), nMap, cur, d_M.data_ptr(), d_cov.data_ptr(), d_fl.data_ptr(), d_new.data_ptr(),
                                   d_sfn.data_ptr(), d_first.data_ptr(), float(G("pixelVar")), d_featFrame=d_ff.data_ptr(),
                                   d_featFirst=d_f1.data_ptr(), d_counts=d_cnt.data_ptr())
        torch.cuda.synchronize()
        M, cov, fl = d_M.cpu().numpy(), d_cov.cpu().numpy(), d_fl.cpu().numpy()
        assert np.array_equal(fl, G("flags_ref")), f"scene {sc}: types differ at {np.nonzero(fl != G('flags_ref'))[0][:8]}"
        assert np.array_equal(d_new.cpu().numpy(), G("newPt_ref")) and np.array_equal(d_sfn.cpu().numpy(), G("staticFrameNum_ref"))
        dM, dC = np.abs(M - G("M_ref")).max(), np.abs(cov - G("cov_ref")).max()
        assert np.array_equal(M, G("M_ref")) and np.array_equal(cov, G("cov_ref")), f"scene {sc}: max |dM| {dM:.3e}, max |dcov| {dC:.3e}"
        pf_out = d_pf.cpu().numpy()
        assert np.array_equal((pf_out >= 0).astype(np.uint8), G("hasFeature_ref"))
        assert np.array_equal(d_s2m.cpu().numpy() >= 0, pf_out.T >= 0)
        assert np.array_equal(d_fstat.cpu().numpy(), G("featStatic_ref"))
        cnt = d_cnt.tolist()
        assert cnt[1] == int((((fl & 2) != 0) & ((G("flags") & 2) == 0)).sum()) and cnt[0] > 50
        examined += cnt[0]
        with pytest.raises(Exception):   # the history's newest entry is not the frame that is asked for
            th.map_points_classify_dev(s, cams, d_pf.data_ptr(), nMap, cur + 1, d_M.data_ptr(), d_cov.data_ptr(), d_fl.data_ptr(),
                                       d_new.data_ptr(), d_sfn.data_ptr(), d_first.data_ptr())
        # nothing to examine (every point certain and static, or false): an empty worklist, not one byte of the map changes
        d_fl2 = torch.from_numpy(np.where(np.arange(nMap) % 5 == 0, 2, 0).astype(np.uint8)).to(dev)
        before = [x.clone() for x in (d_M, d_cov, d_new, d_sfn, d_pf, d_s2m, d_fstat)]
        th.map_points_classify_dev(s, cams, d_pf.data_ptr(), nMap, cur, d_M.data_ptr(), d_cov.data_ptr(), d_fl2.data_ptr(), d_new.data_ptr(),
                                   d_sfn.data_ptr(), d_first.data_ptr(), float(G("pixelVar")), d_featFrame=d_ff.data_ptr(),
                                   d_featFirst=d_f1.data_ptr(), d_counts=d_cnt.data_ptr())
        torch.cuda.synchronize()
        assert d_cnt.tolist() == [0, 0] and all(torch.equal(a, b) for a, b in zip(before, (d_M, d_cov, d_new, d_sfn, d_pf, d_s2m, d_fstat)))
        assert np.array_equal(d_fl2.cpu().numpy(), np.where(np.arange(nMap) % 5 == 0, 2, 0))
        th.close()
    assert examined > 150


def _relink_ring(g, sc, dev, s):
    """the history of scene sc of tests/golden/update_points_relink_golden.npz: every segment's pixels on its own slot, the ring filled frame
    by frame (slots dead outside their segments), the scene's poses, the pool of linked segments loaded"""
    import torch

    from coslam_amd.poseupdate import TrackHistory

    G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
    hR, hT, hXY, ref, pool = G("histR"), G("histT"), np.nan_to_num(G("histXY"), nan=-1e9), G("featRef"), G("segPool")
    nC, H = hR.shape[0], hR.shape[1]
    N = hXY.shape[2] // 2
    cur, nMap = int(G("curFrame")), G("M0").shape[0]
    # first | last frame of every slot's segment (what trackSpan would have held while the segment's track was alive)
    span = np.full((nC, 2 * N), -1, np.int32)
    for c in range(nC):
        for sl, fr, fi, _ in ref[:, c]:
            if sl >= 0:
                span[c, sl], span[c, N + sl] = fi, fr
        for sl, la, fi, _ in pool[c]:
            if sl >= 0:
                span[c, sl], span[c, N + sl] = fi, la
    th = TrackHistory(nC, N, 64, storeLen=H + 3)   # walks of 64 nodes over a store deeper than the history
    keep = dict(K=torch.from_numpy(G("K").copy()).to(dev), iK=torch.from_numpy(G("iK").copy()).to(dev), fl=torch.from_numpy(G("flags").copy()).to(dev),
                span=torch.from_numpy(span).to(dev), fstat=torch.from_numpy(G("featStatic").copy()).to(dev),
                scratch=torch.ones((nC, N), dtype=torch.uint8, device=dev), s2m=torch.full((nC, N), -1, dtype=torch.int32, device=dev), misc=[])
    eye = torch.from_numpy(np.tile(np.eye(3).reshape(9), (nC, 1))).to(dev)
    zero = torch.zeros((nC, 3), dtype=torch.float64, device=dev)
    for j in range(H - 1, -1, -1):
        f = cur - j
        xy = torch.from_numpy(hXY[:, j].copy()).to(dev)
        st = torch.from_numpy(((span[:, :N] >= 0) & (span[:, :N] <= f) & (f <= span[:, N:])).astype(np.int32) - 1).to(dev)
        keep["misc"] += [xy, st]
        cams = [dict(K=keep["K"][c].data_ptr(), iK=keep["iK"][c].data_ptr(), xy=xy[c].data_ptr(), state=st[c].data_ptr(),
                     slot2map=keep["s2m"][c].data_ptr(), trackSpan=keep["span"][c].data_ptr(), isStatic=keep["scratch"][c].data_ptr()) for c in range(nC)]
        th.detect_dynamic_dev(s, cams, eye.data_ptr(), zero.data_ptr(), nMap, keep["fl"].data_ptr(), f, minLen=1 << 30)
    cam_i = np.repeat(np.arange(nC), H).astype(np.int32)
    frm_i = np.tile(cur - np.arange(H), nC).astype(np.int32)
    d = [torch.from_numpy(a).to(dev) for a in (cam_i, frm_i, hR.reshape(-1, 9).copy(), hT.reshape(-1, 3).copy())]
    th.set_poses_dev(s, len(cam_i), *[x.data_ptr() for x in d])
    torch.cuda.synchronize()
    th.load_segments(pool)
    keep["misc"] += d + [eye, zero]
    cams = [dict(K=keep["K"][c].data_ptr(), iK=keep["iK"][c].data_ptr(), trackSpan=keep["span"][c].data_ptr(), isStatic=keep["fstat"][c].data_ptr())
            for c in range(nC)]
    return th, cams, keep


def test_map_points_classify_over_feature_references_reproduces_the_reference():
    """cs_map_points_classify_dev behind cs_track_history_set_classify_refs against tests/golden/classify_relink_golden.npz -- the
    reference's own CoSLAM::mapPointsClassify over chains as the registration loops and lost tracks leave them (748 chains, 64 stale
    heads, 516 with linked segments; 70 frames of history, isStaticPoint's window of 60 ends inside chains): positions and covariances
    bit for bit, types, flags, counters, which features / references stay, the features' types -- with the table at this frame and
    with the live references one frame behind (as the frame loop holds them when the classification runs)."""
    import os

    import torch

    from tests.test_oracle_cpu import classify_relink_expect, shifted_refs

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "classify_relink_golden.npz"))
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    examined = stale_detached = 0
    for sc in range(int(g["n_scenes"])):
        G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
        cur, nMap = int(G("curFrame")), G("M0").shape[0]
        for behind in (False, True):
            th, cams, keep = _relink_ring(g, sc, dev, s)
            nC = len(cams)
            ref0 = shifted_refs(G)[0] if behind else G("featRef").copy()
            d_ref, d_rstat = torch.from_numpy(ref0.copy()).to(dev), torch.from_numpy(G("refStatic").copy()).to(dev)
            d_s2m = torch.from_numpy(G("slot2map").copy()).to(dev)
            for c in range(nC):
                cams[c]["slot2map"] = d_s2m[c].data_ptr()
            d_M, d_cov = torch.from_numpy(G("M0").copy()).to(dev), torch.from_numpy(G("cov0").copy()).to(dev)
            d_fl, d_new = torch.from_numpy(G("flags").copy()).to(dev), torch.from_numpy(G("newPt").copy()).to(dev)
            d_sfn, d_first = torch.from_numpy(G("staticFrameNum").copy()).to(dev), torch.from_numpy(G("firstFrame").copy()).to(dev)
            d_pf = torch.from_numpy(G("pointFeat").copy()).to(dev)
            d_cnt = torch.zeros(2, dtype=torch.int32, device=dev)
            th.set_classify_refs(d_ref.data_ptr(), d_rstat.data_ptr())
            th.map_points_classify_dev(s, cams, d_pf.data_ptr(), nMap, cur, d_M.data_ptr(), d_cov.data_ptr(), d_fl.data_ptr(), d_new.data_ptr(),
                                       d_sfn.data_ptr(), d_first.data_ptr(), float(G("pixelVar")), d_counts=d_cnt.data_ptr())
            torch.cuda.synchronize()
            M, cov, fl = d_M.cpu().numpy(), d_cov.cpu().numpy(), d_fl.cpu().numpy()
            assert np.array_equal(fl, G("flags_ref")), f"scene {sc} behind {behind}: types differ at {np.nonzero(fl != G('flags_ref'))[0][:8]}"
            assert np.array_equal(d_new.cpu().numpy(), G("newPt_ref")) and np.array_equal(d_sfn.cpu().numpy(), G("staticFrameNum_ref"))
            dM, dC = np.abs(M - G("M_ref")).max(), np.abs(cov - G("cov_ref")).max()
            assert np.array_equal(M, G("M_ref")) and np.array_equal(cov, G("cov_ref")), f"scene {sc}: max |dM| {dM:.3e}, max |dcov| {dC:.3e}"
            pf, ref = d_pf.cpu().numpy(), d_ref.cpu().numpy()
            has, dyn = classify_relink_expect(G, pf, ref, keep["fstat"].cpu().numpy(), d_rstat.cpu().numpy())
            assert np.array_equal(has, G("hasFeature_ref")) and np.array_equal(dyn, G("featDyn_ref") * has)
            gone = (ref0[:, :, 0] >= 0) & (ref[:, :, 0] < 0)
            assert np.array_equal(gone, (G("hasFeature_ref") == 0) & (ref0[:, :, 0] >= 0))
            assert np.array_equal(ref[~gone], ref0[~gone])                                   # every other reference stands as it was
            assert int((d_s2m.cpu().numpy() >= 0).sum()) == int((pf >= 0).sum())
            cnt = d_cnt.tolist()
            assert cnt[1] == int((((fl & 2) != 0) & ((G("flags") & 2) == 0)).sum()) and cnt[0] > 50
            if not behind:
                examined += cnt[0]
                stale_detached += int((gone & (G("pointFeat") < 0)).sum())
            # without the references the same call sees this frame's features only: some points come out differently
            if behind:
                th.set_classify_refs(None)
                d_M2, d_cov2, d_fl2 = torch.from_numpy(G("M0").copy()).to(dev), torch.from_numpy(G("cov0").copy()).to(dev), torch.from_numpy(G("flags").copy()).to(dev)
                d_new2, d_sfn2, d_pf2 = torch.from_numpy(G("newPt").copy()).to(dev), torch.from_numpy(G("staticFrameNum").copy()).to(dev), torch.from_numpy(G("pointFeat").copy()).to(dev)
                th.map_points_classify_dev(s, cams, d_pf2.data_ptr(), nMap, cur, d_M2.data_ptr(), d_cov2.data_ptr(), d_fl2.data_ptr(), d_new2.data_ptr(),
                                           d_sfn2.data_ptr(), d_first.data_ptr(), float(G("pixelVar")))
                torch.cuda.synchronize()
                assert not np.array_equal(d_fl2.cpu().numpy(), G("flags_ref")) or not np.array_equal(d_M2.cpu().numpy(), G("M_ref"))
            th.close()
    assert examined > 150 and stale_detached >= 1


def test_relinked_and_stale_feature_chains_reproduce_the_reference():
    """cs_update_new_poses_points_ref_dev / cs_refine_map_points_ref_dev / cs_check_unify_ref_dev against
    tests/golden/update_points_relink_golden.npz (VERDICT r04 missing 3): chains as `pFeat->preFrame = p->pFeatures[iCam]`
    (reference src/app/SL_CoSLAM.cpp:775-779, :997-1000) and lost tracks leave them -- 618 chains, 188 with a stale head, 462 with older
    segments linked behind, put through the reference's OWN updateNewPosesPoints, refineMapPoint and checkUnify compiled in place.  Every
    point, covariance and verdict bit for bit, with walks of at most 64 nodes over a 43 / 73-frame store (the golden chains are shorter
    than 64 nodes: the bound does not bite)."""
    import os

    import torch

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "update_points_relink_golden.npz"))
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)   # noqa: E731
    n_upd = n_uni = 0
    for sc in range(int(g["n_scenes"])):
        G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
        th, cams, keep = _relink_ring(g, sc, dev, s)
        ref = G("featRef")
        nMap, nC = ref.shape[0], ref.shape[1]
        d_ref = d(ref)
        d_M, d_cov = d(G("M0")), d(G("cov0"))
        d_lf, d_ic, d_cnt = d(G("lastFrame")), d(G("isCurrent")), torch.zeros(2, dtype=torch.int32, device=dev)
        th.update_new_poses_points_ref_dev(s, cams, d_ref.data_ptr(), nMap, d_M.data_ptr(), d_cov.data_ptr(), keep["fl"].data_ptr(), float(G("sigma")),
                                           d_lastFrame=d_lf.data_ptr(), d_isCurrent=d_ic.data_ptr(), firstKeyFrame=int(G("firstKey")),
                                           d_counts=d_cnt.data_ptr())
        torch.cuda.synchronize()
        M, cov = d_M.cpu().numpy(), d_cov.cpu().numpy()
        n_ref = int((G("M_ref") != G("M0")).any(axis=1).sum())
        assert int(d_cnt.sum()) == n_ref, (sc, d_cnt.tolist(), n_ref)
        assert np.array_equal(M, G("M_ref")) and np.array_equal(cov, G("cov_ref")), f"scene {sc}: max |dM| {np.abs(M - G('M_ref')).max():.3e}"
        n_upd += n_ref
        d_M2, d_cov2, d_sel, d_n = d(G("M0")), d(G("cov0")), d(G("refine_select")), torch.zeros(1, dtype=torch.int32, device=dev)
        th.refine_map_points_ref_dev(s, cams, d_ref.data_ptr(), nMap, d_M2.data_ptr(), d_cov2.data_ptr(), float(G("sigma")), d_select=d_sel.data_ptr(),
                                     d_count=d_n.data_ptr())
        torch.cuda.synchronize()
        assert int(d_n.item()) == int(G("refine_select").sum())
        assert np.array_equal(d_M2.cpu().numpy(), G("M_refine")) and np.array_equal(d_cov2.cpu().numpy(), G("cov_refine")), sc
        # ... and as ONE launch with the references' advance in front (cs_feat_ref_advance_refine_dev): on tables that are already at this
        # frame the advance changes nothing -- live heads are "tracked on", stale ones stay -- so the refined points are the reference's again;
        # the marks are consumed when asked
        cur = int(G("curFrame"))
        pf = np.where(ref[:, :, 1] == cur, ref[:, :, 0], -1).astype(np.int32)
        for clear in (False, True):
            d_M3, d_cov3, d_sel3, d_ref3, d_pf3 = d(G("M0")), d(G("cov0")), d(G("refine_select")), d(ref), d(pf)
            d_list = torch.arange(nMap, dtype=torch.int32, device=dev)
            d_fc = torch.zeros(5, dtype=torch.int32, device=dev)
            for c in range(nC):
                cams[c]["trackSpan"] = keep["span"][c].data_ptr()
            th.feat_ref_advance_refine_dev(s, cams, nMap, d_pf3.data_ptr(), cur, d_ref3.data_ptr(), 0, d_list.data_ptr(), nMap, True, d_sel3.data_ptr(), clear,
                                           d_M3.data_ptr(), d_cov3.data_ptr(), float(G("sigma")), d_counts=d_fc.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(d_M3.cpu().numpy(), G("M_refine")) and np.array_equal(d_cov3.cpu().numpy(), G("cov_refine")), (sc, clear)
            assert np.array_equal(d_ref3.cpu().numpy(), ref) and d_fc.tolist() == [0, 0, 0, 0, 0]
            assert int(d_sel3.sum().item()) == (0 if clear else int(G("refine_select").sum()))
        nP = len(G("unify_ok"))
        a = np.stack([np.where(G("unify_has1")[q][:, None] > 0, ref[G("unify_pts")[q][0]], -1) for q in range(nP)]).astype(np.int32)
        b = np.stack([np.where(G("unify_has2")[q][:, None] > 0, ref[G("unify_pts")[q][1]], -1) for q in range(nP)]).astype(np.int32)
        d_a, d_b, d_M1, d_M2u = d(a), d(b), d(G("unify_M1")), d(G("unify_M2"))
        d_ok = torch.zeros(nP, dtype=torch.uint8, device=dev)
        d_Mu, d_covu = torch.zeros((nP, 3), dtype=torch.float64, device=dev), torch.zeros((nP, 9), dtype=torch.float64, device=dev)
        th.check_unify_ref_dev(s, cams, nP, d_a.data_ptr(), d_b.data_ptr(), d_M1.data_ptr(), d_M2u.data_ptr(), float(G("sigma")), d_ok.data_ptr(),
                               d_Mu.data_ptr(), d_covu.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_ok.cpu().numpy().astype(bool), G("unify_ok").astype(bool)), sc
        assert np.array_equal(d_Mu.cpu().numpy(), G("unify_M")) and np.array_equal(d_covu.cpu().numpy(), G("unify_cov"), equal_nan=True), sc
        n_uni += nP
        th.close()
    assert n_upd > 150 and n_uni > 350


def test_feature_references_follow_the_registration_like_the_references_pointers():
    """cs_feat_ref_advance_dev on a scripted sequence of one camera's pointFeat column: a point tracked on (the frame moves, first and
    segment stay), lost (the reference stays: stale), registered to a NEW track while the old reference is still there (the old one
    becomes a pool segment, the new one starts at this frame: reference src/app/SL_CoSLAM.cpp:775-779), re-registered a second time
    (the new segment names the older one), a first feature (the track's first frame, no segment), and a feature detached by the
    classification (its slot's track lives on without the point: cleared; the next attachment is a first feature again) next to a track
    that simply ended (stale).  Counters and pool contents are checked."""
    import torch

    from coslam_amd.poseupdate import TrackHistory

    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    nC, N, nMap = 2, 16, 4
    th = TrackHistory(nC, N, 8, storeLen=32)
    span = np.full((nC, 2 * N), -1, np.int32)
    d_span = torch.from_numpy(span).to(dev)
    d_stat = torch.ones((nC, N), dtype=torch.uint8, device=dev)
    cams = [dict(trackSpan=d_span[c].data_ptr(), isStatic=d_stat[c].data_ptr()) for c in range(nC)]
    d_ref = torch.full((nMap, nC, 4), -1, dtype=torch.int32, device=dev)
    d_rs = torch.zeros((nMap, nC), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(5, dtype=torch.int32, device=dev)

    def step(frame, pf, first_of, dyn=()):
        sp = span.copy()
        for (c, sl), f1 in first_of.items():
            sp[c, sl], sp[c, N + sl] = f1, frame
        d_span.copy_(torch.from_numpy(sp))
        st = np.ones((nC, N), np.uint8)
        for c, sl in dyn:
            st[c, sl] = 0
        d_stat.copy_(torch.from_numpy(st))
        d_pf = torch.from_numpy(np.asarray(pf, np.int32)).to(dev)
        th.feat_ref_advance_dev(s, cams, nMap, d_pf.data_ptr(), frame, d_ref.data_ptr(), d_refStatic=d_rs.data_ptr(), d_counts=d_cnt.data_ptr())
        torch.cuda.synchronize()
        return d_ref.cpu().numpy().copy()

    none = [[-1, -1]] * nMap
    # frame 10: point 0 on slot 3 of camera 0 (track since frame 4), point 1 on slot 5 of camera 1 (dynamic feature)
    r = step(10, [[3, -1], [-1, 5], [-1, -1], [-1, -1]], {(0, 3): 4, (1, 5): 10}, dyn=[(1, 5)])
    assert r[0, 0].tolist() == [3, 10, 4, -1] and r[1, 1].tolist() == [5, 10, 10, -1] and r[2, 0, 0] == -1
    assert d_rs.cpu().numpy()[0, 0] == 1 and d_rs.cpu().numpy()[1, 1] == 0
    # frame 11: both tracked on; frame 12: camera 0 lost point 0 (stale), point 1 tracked on
    r = step(11, [[3, -1], [-1, 5], [-1, -1], [-1, -1]], {(0, 3): 4, (1, 5): 10})
    assert r[0, 0].tolist() == [3, 11, 4, -1] and r[1, 1].tolist() == [5, 11, 10, -1]
    r = step(12, [[-1, -1], [-1, 5], [-1, -1], [-1, -1]], {(1, 5): 10})
    assert r[0, 0].tolist() == [3, 11, 4, -1]
    # frame 15: point 0 registered to the track on slot 7 (alive since frame 13): re-linked behind the stale reference
    r = step(15, [[7, -1], [-1, 5], [-1, -1], [-1, -1]], {(0, 7): 13, (1, 5): 10})
    assert r[0, 0].tolist() == [7, 15, 15, 0]
    r = step(16, [[7, -1], [-1, 5], [-1, -1], [-1, -1]], {(0, 7): 13, (1, 5): 10})
    assert r[0, 0].tolist() == [7, 16, 15, 0]
    # frame 20: lost again at 17, now registered to slot 3 -- the SAME slot number as the first track, a new track (since frame 18)
    r = step(20, [[3, -1], [-1, -1], [-1, -1], [-1, -1]], {(0, 3): 18})
    assert r[0, 0].tolist() == [3, 20, 20, 1] and r[1, 1].tolist() == [5, 16, 10, -1]
    # frame 21: the classification detached point 0's feature in camera 0 -- slot 3's track lives on (last frame 21), pointFeat no longer
    # names it; frame 22: attached again = a first feature
    r = step(21, none, {(0, 3): 18})
    assert r[0, 0, 0] == -1 and r[1, 1].tolist() == [5, 16, 10, -1]   # (a stale reference is not "detached")
    r = step(22, [[3, -1], [-1, -1], [-1, -1], [-1, -1]], {(0, 3): 18})
    assert r[0, 0].tolist() == [3, 22, 18, -1]
    cnt, cap = th.segment_counts()
    assert cnt.tolist() == [2, 0] and cap >= 1024
    # tracked on, first, re-linked, dropped, detached
    assert d_cnt.cpu().numpy().tolist() == [6, 3, 2, 0, 1], d_cnt.cpu().numpy().tolist()
    th.close()


def _golden_ring(g, sc, dev, s):
    """the history of golden scene sc: the ring filled frame by frame with placeholder poses, then given the scene's poses (the way
    test_update_new_poses_points_reproduces_the_reference_on_its_golden_scenes builds it)"""
    import torch

    from coslam_amd.poseupdate import TrackHistory

    G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
    hR, hT, hXY, span = G("histR"), G("histT"), G("histXY"), G("trackSpan")
    nC, H = hR.shape[0], hR.shape[1]
    N = hXY.shape[2] // 2
    cur, nMap = int(G("curFrame")), G("M0").shape[0]
    th = TrackHistory(nC, N, H + 3)
    keep = dict(K=torch.from_numpy(G("K").copy()).to(dev), iK=torch.from_numpy(G("iK").copy()).to(dev), fl=torch.from_numpy(G("flags").copy()).to(dev),
                span=torch.from_numpy(span.copy()).to(dev), scratch=torch.ones((nC, N), dtype=torch.uint8, device=dev),
                s2m=torch.full((nC, N), -1, dtype=torch.int32, device=dev), misc=[])
    eye = torch.from_numpy(np.tile(np.eye(3).reshape(9), (nC, 1))).to(dev)
    zero = torch.zeros((nC, 3), dtype=torch.float64, device=dev)
    for j in range(H - 1, -1, -1):
        xy = torch.from_numpy(hXY[:, j].copy()).to(dev)
        st = torch.from_numpy(((span[:, :N] >= 0) & (span[:, :N] <= cur - j)).astype(np.int32) - 1).to(dev)
        keep["misc"] += [xy, st]
        cams = [dict(K=keep["K"][c].data_ptr(), iK=keep["iK"][c].data_ptr(), xy=xy[c].data_ptr(), state=st[c].data_ptr(),
                     slot2map=keep["s2m"][c].data_ptr(), trackSpan=keep["span"][c].data_ptr(), isStatic=keep["scratch"][c].data_ptr()) for c in range(nC)]
        th.detect_dynamic_dev(s, cams, eye.data_ptr(), zero.data_ptr(), nMap, keep["fl"].data_ptr(), cur - j, minLen=1 << 30)
    cam_i = np.repeat(np.arange(nC), H).astype(np.int32)
    frm_i = np.tile(cur - np.arange(H), nC).astype(np.int32)
    d = [torch.from_numpy(a).to(dev) for a in (cam_i, frm_i, hR.reshape(-1, 9).copy(), hT.reshape(-1, 3).copy())]
    th.set_poses_dev(s, len(cam_i), *[x.data_ptr() for x in d])
    keep["misc"] += d + [eye, zero]
    cams = [dict(K=keep["K"][c].data_ptr(), iK=keep["iK"][c].data_ptr(), trackSpan=keep["span"][c].data_ptr(), isStatic=keep["scratch"][c].data_ptr())
            for c in range(nC)]
    return th, cams, keep


def test_check_unify_reproduces_the_reference_on_its_golden_pairs():
    """cs_check_unify_dev against tests/golden/update_points_golden.npz: 594 pairs of temporary map points the reference's own
    CoSLAM::checkUnify (src/app/SL_CoSLAM.cpp:561-665, compiled in place) judged -- the two halves of one point's cameras and two
    different points: verdict, unified point and covariance bit for bit, the gate's `Rs + 3 * i` (:657) as written, NaN covariances of
    the rotation-only rig included."""
    import os

    import torch

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "update_points_golden.npz"))
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    n_all = n_yes = 0
    for sc in range(int(g["n_scenes"])):
        G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
        nP = len(G("unify_ok"))
        if nP == 0:
            continue
        th, cams, keep = _golden_ring(g, sc, dev, s)
        pf = G("pointFeat")
        a = np.stack([np.where(G("unify_has1")[q] > 0, pf[G("unify_pts")[q][0]], -1) for q in range(nP)]).astype(np.int32)
        b = np.stack([np.where(G("unify_has2")[q] > 0, pf[G("unify_pts")[q][1]], -1) for q in range(nP)]).astype(np.int32)
        d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)   # noqa: E731
        d_a, d_b, d_M1, d_M2 = d(a), d(b), d(G("unify_M1")), d(G("unify_M2"))
        d_ok = torch.zeros(nP, dtype=torch.uint8, device=dev)
        d_M, d_cov = torch.zeros((nP, 3), dtype=torch.float64, device=dev), torch.zeros((nP, 9), dtype=torch.float64, device=dev)
        th.check_unify_dev(s, cams, nP, d_a.data_ptr(), d_b.data_ptr(), d_M1.data_ptr(), d_M2.data_ptr(), float(G("sigma")), d_ok.data_ptr(),
                           d_M.data_ptr(), d_cov.data_ptr())
        torch.cuda.synchronize()
        ok, M, cov = d_ok.cpu().numpy(), d_M.cpu().numpy(), d_cov.cpu().numpy()
        assert np.array_equal(ok.astype(bool), G("unify_ok").astype(bool)), (sc, np.nonzero(ok.astype(bool) != G("unify_ok").astype(bool))[0][:5])
        assert np.array_equal(M, G("unify_M")), sc
        assert np.array_equal(cov, G("unify_cov"), equal_nan=True), sc
        n_all += nP
        n_yes += int(ok.sum())
    assert n_all > 500 and 200 < n_yes < n_all - 200


def test_intracam_new_map_points_reproduce_the_reference():
    """cs_newpts_intracam_dev against tests/golden/intracam_newpts_golden.npz: the reference's own SingleSLAM::newMapPoints (reference
    src/app/SL_SingleSLAM.cpp:922-1004, compiled in place) on three one-camera scenes.  The ring is fed frame by frame (walks 64 frames
    deep, 43-73 frames kept), the kernel appends behind a map that already holds 11 points: the same points in the same order -- position
    and covariance bit for bit, first frame, TYPE_MAP_STATIC, bNewPt, the feature attached in pointFeat and slot2map --, the old map
    untouched; a map with room for only 5 more drops the rest and says so; a camera that is not ready (d_ready below readyMin) adds nothing."""
    import os

    import torch

    from coslam_amd.poseupdate import TrackHistory

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "intracam_newpts_golden.npz"))
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)   # noqa: E731
    total = 0
    for sc in range(int(g["n_scenes"])):
        G = lambda k: g[f"s{sc}_{k}"]   # noqa: E731
        hR, hT, hXY, span = G("histR"), G("histT"), G("histXY"), G("trackSpan")
        H, N, cur = hR.shape[0], len(G("state")), int(G("curFrame"))
        th = TrackHistory(1, N, 64, storeLen=H + 3)
        dK, diK, d_span, d_fs = d(G("K")), d(G("iK")), d(span[None]), d(G("isStatic")[None])
        d_fl0 = torch.zeros(64, dtype=torch.uint8, device=dev)
        d_scr = torch.ones((1, N), dtype=torch.uint8, device=dev)
        d_s2m_feed = torch.full((1, N), -1, dtype=torch.int32, device=dev)
        eye, zero = d(np.eye(3).reshape(1, 9)), torch.zeros((1, 3), dtype=torch.float64, device=dev)
        keep = []
        for j in range(H - 1, -1, -1):
            f = cur - j
            xy = d(hXY[j][None])
            st = d((((span[:N] >= 0) & (span[:N] <= f)).astype(np.int32) - 1)[None])
            keep += [xy, st]
            th.detect_dynamic_dev(s, [dict(K=dK.data_ptr(), iK=diK.data_ptr(), xy=xy.data_ptr(), state=st.data_ptr(), slot2map=d_s2m_feed.data_ptr(),
                                           trackSpan=d_span.data_ptr(), isStatic=d_scr.data_ptr())], eye.data_ptr(), zero.data_ptr(), 64,
                                  d_fl0.data_ptr(), f, minLen=1 << 30)
        pose = [d(np.zeros(H, np.int32)), d((cur - np.arange(H)).astype(np.int32)), d(hR), d(hT)]
        th.set_poses_dev(s, H, *[x.data_ptr() for x in pose])
        n_ref = len(G("new_slot"))
        for cap_extra, ready in ((4096, None), (5, None), (4096, 1)):
            base = 11
            cap = base + cap_extra
            d_M, d_cov = torch.full((cap, 3), 7.0, dtype=torch.float64, device=dev), torch.full((cap, 9), 7.0, dtype=torch.float64, device=dev)
            d_fl, d_np = torch.full((cap,), 9, dtype=torch.uint8, device=dev), torch.full((cap,), 9, dtype=torch.uint8, device=dev)
            d_ff, d_pf = torch.full((cap,), -5, dtype=torch.int32, device=dev), torch.full((cap, 1), -9, dtype=torch.int32, device=dev)
            d_cnt, d_mc = torch.zeros(3, dtype=torch.int32, device=dev), torch.tensor([base], dtype=torch.int32, device=dev)
            d_st, d_s2m = d(G("state")[None]), d(G("slot2map")[None])
            d_scratch = torch.zeros(th.newpts_intracam_scratch_bytes(), dtype=torch.uint8, device=dev)
            d_ready = None if ready is None else torch.tensor([ready], dtype=torch.int32, device=dev)
            cams = [dict(K=dK.data_ptr(), iK=diK.data_ptr(), state=d_st.data_ptr(), slot2map=d_s2m.data_ptr(), trackSpan=d_span.data_ptr(),
                         isStatic=d_fs.data_ptr())]
            th.newpts_intracam_dev(s, cams, d_M.data_ptr(), d_cov.data_ptr(), d_fl.data_ptr(), d_np.data_ptr(), d_ff.data_ptr(), d_pf.data_ptr(), cap,
                                   d_mc.data_ptr(), d_scratch.data_ptr(), float(G("sigma")), d_ready=None if d_ready is None else d_ready.data_ptr(),
                                   readyMin=2, minTrackLen=int(G("minTrackLen")), maxWalk=1024, maxEpiErr=float(G("maxEpiErr")), d_counts=d_cnt.data_ptr())
            torch.cuda.synchronize()
            want = 0 if ready is not None else min(n_ref, cap_extra)
            assert int(d_mc.item()) == base + want, (sc, cap_extra, ready, int(d_mc.item()))
            cnt = d_cnt.cpu().numpy().tolist()
            assert cnt[1] == want and cnt[2] == (0 if ready is not None else n_ref - want)
            M, cov = d_M.cpu().numpy(), d_cov.cpu().numpy()
            assert (M[:base] == 7.0).all() and (d_fl.cpu().numpy()[:base] == 9).all() and (M[base + want:] == 7.0).all()
            assert np.array_equal(M[base:base + want], G("new_M")[:want]) and np.array_equal(cov[base:base + want], G("new_cov")[:want]), (sc, cap_extra)
            assert np.array_equal(d_ff.cpu().numpy()[base:base + want], G("new_first")[:want])
            assert (d_fl.cpu().numpy()[base:base + want] == 0).all() and (d_np.cpu().numpy()[base:base + want] == 1).all()
            assert np.array_equal(d_pf.cpu().numpy()[base:base + want, 0], G("new_slot")[:want])
            s2m = d_s2m.cpu().numpy()[0]
            exp = G("slot2map").copy()
            exp[G("new_slot")[:want]] = base + np.arange(want)
            assert np.array_equal(s2m, exp)
            if ready is None and cap_extra > 100:
                total += want
        th.close()
    assert total > 150


def test_pose_update_and_classification_as_two_launches_equal_the_two_calls():
    """cs_pose_update_classify_frame_dev (round 6: the gate's lane of a map point also lists it for mapPointsClassify, the walks' camera
    centres are blocks of the same launch) against cs_pose_update_frame_dev + cs_map_points_classify_dev on a copy of the same state,
    frame after frame: every array the calls write is byte-identical, the classification examined points in every frame."""
    import torch

    from coslam_amd.poseupdate import TrackHistory

    sc = Scene(T=14)
    nC, N, nMap = sc.nC, sc.N, sc.nMap
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(3)
    fl0 = sc.flags0.copy()
    fl0[rng.random(nMap) < 0.15] |= 4                      # more uncertain points for the classification to look at
    newpt0 = (rng.random(nMap) < 0.3).astype(np.uint8)
    first0 = rng.integers(-40, 1, nMap).astype(np.int32)

    def state():
        d = dict(map=torch.from_numpy(sc.map0.copy()).to(dev), cov=torch.from_numpy(sc.cov0.copy()).to(dev), fl=torch.from_numpy(fl0.copy()).to(dev),
                 newpt=torch.from_numpy(newpt0.copy()).to(dev), sfn=torch.zeros(nMap, dtype=torch.int32, device=dev),
                 first=torch.from_numpy(first0.copy()).to(dev), err=[torch.zeros(N, dtype=torch.float64, device=dev) for _ in range(nC)],
                 stat=[torch.ones(N, dtype=torch.uint8, device=dev) for _ in range(nC)], cnt=torch.zeros(3, nC, dtype=torch.int32, device=dev),
                 ccnt=torch.zeros(2, dtype=torch.int32, device=dev), th=TrackHistory(nC, N, 32))
        return d

    A, B = state(), state()
    d_K = torch.from_numpy(sc.K.reshape(9).copy()).to(dev)
    d_iK = torch.from_numpy(sc.iK.reshape(9).copy()).to(dev)
    examined = 0
    for f in range(sc.T):
        recs = sc.frame(f)
        pf = Scene.point_feat(recs, nMap)
        Rs = np.stack([sc.Re[f][c].reshape(9) for c in range(nC)])
        ts = np.stack([sc.te[f][c] for c in range(nC)])
        d_R, d_t = torch.from_numpy(Rs).to(dev), torch.from_numpy(ts).to(dev)
        outs = []
        for S, fused in ((A, False), (B, True)):
            keep, cams = [], []
            for c, r in enumerate(recs):
                t_ = {k: torch.from_numpy(v.copy()).to(dev) for k, v in r.items()}
                keep.append(t_)
                cams.append(dict(K=d_K.data_ptr(), iK=d_iK.data_ptr(), xy=t_["xy"].data_ptr(), state=t_["state"].data_ptr(),
                                 slot2map=t_["slot2map"].data_ptr(), trackSpan=t_["trackSpan"].data_ptr(), reprojErr=S["err"][c].data_ptr(),
                                 isStatic=S["stat"][c].data_ptr()))
            d_pf = torch.from_numpy(pf.copy()).to(dev)
            kw = dict(d_numNodes=S["cnt"][0].data_ptr(), d_numOut=S["cnt"][1].data_ptr(), d_numDyn=S["cnt"][2].data_ptr())
            if fused:
                S["th"].pose_update_classify_frame_dev(s, cams, d_pf.data_ptr(), nMap, d_R.data_ptr(), d_t.data_ptr(), S["map"].data_ptr(),
                                                       S["cov"].data_ptr(), S["fl"].data_ptr(), 0, SIGMA, f, S["newpt"].data_ptr(), S["sfn"].data_ptr(),
                                                       S["first"].data_ptr(), 12.0, maxEpiErr=MAX_EPI, d_counts=S["ccnt"].data_ptr(), **kw)
            else:
                S["th"].pose_update_frame_dev(s, cams, d_pf.data_ptr(), nMap, d_R.data_ptr(), d_t.data_ptr(), S["map"].data_ptr(), S["cov"].data_ptr(),
                                              S["fl"].data_ptr(), 0, SIGMA, f, maxEpiErr=MAX_EPI, **kw)
                S["th"].map_points_classify_dev(s, cams, d_pf.data_ptr(), nMap, f, S["map"].data_ptr(), S["cov"].data_ptr(), S["fl"].data_ptr(),
                                                S["newpt"].data_ptr(), S["sfn"].data_ptr(), S["first"].data_ptr(), 12.0, d_counts=S["ccnt"].data_ptr())
            torch.cuda.synchronize()
            outs.append(dict(pf=d_pf.cpu().numpy(), s2m=[k_["slot2map"].cpu().numpy() for k_ in keep], map=S["map"].cpu().numpy(),
                             cov=S["cov"].cpu().numpy(), fl=S["fl"].cpu().numpy(), newpt=S["newpt"].cpu().numpy(), sfn=S["sfn"].cpu().numpy(),
                             err=[e.cpu().numpy() for e in S["err"]], stat=[e.cpu().numpy() for e in S["stat"]], cnt=S["cnt"].cpu().numpy(),
                             ccnt=S["ccnt"].cpu().numpy()))
        a, b = outs
        for k in a:
            xs, ys = (a[k], b[k]) if isinstance(a[k], list) else ([a[k]], [b[k]])
            for x, y in zip(xs, ys):
                assert x.tobytes() == y.tobytes(), (f, k)
        examined += int(a["ccnt"][0])
    assert examined > 50
    A["th"].close(), B["th"].close()
