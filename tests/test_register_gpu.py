"""cs_register_search* (the search step of CoSLAM's map-point registration, reference src/app/SL_CoSLAM.cpp:731-757,
955-980, 1118-1145 + searchMahaNearestFeatPt, src/app/SL_SingleSLAM.cpp:1141-1164) against
  * the answers of the reference's own searchMahaNearestFeatPt (tests/golden/register_golden.npz), and
  * the oracle's restatement of the loops: every output table bit for bit (integers and binary64, same operation order),
at the headline's size (8 cameras x 2000 slots x 1500 points), on ragged / empty / duplicated inputs, and chained behind
the tracker's on-device hand-back."""
import os

import numpy as np
import pytest

import coslam_amd
import oracle
from coslam_amd.synth import Scene

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 640, 480
PIXEL_ERR_VAR = 10.0            # Const::PIXEL_ERR_VAR, reference src/app/SL_GlobParam.cpp:37
MODES = {"static": (PIXEL_ERR_VAR, 3 * PIXEL_ERR_VAR, PIXEL_ERR_VAR),       # SL_CoSLAM.cpp:750-756
         "dynamic": (PIXEL_ERR_VAR, 4 * PIXEL_ERR_VAR, PIXEL_ERR_VAR),      # :975-979
         "active": (2.5 * PIXEL_ERR_VAR, 3 * PIXEL_ERR_VAR, PIXEL_ERR_VAR)}  # :1131-1143


def assert_tables_equal(g, o, what=""):
    assert np.array_equal(g["slot"], o["slot"]), f"{what}: {(g['slot'] != o['slot']).sum()} candidates differ"
    assert np.array_equal(g["flags"], o["flags"]), what
    for k in ("m", "var", "dist"):
        assert np.array_equal(g[k], o[k]), f"{what}: {k} differs by {np.abs(g[k] - o[k]).max()}"


def test_register_search_returns_what_the_reference_returns(hip):
    """Golden queries (m, var, maxDist) with the reference's answers.  K = R = I, t = 0, M = (m, 1) and a covariance whose
    third row / column is zero make the kernel's projection and projected covariance EXACTLY the query's (sigma 0)."""
    g = np.load(os.path.join(GOLD, "register_golden.npz"))
    N = len(g["state"])
    I3 = np.eye(3)
    for md in np.unique(g["maxDist"]):
        q = np.nonzero(g["maxDist"] == md)[0]
        Ms = np.concatenate([g["m"][q], np.ones((len(q), 1))], axis=1)
        covs = np.zeros((len(q), 3, 3))
        covs[:, :2, :2] = g["var"][q].reshape(-1, 2, 2)
        r = coslam_amd.register_search(W, H, I3, I3, np.zeros(3), [g["xy"]], [g["state"]], [np.full(N, -1, np.int32)], [None],
                                       Ms, covs, np.full((len(q), 1), -1, np.int32), 0.0, float(md), 1.0)
        assert np.array_equal(r["m"][:, 0], g["m"][q]) and np.array_equal(r["var"][:, 0], g["var"][q])
        assert np.array_equal(r["slot"][:, 0], g["slot"][q])


def _rig(n_cams, N, P, seed, frame=3):
    """cameras of the synthetic rig, random feature records around the projections of the scene's points"""
    rng = np.random.default_rng(seed)
    sc = Scene(n_cams, W, H, P, seed=seed)
    Ks = np.stack([sc.K for _ in range(n_cams)])
    Rs = np.stack([sc.pose(c, frame)[0] for c in range(n_cams)])
    ts = np.stack([sc.pose(c, frame)[1] for c in range(n_cams)])
    Ms = sc.points[:P].copy()
    Ms[::37] *= -1.0                                    # some behind the cameras
    A = rng.normal(size=(P, 3, 3)) * 0.02
    covs = A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3)
    xy, state, s2m, dyn = [], [], [], []
    for c in range(n_cams):
        uv, vis = sc.project(c, frame)
        x = rng.uniform(0, W, N)
        y = rng.uniform(0, H, N)
        k = np.nonzero(vis[:P])[0][: N // 2]            # half of the slots sit near a point's projection
        x[: len(k)] = uv[k, 0] + rng.normal(0, 2.0, len(k))
        y[: len(k)] = uv[k, 1] + rng.normal(0, 2.0, len(k))
        x[5::101] = x[4::101][: len(x[5::101])]         # exact duplicates: first must win
        y[5::101] = y[4::101][: len(y[5::101])]
        r = rng.uniform(size=N)
        state.append(np.where(r < 0.06, -1, np.where(r < 0.08, -2, np.where(r < 0.3, 1, 0))).astype(np.int32))
        xy.append(np.concatenate([x, y]))
        s2m.append(np.where(rng.uniform(size=N) < 0.4, rng.integers(0, P, N), -1).astype(np.int32))
        dyn.append((rng.uniform(size=N) < 0.1).astype(np.uint8))
    pf = np.where(rng.uniform(size=(P, n_cams)) < 0.25, rng.integers(0, N, (P, n_cams)), -1).astype(np.int32)
    return Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf


@pytest.mark.parametrize("mode", ["static", "dynamic", "active"])
@pytest.mark.parametrize("n_cams,N,P", [(8, 2000, 1500), (3, 777, 211), (1, 65, 9), (2, 1, 40), (2, 5000, 333), (1, 9001, 70)])
def test_register_search_matches_oracle(hip, mode, n_cams, N, P):
    Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf = _rig(n_cams, N, P, seed=100 + N)
    sS, mD, sM = MODES[mode]
    o = oracle.register_search(W, H, Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf, sS, mD, sM)
    g = coslam_amd.register_search(W, H, Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf, sS, mD, sM)
    assert_tables_equal(g, o, f"{mode} {n_cams}x{N}x{P}")
    if P >= 1000:   # the headline's size: every branch was taken
        s = o["slot"]
        assert (s >= 0).sum() > P // 4 and (s == -1).any() and (s == -2).any() and (s == -3).any()
        assert (o["flags"][s >= 0] & 4).any() and not (o["flags"][s >= 0] & 4).all()


def test_register_search_edge_cases(hip):
    Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf = _rig(2, 300, 50, seed=9)
    sS, mD, sM = MODES["static"]
    # a camera whose frame has no feature at all; isDynamic absent
    state[1][:] = -1
    o = oracle.register_search(W, H, Ks, Rs, ts, xy, state, s2m, [None, None], Ms, covs, pf, sS, mD, sM)
    g = coslam_amd.register_search(W, H, Ks, Rs, ts, xy, state, s2m, [None, None], Ms, covs, pf, sS, mD, sM)
    assert_tables_equal(g, o, "empty camera")
    assert (o["slot"][:, 1] == -4).any() and not (o["slot"][:, 1] >= 0).any()
    # every point already attached everywhere
    full = np.zeros_like(pf)
    g = coslam_amd.register_search(W, H, Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, full, sS, mD, sM)
    assert (g["slot"] == -1).all() and not g["dist"].any()
    # no points: nothing to do
    g = coslam_amd.register_search(W, H, Ks, Rs, ts, xy, state, s2m, dyn, Ms[:0], covs[:0], pf[:0], sS, mD, sM)
    assert g["slot"].shape == (0, 2)
    with pytest.raises(coslam_amd.CoslamHipError):
        coslam_amd.register_search(W, H, Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf, sS, 0.0, sM)     # maxDist must be > 0
    with pytest.raises(coslam_amd.CoslamHipError):
        coslam_amd.register_search_dev(0, [], 10, W, H, 1, 1, 1, 1, sS, mD, sM, 1, 1, 1, 1, 1)             # no cameras


def test_register_search_behind_the_tracker_and_the_hand_back(hip):
    """Data-coupled: detect on rendered frames of 3 cameras -> cs_klt_handback_dev -> cs_register_search_dev reading the
    hand-back's device records in place; the oracle runs the same chain on the host."""
    import torch

    dev = torch.device("cuda:0")
    n_cams, fw, fh, P = 3, 50, 40, 1200
    N = fw * fh
    sc = Scene(n_cams, W, H, 4000, seed=31)
    cfg = coslam_amd.KLT_SequenceTrackerConfig(nIterations=10, nLevels=4, levelSkip=1, windowWidth=7, trackWithGain=1,
                                               minCornerness=3000.0, convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=5)
    K = sc.K
    kud = np.zeros(7)
    rng = np.random.default_rng(3)
    Ms = sc.points[:P].copy()
    A = rng.normal(size=(P, 3, 3)) * 0.01
    covs = A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3)
    d_K, d_kud, d_map = (torch.from_numpy(a.copy()).to(dev) for a in (K.ravel(), kud, Ms))
    d_cov = torch.from_numpy(covs.reshape(-1).copy()).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    hb, o_xy, o_state, o_s2m, Rs, ts, keep = [], [], [], [], [], [], []
    for c in range(n_cams):
        trk = coslam_amd.KLT_SequenceTracker(cfg, device=0)
        trk.allocate(W, H, 4, fw, fh)
        _, dest = trk.detect(sc.render(c, 0))
        trk.close()
        s2m = np.full(N, -1, np.int32)
        span, xyo = np.full(2 * N, -1, np.int32), np.zeros(2 * N)
        r = oracle.handback(dest, W, H, K, kud, Ms, s2m.copy(), span, xyo, 0)
        o_xy.append(xyo)
        o_state.append(r["state"])
        s2m[::3] = rng.integers(0, P, len(s2m[::3]))     # "map initialisation" after the first hand-back, on both sides
        o_s2m.append(s2m)
        t = dict(dest=torch.from_numpy(dest.view(np.int32).copy()).to(dev), s2m=torch.full((N,), -1, dtype=torch.int32, device=dev),
                 tl=torch.full((2 * N,), -1, dtype=torch.int32, device=dev), xy=torch.zeros(2 * N, dtype=torch.float64, device=dev),
                 state=torch.zeros(N, dtype=torch.int32, device=dev), Ms=torch.zeros(192 * 3, dtype=torch.float64, device=dev),
                 ms=torch.zeros(192 * 2, dtype=torch.float64, device=dev), sel=torch.zeros(192, dtype=torch.int32, device=dev),
                 npts=torch.zeros(1, dtype=torch.int32, device=dev))
        keep.append(t)
        hb.append(dict(dest=t["dest"].data_ptr(), K=d_K.data_ptr(), kud=d_kud.data_ptr(), mapPts=d_map.data_ptr(),
                       slot2map=t["s2m"].data_ptr(), trackSpan=t["tl"].data_ptr(), xy=t["xy"].data_ptr(), state=t["state"].data_ptr(),
                       Ms=t["Ms"].data_ptr(), ms=t["ms"].data_ptr(), sel=t["sel"].data_ptr(), npts=t["npts"].data_ptr()))
        R, tt = sc.pose(c, 0)
        Rs.append(R)
        ts.append(tt)
    coslam_amd.handback_dev(stream, hb, N, W, H, 16, 12, 192, frame=0)
    for c in range(n_cams):
        keep[c]["s2m"].copy_(torch.from_numpy(o_s2m[c]))
    d_R = torch.from_numpy(np.stack(Rs).reshape(-1).copy()).to(dev)
    d_t = torch.from_numpy(np.stack(ts).reshape(-1).copy()).to(dev)
    pf = np.full((P, n_cams), -1, np.int32)
    d_pf = torch.from_numpy(pf).to(dev)
    out = dict(slot=torch.zeros(P * n_cams, dtype=torch.int32, device=dev), m=torch.zeros(P * n_cams * 2, dtype=torch.float64, device=dev),
               var=torch.zeros(P * n_cams * 4, dtype=torch.float64, device=dev), dist=torch.zeros(P * n_cams, dtype=torch.float64, device=dev),
               flags=torch.zeros(P * n_cams, dtype=torch.int32, device=dev))
    cams = [dict(K=d_K.data_ptr(), R=d_R.data_ptr() + 72 * c, t=d_t.data_ptr() + 24 * c, xy=keep[c]["xy"].data_ptr(),
                 state=keep[c]["state"].data_ptr(), slot2map=keep[c]["s2m"].data_ptr()) for c in range(n_cams)]
    sS, mD, sM = MODES["active"]
    coslam_amd.register_search_dev(stream, cams, N, W, H, P, d_map.data_ptr(), d_cov.data_ptr(), d_pf.data_ptr(), sS, mD, sM,
                                   out["slot"].data_ptr(), out["m"].data_ptr(), out["var"].data_ptr(), out["dist"].data_ptr(),
                                   out["flags"].data_ptr())
    torch.cuda.synchronize()
    for c in range(n_cams):
        assert np.array_equal(keep[c]["xy"].cpu().numpy(), o_xy[c]) and np.array_equal(keep[c]["s2m"].cpu().numpy(), o_s2m[c])
    o = oracle.register_search(W, H, np.stack([K] * n_cams), np.stack(Rs), np.stack(ts), o_xy, o_state, o_s2m, [None] * n_cams,
                               Ms, covs, pf, sS, mD, sM)
    g = dict(slot=out["slot"].cpu().numpy().reshape(P, n_cams), flags=out["flags"].cpu().numpy().reshape(P, n_cams),
             m=out["m"].cpu().numpy().reshape(P, n_cams, 2), var=out["var"].cpu().numpy().reshape(P, n_cams, 4),
             dist=out["dist"].cpu().numpy().reshape(P, n_cams))
    assert_tables_equal(g, o, "chained")
    assert (o["slot"] >= 0).sum() > 500
    # points that project onto a detected corner find it: within ~4 px (scaled distance r^2 / (sigma^2 maxDist) < 1e-3), and
    # such a candidate passes the mergability term (r^2 / pixelErrVar^2 <= 1)
    near = (o["slot"] >= 0) & (o["dist"] < 1e-3)
    assert near.sum() > 30 and (o["flags"][near] & 4).all()
    # ... and the two passes of a frame (active points: 2.5 sigma; current static points: 1 sigma, a different point set of a
    # different size and a pointFeat table with attached points) in ONE launch give the tables of two separate launches
    from coslam_amd.register import register_search_passes_dev

    P2 = 700
    pf2 = np.full((P2, n_cams), -1, np.int32)
    pf2[::7, 1] = 5                                   # these points already have a feature of this frame in camera 1
    d_pf2 = torch.from_numpy(pf2).to(dev)

    def tables(n):
        return dict(slot=torch.full((n * n_cams,), 99, dtype=torch.int32, device=dev), m=torch.zeros(n * n_cams * 2, dtype=torch.float64, device=dev),
                    var=torch.zeros(n * n_cams * 4, dtype=torch.float64, device=dev), dist=torch.zeros(n * n_cams, dtype=torch.float64, device=dev),
                    flags=torch.zeros(n * n_cams, dtype=torch.int32, device=dev))

    sS2, mD2, sM2 = MODES["static"]
    sep, fus = [tables(P), tables(P2)], [tables(P), tables(P2)]
    specs = [(P, d_map.data_ptr(), d_cov.data_ptr(), d_pf.data_ptr(), sS, mD, sM),
             (P2, d_map.data_ptr() + 24 * 100, d_cov.data_ptr() + 72 * 100, d_pf2.data_ptr(), sS2, mD2, sM2)]
    for (n, pM, pC, pF, a, b, c_), o_ in zip(specs, sep):
        coslam_amd.register_search_dev(stream, cams, N, W, H, n, pM, pC, pF, a, b, c_, o_["slot"].data_ptr(), o_["m"].data_ptr(),
                                       o_["var"].data_ptr(), o_["dist"].data_ptr(), o_["flags"].data_ptr())
    register_search_passes_dev(stream, cams, N, W, H,
                               [dict(P=n, sigmaSearch=a, maxDist=b, sigmaMerge=c_, M=pM, cov=pC, pointFeat=pF, slot=o_["slot"].data_ptr(),
                                     m=o_["m"].data_ptr(), var=o_["var"].data_ptr(), dist=o_["dist"].data_ptr(), flags=o_["flags"].data_ptr())
                                for (n, pM, pC, pF, a, b, c_), o_ in zip(specs, fus)])
    torch.cuda.synchronize()
    for o1, o2 in zip(sep, fus):
        for k in o1:
            assert np.array_equal(o1[k].cpu().numpy().view(np.uint8), o2[k].cpu().numpy().view(np.uint8)), k
    assert (fus[1]["slot"].cpu().numpy().reshape(P2, n_cams)[::7, 1] == -1).all()


def test_current_points_list_drives_the_search_like_the_whole_table(hip):
    """cs_register_list_current_dev + cs_register_pass::list (the registration must walk curMapPts -- the points with a
    feature of this frame, wherever they sit in the map, the points genNewMapPoints appended included -- not the first P rows):
    the list is the map-ordered set {p < mapCount, not false, some pointFeat[p][c] >= 0}; the rows of unlisted points lose their
    candidates; a search over the list fills the listed rows of whole-map tables with exactly what a search over all rows writes
    there and touches no other row; a pass capped below the list's length searches its first P entries only."""
    import torch

    from coslam_amd.register import register_list_current_dev, register_search_passes_dev

    n_cams, N, P = 4, 700, 3000
    Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf = _rig(n_cams, N, P, seed=77)
    rng = np.random.default_rng(9)
    pf[rng.uniform(size=P) < 0.6] = -1                     # most points hold no feature this frame
    flags = np.where(rng.uniform(size=P) < 0.1, 2, np.where(rng.uniform(size=P) < 0.1, 1, 0)).astype(np.uint8)
    live = 2600                                            # rows behind the live count are spare capacity
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    d_K, d_R, d_t = d(Ks.reshape(n_cams, 9)), d(Rs.reshape(n_cams, 9)), d(ts)
    d_xy, d_st, d_s2m, d_dyn = [d(a) for a in xy], [d(a) for a in state], [d(a) for a in s2m], [d(a) for a in dyn]
    cams = [dict(K=d_K[c].data_ptr(), R=d_R[c].data_ptr(), t=d_t[c].data_ptr(), xy=d_xy[c].data_ptr(), state=d_st[c].data_ptr(),
                 slot2map=d_s2m[c].data_ptr(), isDynamic=d_dyn[c].data_ptr()) for c in range(n_cams)]
    d_M, d_cov, d_pf, d_fl = d(Ms), d(covs.reshape(P, 9)), d(pf), d(flags)
    d_cnt = torch.tensor([live], dtype=torch.int32, device=dev)
    d_list = torch.full((P,), 12345, dtype=torch.int32, device=dev)
    d_n = torch.zeros(1, dtype=torch.int32, device=dev)

    def tables(fill):
        return dict(slot=torch.full((P, n_cams), fill, dtype=torch.int32, device=dev), m=torch.full((P, n_cams, 2), 7.5, dtype=torch.float64, device=dev),
                    var=torch.full((P, n_cams, 4), 7.5, dtype=torch.float64, device=dev), dist=torch.full((P, n_cams), 7.5, dtype=torch.float64, device=dev),
                    flags=torch.full((P, n_cams), fill, dtype=torch.int32, device=dev))

    T_list, T_all = tables(99), tables(99)
    # a first call with MORE points holding a feature, its rows filled; the call that counts then drops the rows that left the list
    pf_more = pf.copy()
    extra = rng.choice(np.nonzero((pf < 0).all(axis=1) & (np.arange(P) < live) & ((flags & 2) == 0))[0], 300, replace=False)
    pf_more[extra, 0] = 5
    d_pf_more = d(pf_more)
    register_list_current_dev(s, n_cams, P, d_cnt.data_ptr(), d_pf_more.data_ptr(), d_fl.data_ptr(), d_list.data_ptr(), d_n.data_ptr(),
                              T_list["slot"].data_ptr())
    torch.cuda.synchronize()
    want_more = np.nonzero((np.arange(P) < live) & ((flags & 2) == 0) & (pf_more >= 0).any(axis=1))[0]
    assert d_n.item() == len(want_more) and np.array_equal(d_list.cpu().numpy()[:len(want_more)], want_more)
    assert (T_list["slot"].cpu().numpy() == 99).all()          # (nothing had been on a list before: no row to clear)
    register_list_current_dev(s, n_cams, P, d_cnt.data_ptr(), d_pf.data_ptr(), d_fl.data_ptr(), d_list.data_ptr(), d_n.data_ptr(),
                              T_list["slot"].data_ptr())
    torch.cuda.synchronize()
    want = np.nonzero((np.arange(P) < live) & ((flags & 2) == 0) & (pf >= 0).any(axis=1))[0]
    lst = d_list.cpu().numpy()
    assert d_n.item() == len(want) > 500 and np.array_equal(lst[:len(want)], want) and (lst[len(want):] == -1).all()
    unlisted = np.setdiff1d(np.arange(P), want)
    sl0 = T_list["slot"].cpu().numpy()
    assert (sl0[extra] == -1).all() and (sl0[np.setdiff1d(np.arange(P), extra)] == 99).all()   # exactly the rows that left the list
    T_list["slot"][torch.from_numpy(unlisted).to(dev)] = -1     # (a table in use starts at -1 and only listed rows are ever written)

    def a_pass(T, P_, lst_=0):
        return dict(P=P_, sigmaSearch=PIXEL_ERR_VAR, maxDist=3 * PIXEL_ERR_VAR, sigmaMerge=PIXEL_ERR_VAR, M=d_M.data_ptr(), cov=d_cov.data_ptr(),
                    pointFeat=d_pf.data_ptr(), slot=T["slot"].data_ptr(), m=T["m"].data_ptr(), var=T["var"].data_ptr(), dist=T["dist"].data_ptr(),
                    flags=T["flags"].data_ptr(), mapFlags=d_fl.data_ptr(), maxDistDynamic=4 * PIXEL_ERR_VAR, list=lst_)

    register_search_passes_dev(s, cams, N, W, H, [a_pass(T_all, P)])
    register_search_passes_dev(s, cams, N, W, H, [a_pass(T_list, P, d_list.data_ptr())])
    torch.cuda.synchronize()
    for k in T_all:
        a, b = T_all[k].cpu().numpy(), T_list[k].cpu().numpy()
        assert np.array_equal(a[want], b[want]), k
    assert (T_list["flags"].cpu().numpy()[unlisted] == 99).all() and (T_list["m"].cpu().numpy()[unlisted] == 7.5).all()
    assert (T_list["slot"].cpu().numpy()[want] >= 0).sum() > 300
    # a cap below the list's length: the first 256 listed points only
    T_cap = tables(55)
    register_search_passes_dev(s, cams, N, W, H, [a_pass(T_cap, 256, d_list.data_ptr())])
    torch.cuda.synchronize()
    sl = T_cap["slot"].cpu().numpy()
    assert np.array_equal(sl[want[:256]], T_all["slot"].cpu().numpy()[want[:256]]) and (sl[want[256:]] == 55).all()


def test_candidate_records_of_listed_rows_round_trip(hip):
    """cs_register_candidates_pack_list_dev / _unpack_list_dev: with the cameras sharded over ranks the own cameras' columns of the listed
    rows travel as records of `cap` rows; unpacking every rank's record rebuilds the tables' listed rows, leaves the skipped rank's
    columns and the unlisted rows alone."""
    import ctypes as C

    import torch

    from coslam_amd._lib import check

    L = coslam_amd.lib()
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(3)
    nCams, nOwn, P, cap = 4, 2, 900, 256
    world = nCams // nOwn
    lst = np.full(P, -1, np.int32)
    rows = np.sort(rng.choice(P, 200, replace=False)).astype(np.int32)
    lst[:200] = rows
    slot = rng.integers(-4, 500, (P, nCams)).astype(np.int32)
    flags = rng.integers(0, 8, (P, nCams)).astype(np.int32)
    merg = rng.integers(0, 3, (P, nCams)).astype(np.uint8)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    d_list, d_slot, d_flags, d_merg = d(lst), d(slot), d(flags), d(merg)
    recv = torch.zeros(world * 3 * nOwn * cap, dtype=torch.int32, device=dev)
    vp = C.c_void_p
    for r in range(world):
        send = recv[r * 3 * nOwn * cap:(r + 1) * 3 * nOwn * cap]
        check(L.cs_register_candidates_pack_list_dev(0, vp(s), cap, nCams, r * nOwn, nOwn, vp(d_list.data_ptr()), vp(d_slot.data_ptr()),
                                                     vp(d_flags.data_ptr()), vp(d_merg.data_ptr()), vp(send.data_ptr())), "pack")
    o_slot, o_flags = torch.full((P, nCams), -77, dtype=torch.int32, device=dev), torch.full((P, nCams), -77, dtype=torch.int32, device=dev)
    o_merg = torch.full((P, nCams), 77, dtype=torch.uint8, device=dev)
    check(L.cs_register_candidates_unpack_list_dev(0, vp(s), cap, nCams, nOwn, 1, vp(d_list.data_ptr()), vp(recv.data_ptr()), vp(o_slot.data_ptr()),
                                                   vp(o_flags.data_ptr()), vp(o_merg.data_ptr())), "unpack")
    torch.cuda.synchronize()
    gs, gf, gm = o_slot.cpu().numpy(), o_flags.cpu().numpy(), o_merg.cpu().numpy()
    own1 = slice(nOwn, 2 * nOwn)                            # rank 1's columns: skipped
    assert (gs[:, own1] == -77).all() and (gm[:, own1] == 77).all()
    assert np.array_equal(gs[rows][:, :nOwn], slot[rows][:, :nOwn]) and np.array_equal(gf[rows][:, :nOwn], flags[rows][:, :nOwn])
    assert np.array_equal(gm[rows][:, :nOwn], merg[rows][:, :nOwn])
    other = np.setdiff1d(np.arange(P), rows)
    assert (gs[other] == -77).all() and (gf[other] == -77).all()


def test_current_points_list_with_a_cap_drops_the_overflow_cleanly(hip):
    """cs_register_list_current_cap_dev: the passes behind the list cover listCap rows.  A current point beyond
    the cap must not keep an older frame's candidates: it is left off the list, its candidate row is cleared, the overflow is counted."""
    import torch

    from coslam_amd.register import register_list_current_dev

    n_cams, P, cap = 4, 3000, 400
    rng = np.random.default_rng(4)
    pf = np.where(rng.uniform(size=(P, n_cams)) < 0.15, 5, -1).astype(np.int32)
    flags = np.where(rng.uniform(size=P) < 0.1, 2, 0).astype(np.uint8)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    d_pf, d_fl = torch.from_numpy(pf).to(dev), torch.from_numpy(flags).to(dev)
    d_list = torch.full((P,), -1, dtype=torch.int32, device=dev)
    d_n, d_over = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    slot = torch.full((P, n_cams), 7, dtype=torch.int32, device=dev)    # every row carries a (stale) candidate
    want = np.nonzero(((flags & 2) == 0) & (pf >= 0).any(axis=1))[0]
    assert len(want) > 2 * cap
    for rep in (1, 2):
        register_list_current_dev(s, n_cams, P, 0, d_pf.data_ptr(), d_fl.data_ptr(), d_list.data_ptr(), d_n.data_ptr(), slot.data_ptr(),
                                  listCap=cap, d_overflow=d_over.data_ptr())
        torch.cuda.synchronize()
        lst = d_list.cpu().numpy()
        assert d_n.item() == cap and np.array_equal(lst[:cap], want[:cap]) and (lst[cap:] == -1).all()
        assert d_over.item() == rep * (len(want) - cap)                 # accumulates over the frames
        sl = slot.cpu().numpy()
        assert (sl[want[cap:]] == -1).all() and (sl[want[:cap]] == 7).all()
    # no cap given: the whole list, nothing counted
    d_over.zero_()
    register_list_current_dev(s, n_cams, P, 0, d_pf.data_ptr(), d_fl.data_ptr(), d_list.data_ptr(), d_n.data_ptr(), 0, d_overflow=d_over.data_ptr())
    torch.cuda.synchronize()
    assert d_n.item() == len(want) and d_over.item() == 0 and np.array_equal(d_list.cpu().numpy()[:len(want)], want)
