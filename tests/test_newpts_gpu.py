"""From the NCC stage's candidate pairs to new map points on the device (cs_newpts_from_pairs_dev) against the restatement of
NewMapPtsNCC::run / output (oracle.new_map_points_from_pairs): seeds and the disparity guide, greedy matches with ties and conflicts,
tracks over up to four cameras, the re-projection / behind-the-camera gate, the covariances, point types, the features' new owners."""
import numpy as np
import pytest

from tests.poseupdate_scene import Scene

pytestmark = pytest.mark.gpu
W_IMG, H_IMG = 640, 480


def _scene(seed, nC=4, N=512, nMap=400, extra=300):
    rng = np.random.default_rng(seed)
    sc = Scene(nC=nC, N=N, nMap=nMap, T=3, seed=seed)
    f = 2
    recs = sc.frame(f)
    xy = [r["xy"].copy() for r in recs]
    state = [r["state"].copy() for r in recs]
    s2m = [r["slot2map"].copy() for r in recs]
    for c in range(nC):
        state[c][state[c] == 1] = 0
    R = np.stack([sc.Re[f][c].reshape(9) for c in range(nC)])
    t = np.stack([sc.te[f][c] for c in range(nC)])
    cap = nMap + extra
    mapPts = np.zeros((cap, 3))
    mapPts[:nMap] = sc.map0
    mapCov = np.zeros((cap, 9))
    mapCov[:nMap] = sc.cov0
    flags = np.zeros(cap, dtype=np.uint8)
    flags[:nMap] = sc.flags0
    pf = np.full((cap, nC), -1, dtype=np.int32)
    pf[:nMap] = Scene.point_feat([dict(state=state[c], slot2map=s2m[c]) for c in range(nC)], nMap)
    is_static = [(rng.random(N) < 0.85).astype(np.uint8) for _ in range(nC)]
    # candidate pairs: the same scene point unmapped in both cameras (true), plus wrong pairs; scores partly tied
    pairs = []
    for a in range(nC - 1):
        by_pt = {int(sc.slotPt[a + 1, j]): j for j in range(N) if state[a + 1][j] == 0 and s2m[a + 1][j] < 0}
        lst = []
        for i in range(N):
            if state[a][i] != 0 or s2m[a][i] >= 0:
                continue
            p = int(sc.slotPt[a, i])
            if p in by_pt and rng.random() < 0.8:
                lst.append((i, by_pt[p], rng.uniform(0, 5), round(float(rng.uniform(0.8, 1.0)), 2)))
            for _ in range(int(rng.integers(0, 3))):       # distractors
                j = int(rng.integers(0, N))
                if state[a + 1][j] == 0 and s2m[a + 1][j] < 0:
                    lst.append((i, j, rng.uniform(0, 50), round(float(rng.uniform(0.8, 1.0)), 2)))
        seen, uniq = set(), []
        for q in lst:
            if (q[0], q[1]) not in seen:
                seen.add((q[0], q[1])), uniq.append(q)
        rng.shuffle(uniq)
        pairs.append(uniq)
    return dict(sc=sc, nC=nC, N=N, nMap=nMap, cap=cap, xy=xy, state=state, s2m=s2m, R=R, t=t, mapPts=mapPts, mapCov=mapCov, flags=flags, pf=pf,
                is_static=is_static, pairs=pairs, frame=f)


@pytest.mark.parametrize("seed,max_disp", [(11, 80.0), (12, 25.0), (13, 1e9)])
def test_new_map_points_equal_the_restatement(hip, seed, max_disp):
    import torch

    import oracle
    from coslam_amd.ncc import NCC_PAIR_DTYPE
    from coslam_amd.newpts import NewPtsJob, newpts_from_pairs_dev, newpts_scratch_bytes

    S = _scene(seed)
    sc, nC, N, nMap, cap = S["sc"], S["nC"], S["N"], S["nMap"], S["cap"]
    if seed == 13:
        S["flags"][:nMap] |= 4      # no seeds at all (every map point uncertain): the unguided greedy
    o = dict(mapPts=S["mapPts"].copy(), mapCov=S["mapCov"].copy(), flags=S["flags"].copy(), newPt=np.zeros(cap, np.uint8),
             first=np.zeros(cap, np.int32), pf=S["pf"].copy(), s2m=[x.copy() for x in S["s2m"]], reproj=[np.zeros(N) for _ in range(nC)])
    res = oracle.new_map_points_from_pairs(N, S["pairs"], [sc.K] * nC, [sc.iK] * nC, S["R"], S["t"], S["xy"], S["state"], o["s2m"], S["is_static"],
                                           o["mapPts"], o["mapCov"], o["flags"], o["newPt"], o["first"], o["pf"], nMap, S["frame"],
                                           max_disp=max_disp, reproj=o["reproj"], W=W_IMG, H=H_IMG)
    assert len(res["new"]) > 20 and len(res["tracks"]) > len(res["new"]) and any(len(t) >= 3 for t in res["tracks"])
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    dK, diK = d(sc.K.reshape(9)), d(sc.iK.reshape(9))
    dxy, dst, ds2m, dstat = d(np.stack(S["xy"])), d(np.stack(S["state"])), d(np.stack(S["s2m"])), d(np.stack(S["is_static"]))
    drep = torch.zeros((nC, N), dtype=torch.float64, device=dev)
    CAPP = 4096
    dpairs = torch.zeros((nC - 1, CAPP * NCC_PAIR_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    dcnt = torch.zeros(nC - 1, dtype=torch.int32, device=dev)
    for a in range(nC - 1):
        arr = np.zeros(len(S["pairs"][a]), dtype=NCC_PAIR_DTYPE)
        for k, (i, j, e, n) in enumerate(S["pairs"][a]):
            arr[k] = (i, j, e, n)
        dpairs[a, :arr.nbytes] = torch.from_numpy(arr.view(np.uint8)).to(dev)
        dcnt[a] = len(arr)
    cams = [dict(K=dK.data_ptr(), iK=diK.data_ptr(), xy=dxy[c].data_ptr(), state=dst[c].data_ptr(), slot2map=ds2m[c].data_ptr(),
                 isStatic=dstat[c].data_ptr(), reprojErr=drep[c].data_ptr()) for c in range(nC)]
    job = NewPtsJob(cams, [dpairs[a].data_ptr() for a in range(nC - 1)], [dcnt[a:a + 1].data_ptr() for a in range(nC - 1)])
    dM, dC, dF, dPf = d(S["mapPts"]), d(S["mapCov"]), d(S["flags"]), d(S["pf"])
    dNew, dFirst = torch.zeros(cap, dtype=torch.uint8, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
    dCount = torch.tensor([nMap], dtype=torch.int32, device=dev)
    dScr = torch.zeros(newpts_scratch_bytes(nC, N), dtype=torch.uint8, device=dev)
    dOut = torch.zeros(4 + nC, dtype=torch.int32, device=dev)
    dR, dT = d(S["R"]), d(S["t"])
    newpts_from_pairs_dev(torch.cuda.current_stream().cuda_stream, job, N, CAPP, dR.data_ptr(), dT.data_ptr(), dM.data_ptr(), dC.data_ptr(),
                          dF.data_ptr(), dNew.data_ptr(), dFirst.data_ptr(), dPf.data_ptr(), cap, dCount.data_ptr(), S["frame"], dScr.data_ptr(),
                          dOut.data_ptr(), maxDisp=max_disp, W=W_IMG, H=H_IMG)
    torch.cuda.synchronize()
    out = dOut.cpu().tolist()
    assert out[0] == len(res["new"]) and out[1] == len(res["tracks"]) and out[3] == 0
    assert out[4:4 + nC - 1] == [int((res["matches"][a] >= 0).sum()) for a in range(nC - 1)]
    assert int(dCount.item()) == res["map_count"] == nMap + len(res["new"])
    # the scratch: every matched feature's partner, and the matched rows / columns of every pair as bit masks
    scr = dScr.view(torch.int32).cpu().numpy()
    nW = (N + 31) // 32
    match = scr[:(nC - 1) * N].reshape(nC - 1, N)
    rows = scr[(nC - 1) * N:(nC - 1) * (N + nW)].view(np.uint32).reshape(nC - 1, nW)
    cols = scr[(nC - 1) * (N + nW):(nC - 1) * (N + 2 * nW)].view(np.uint32).reshape(nC - 1, nW)
    bit = lambda msk, a: ((msk[a][np.arange(N) >> 5] >> (np.arange(N) & 31).astype(np.uint32)) & 1).astype(bool)   # noqa: E731
    for a in range(nC - 1):
        has = res["matches"][a] >= 0
        assert np.array_equal(bit(rows, a), has) and np.array_equal(match[a][has], res["matches"][a][has])
        is_col = np.zeros(N, dtype=bool)
        is_col[res["matches"][a][has]] = True
        assert np.array_equal(bit(cols, a), is_col)
    assert np.array_equal(dM.cpu().numpy(), o["mapPts"]) and np.array_equal(dC.cpu().numpy(), o["mapCov"])
    assert np.array_equal(dF.cpu().numpy(), o["flags"]) and np.array_equal(dNew.cpu().numpy(), o["newPt"]) and np.array_equal(dFirst.cpu().numpy(), o["first"])
    assert np.array_equal(dPf.cpu().numpy(), o["pf"])
    for c in range(nC):
        assert np.array_equal(ds2m[c].cpu().numpy(), o["s2m"][c]) and np.array_equal(drep[c].cpu().numpy(), o["reproj"][c])
    fl_new = o["flags"][res["new"]]
    assert ((fl_new == 4) | (fl_new == 0)).sum() > 10    # uncertain, or certain static after decidePointType; dynamic ones need two DYNAMIC features


def _run_both(S, max_disp, cap=None, pair_cap=4096):
    """oracle.new_map_points_from_pairs and cs_newpts_from_pairs_dev on scene S (map arrays cut to `cap` entries, candidate lists cut
    to `pair_cap` on the oracle's side -- the device is TOLD the full count); returns (oracle's result, oracle's arrays, device arrays)"""
    import torch

    import oracle
    from coslam_amd.ncc import NCC_PAIR_DTYPE
    from coslam_amd.newpts import NewPtsJob, newpts_from_pairs_dev, newpts_scratch_bytes

    sc, nC, N, nMap = S["sc"], S["nC"], S["N"], S["nMap"]
    cap = S["cap"] if cap is None else cap
    o = dict(mapPts=S["mapPts"][:cap].copy(), mapCov=S["mapCov"][:cap].copy(), flags=S["flags"][:cap].copy(), newPt=np.zeros(cap, np.uint8),
             first=np.zeros(cap, np.int32), pf=S["pf"][:cap].copy(), s2m=[x.copy() for x in S["s2m"]], reproj=[np.zeros(N) for _ in range(nC)])
    res = oracle.new_map_points_from_pairs(N, [pl[:pair_cap] for pl in S["pairs"]], [sc.K] * nC, [sc.iK] * nC, S["R"], S["t"], S["xy"], S["state"],
                                           o["s2m"], S["is_static"], o["mapPts"], o["mapCov"], o["flags"], o["newPt"], o["first"], o["pf"], nMap,
                                           S["frame"], max_disp=max_disp, reproj=o["reproj"], W=W_IMG, H=H_IMG)
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    dK, diK = d(sc.K.reshape(9)), d(sc.iK.reshape(9))
    dxy, dst, ds2m, dstat = d(np.stack(S["xy"])), d(np.stack(S["state"])), d(np.stack(S["s2m"])), d(np.stack(S["is_static"]))
    drep = torch.zeros((nC, N), dtype=torch.float64, device=dev)
    alloc = max(max(len(pl) for pl in S["pairs"]), pair_cap, 1)
    dpairs = torch.zeros((nC - 1, alloc * NCC_PAIR_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    dcnt = torch.zeros(nC - 1, dtype=torch.int32, device=dev)
    for a in range(nC - 1):
        arr = np.zeros(len(S["pairs"][a]), dtype=NCC_PAIR_DTYPE)
        for k, (i, j, e, n) in enumerate(S["pairs"][a]):
            arr[k] = (i, j, e, n)
        if arr.nbytes:
            dpairs[a, :arr.nbytes] = torch.from_numpy(arr.view(np.uint8)).to(dev)
        dcnt[a] = len(arr)
    cams = [dict(K=dK.data_ptr(), iK=diK.data_ptr(), xy=dxy[c].data_ptr(), state=dst[c].data_ptr(), slot2map=ds2m[c].data_ptr(),
                 isStatic=dstat[c].data_ptr(), reprojErr=drep[c].data_ptr()) for c in range(nC)]
    job = NewPtsJob(cams, [dpairs[a].data_ptr() for a in range(nC - 1)], [dcnt[a:a + 1].data_ptr() for a in range(nC - 1)])
    dM, dC, dF, dPf = d(S["mapPts"][:cap]), d(S["mapCov"][:cap]), d(S["flags"][:cap]), d(S["pf"][:cap])
    dNew, dFirst = torch.zeros(cap, dtype=torch.uint8, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
    dCount = torch.tensor([nMap], dtype=torch.int32, device=dev)
    dScr = torch.zeros(newpts_scratch_bytes(nC, N), dtype=torch.uint8, device=dev)
    dOut = torch.zeros(4 + nC, dtype=torch.int32, device=dev)
    dR, dT = d(S["R"]), d(S["t"])
    newpts_from_pairs_dev(torch.cuda.current_stream().cuda_stream, job, N, pair_cap, dR.data_ptr(), dT.data_ptr(), dM.data_ptr(), dC.data_ptr(),
                          dF.data_ptr(), dNew.data_ptr(), dFirst.data_ptr(), dPf.data_ptr(), cap, dCount.data_ptr(), S["frame"], dScr.data_ptr(),
                          dOut.data_ptr(), maxDisp=max_disp, W=W_IMG, H=H_IMG)
    torch.cuda.synchronize()
    g = dict(mapPts=dM.cpu().numpy(), mapCov=dC.cpu().numpy(), flags=dF.cpu().numpy(), newPt=dNew.cpu().numpy(), first=dFirst.cpu().numpy(),
             pf=dPf.cpu().numpy(), s2m=[ds2m[c].cpu().numpy() for c in range(nC)], reproj=[drep[c].cpu().numpy() for c in range(nC)],
             out=dOut.cpu().tolist(), count=int(dCount.item()))
    return res, o, g


def _same_map(o, g, nC):
    for k in ("mapPts", "mapCov", "flags", "newPt", "first", "pf"):
        assert np.array_equal(o[k], g[k]), k
    for c in range(nC):
        assert np.array_equal(o["s2m"][c], g["s2m"][c]) and np.array_equal(o["reproj"][c], g["reproj"][c]), c


def test_no_candidate_pairs_leave_the_map_alone(hip):
    """empty candidate lists (a frame whose NCC stage passes nothing): no track, no point, the map and every feature's owner untouched"""
    S = _scene(21)
    S["pairs"] = [[] for _ in S["pairs"]]
    res, o, g = _run_both(S, 80.0)
    assert res["new"] == [] and g["out"][:4] == [0, 0, 0, 0] and g["count"] == S["nMap"]
    _same_map(o, g, S["nC"])
    assert np.array_equal(g["mapPts"], S["mapPts"]) and all(np.array_equal(g["s2m"][c], S["s2m"][c]) for c in range(S["nC"]))


def test_a_single_pair_makes_one_point(hip):
    S = _scene(22)
    keep = next(q for q in S["pairs"][1] if S["sc"].slotPt[1, q[0]] == S["sc"].slotPt[2, q[1]])     # a true correspondence
    S["pairs"] = [[], [keep], []]
    res, o, g = _run_both(S, 1e9)
    assert len(res["tracks"]) == 1 and g["out"][1] == 1 and g["out"][0] == len(res["new"]) and g["count"] == res["map_count"]
    _same_map(o, g, S["nC"])


def test_map_without_room_takes_the_first_points_and_says_so(hip):
    """mapCap reached: the points are appended in track order until the arrays are full, the rest dropped with flag bit 2 (the reference's
    list grows without bound; a full map is this layout's own case)"""
    S = _scene(23)
    res_all, _, _ = _run_both(S, 80.0)
    room = 7
    assert len(res_all["new"]) > room + 5
    res, o, g = _run_both(S, 80.0, cap=S["nMap"] + room)
    assert len(res["new"]) == room and res["new"] == res_all["new"][:room]
    assert g["out"][3] & 2 and g["count"] >= S["nMap"] + room       # (the count runs on: the caller sees by how much the map fell short)
    _same_map(o, g, S["nC"])


def test_candidate_list_longer_than_its_capacity_is_cut_and_flagged(hip):
    """the NCC stage found more passing pairs than the list holds (count > pairCap): the first pairCap entries are used, flag bit 1"""
    S = _scene(24)
    pair_cap = 48
    assert all(len(pl) > pair_cap + 10 for pl in S["pairs"])
    res, o, g = _run_both(S, 80.0, pair_cap=pair_cap)
    assert g["out"][3] & 1 and g["out"][0] == len(res["new"]) > 3 and g["out"][1] == len(res["tracks"])
    _same_map(o, g, S["nC"])


def test_bad_arguments_are_refused(hip):
    import ctypes as C

    import coslam_amd
    L = coslam_amd.lib()
    assert L.cs_newpts_from_pairs_dev(0, None, 1, 512, None, None, None, 16, None, None, None, None, None, None, None, None, 10, None, 0,
                                      C.c_double(80.0), C.c_double(3.0), C.c_double(10.0), 2, 640, 480, None, None) != 0
    assert b"cs_newpts_from_pairs_dev" in L.cs_last_error()
    L.cs_newpts_scratch_bytes.restype = C.c_size_t
    assert L.cs_newpts_scratch_bytes(1, 512) == 0 and L.cs_newpts_scratch_bytes(4, 0) == 0


def test_candidate_mask_is_addslams_filter(hip):
    """cs_ncc_candidate_mask_dev: features of this frame on tracks of MORE than three frames, unmapped or mapped to a false point
    (NewMapPtsNCC::addSlam + getTrackedFeatPts(.., 3), reference src/app/SL_NewMapPointsInterCam.h:103-131, SL_SingleSLAM.cpp:173-184)"""
    import torch

    from coslam_amd.newpts import ncc_candidate_mask_dev

    rng = np.random.default_rng(5)
    nC, N, cap = 3, 300, 50
    state = rng.integers(-2, 2, (nC, N)).astype(np.int32)
    s2m = np.where(rng.random((nC, N)) < 0.5, rng.integers(0, cap, (nC, N)), -1).astype(np.int32)
    span = np.zeros((nC, 2 * N), dtype=np.int32)
    span[:, :N] = rng.integers(0, 10, (nC, N))
    span[:, N:] = span[:, :N] + rng.integers(0, 8, (nC, N))
    span[:, :N][rng.random((nC, N)) < 0.1] = -1
    flags = rng.choice([0, 1, 2, 4, 6], cap).astype(np.uint8)
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    dv = torch.zeros((nC, N), dtype=torch.int32, device=dev)
    keep = [d(state), d(s2m), d(span), d(flags)]
    ncc_candidate_mask_dev(torch.cuda.current_stream().cuda_stream, nC, N, keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(),
                           cap, dv.data_ptr())
    torch.cuda.synchronize()
    f1, f2 = span[:, :N], span[:, N:]
    want = ((state == 0) | (state == 1)) & (f1 >= 0) & (f2 - f1 >= 3) & ((s2m < 0) | ((flags[np.clip(s2m, 0, cap - 1)] & 2) != 0))
    assert np.array_equal(dv.cpu().numpy(), want.astype(np.int32)) and 20 < want.sum() < nC * N - 20


def test_new_map_points_equal_the_reference_golden(hip):
    """cs_newpts_from_pairs_dev on the reference's own scenes (tests/golden/newpts_golden.npz: featTracksFromMatches +
    NewMapPtsNCC::reconstructTracks + decidePointType run by the reference's code): the same new points in the same order, position and
    covariance bit for bit, the type decidePointType leaves (certain static / uncertain / dynamic), features, reprojErr."""
    import torch

    from coslam_amd.ncc import NCC_PAIR_DTYPE
    from coslam_amd.newpts import NewPtsJob, newpts_from_pairs_dev, newpts_scratch_bytes
    from tests.newpts_golden_util import GOLDEN, exact_inverse_of, scene

    g = np.load(GOLDEN)
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    for sc in range(int(g["n_scenes"])):
        S = scene(g, sc)
        nc, N, NS, cap, n_old = S["nc"], S["N"], S["NS"], S["cap"], S["n_old"]
        dK, diK = d(S["K"].reshape(9)), d(exact_inverse_of(S["K"]).reshape(9))
        dxy, dst, ds2m, dstat = d(np.stack(S["xy"])), d(np.stack(S["state"])), d(np.stack(S["s2m"])), d(np.stack(S["is_static"]))
        drep = torch.zeros((nc, NS), dtype=torch.float64, device=dev)
        CAPP = 1024
        dpairs = torch.zeros((nc - 1, CAPP * NCC_PAIR_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        dcnt = torch.zeros(nc - 1, dtype=torch.int32, device=dev)
        for a in range(nc - 1):
            arr = np.zeros(len(S["pairs"][a]), dtype=NCC_PAIR_DTYPE)
            for k, q in enumerate(S["pairs"][a]):
                arr[k] = q
            dpairs[a, :arr.nbytes] = torch.from_numpy(arr.view(np.uint8)).to(dev)
            dcnt[a] = len(arr)
        cams = [dict(K=dK.data_ptr(), iK=diK.data_ptr(), xy=dxy[c].data_ptr(), state=dst[c].data_ptr(), slot2map=ds2m[c].data_ptr(),
                     isStatic=dstat[c].data_ptr(), reprojErr=drep[c].data_ptr()) for c in range(nc)]
        job = NewPtsJob(cams, [dpairs[a].data_ptr() for a in range(nc - 1)], [dcnt[a:a + 1].data_ptr() for a in range(nc - 1)])
        dM, dC = torch.zeros((cap, 3), dtype=torch.float64, device=dev), torch.zeros((cap, 9), dtype=torch.float64, device=dev)
        dF, dPf = d(S["flags"]), d(S["pf"])
        dNew, dFirst = torch.zeros(cap, dtype=torch.uint8, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
        dCount = torch.tensor([n_old], dtype=torch.int32, device=dev)
        dScr = torch.zeros(newpts_scratch_bytes(nc, NS), dtype=torch.uint8, device=dev)
        dOut = torch.zeros(4 + nc, dtype=torch.int32, device=dev)
        dR, dT = d(S["R"]), d(S["t"])
        newpts_from_pairs_dev(torch.cuda.current_stream().cuda_stream, job, NS, CAPP, dR.data_ptr(), dT.data_ptr(), dM.data_ptr(), dC.data_ptr(),
                              dF.data_ptr(), dNew.data_ptr(), dFirst.data_ptr(), dPf.data_ptr(), cap, dCount.data_ptr(), S["frame"], dScr.data_ptr(),
                              dOut.data_ptr(), maxDisp=80.0, W=S["W"], H=S["H"])
        torch.cuda.synchronize()
        w = S["want"]
        out = dOut.cpu().tolist()
        n_new = len(w["M"])
        assert out[0] == n_new and out[1] == len(w["track_len"]) and out[2] == int((w["track_len"] >= 2).sum()) and out[3] == 0
        assert int(dCount.item()) == n_old + n_new
        new = slice(n_old, n_old + n_new)
        assert np.array_equal(dM.cpu().numpy()[new], w["M"]) and np.array_equal(dC.cpu().numpy()[new], w["cov"])
        assert np.array_equal(dF.cpu().numpy()[new], w["flags"]) and np.array_equal(dFirst.cpu().numpy()[new], w["first"])
        assert np.array_equal(dPf.cpu().numpy()[new], w["feat"]) and np.all(dNew.cpu().numpy()[new] == 1)
        assert np.array_equal(dF.cpu().numpy()[:n_old], S["flags"][:n_old])          # the old points keep their types
        for c in range(nc):
            assert np.array_equal(drep[c].cpu().numpy()[:N], w["reproj"][c])
