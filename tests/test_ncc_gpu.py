"""cs_ncc_* (NCC blocks + the epipolar / NCC matrices of the inter-camera matching; reference src/slam/SL_NCCBlock.cpp:15-54,
258-264, src/slam/SL_FeatureMatching.cpp:3-46) against the reference's own outputs (tests/golden/ncc_golden.npz) and the
oracle's restatement: bytes and binary64 values bit for bit -- the pair scores come out of v_mfma_i32_16x16x32_i8 as exact
integers -- at the headline's size (2000 x 2000 features), on ragged sizes and on degenerate blocks."""
import os

import numpy as np
import pytest

import coslam_amd
import oracle
from coslam_amd.synth import Scene

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(g, img1, x1, y1, img2, x2, y2, scale, F, epiMax, nccMin, what=""):
    o1 = oracle.ncc_blocks(img1, x1, y1, scale)
    o2 = oracle.ncc_blocks(img2, x2, y2, scale)
    for t, o in (("1", o1), ("2", o2)):
        assert np.array_equal(g["valid" + t], o[2]), what
        assert np.array_equal(g["blocks" + t], o[0]), what
        assert np.array_equal(g["abc" + t], o[1]), what
    epi, ncc = oracle.ncc_epi_mat(F, x1, y1, *o1, x2, y2, *o2, epiMax, nccMin)
    assert np.array_equal(g["ncc"], ncc), f"{what}: {(g['ncc'] != ncc).sum()} scores differ"
    assert np.array_equal(g["epi"], epi), what
    return epi, ncc


def test_ncc_matches_the_reference_golden(hip):
    g = np.load(os.path.join(GOLD, "ncc_golden.npz"))
    r = coslam_amd.ncc_match_between(g["img1"], g["x1"], g["y1"], g["img2"], g["x2"], g["y2"], float(g["scale"]), g["F"],
                                     float(g["epiMax"]), float(g["nccMin"]))
    for t in ("1", "2"):
        assert np.array_equal(r["valid" + t], g["valid" + t])
        assert np.array_equal(r["blocks" + t], g["blocks" + t])
        assert np.array_equal(r["abc" + t], g["abc" + t])
    assert np.array_equal(r["epi"], g["epi"]) and np.array_equal(r["ncc"], g["ncc"])


def _fundamental(sc, c1, c2, f=0):
    """F with x1^T F x2 = 0 for the scene's cameras c1, c2 (pixels of the full image)"""
    R1, t1 = sc.pose(c1, f)
    R2, t2 = sc.pose(c2, f)
    R = R1 @ R2.T
    t = t1 - R @ t2
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(sc.K)
    return Ki.T @ tx @ R @ Ki


@pytest.mark.parametrize("M,N", [(2000, 2000), (777, 1301), (1, 65), (64, 1)])
def test_ncc_matches_oracle_on_rendered_cameras(hip, M, N):
    """two cameras of the synthetic rig, small images = 0.3 x decimations of the rendered frames, features at the projections
    of the scene's points (so true matches exist) plus random ones"""
    W, H, scale = 640, 480, 0.3
    sc = Scene(2, W, H, 3000, seed=77)
    rng = np.random.default_rng(M + N)
    small = []
    for c in range(2):
        im = sc.render(c, 0).astype(np.float64)
        ys = (np.arange(int(H * scale)) / scale).astype(int)
        xs = (np.arange(int(W * scale)) / scale).astype(int)
        small.append(np.ascontiguousarray(im[np.ix_(ys, xs)]).astype(np.uint8))
    pts = []
    for c, n in ((0, M), (1, N)):
        uv, vis = sc.project(c, 0)
        k = np.nonzero(vis)[0][: n // 2]
        x = np.concatenate([uv[k, 0], rng.uniform(-20, W + 20, n - len(k))])
        y = np.concatenate([uv[k, 1], rng.uniform(-20, H + 20, n - len(k))])
        pts.append((x, y))
    F = _fundamental(sc, 0, 1)
    # NewMapPtsNCCParam's defaults (epipolar 50 px, NCC 0.8), then a looser score gate so that many pairs are kept
    r = coslam_amd.ncc_match_between(small[0], *pts[0], small[1], *pts[1], scale, F, 50.0, 0.8)
    _check(r, small[0], *pts[0], small[1], *pts[1], scale, F, 50.0, 0.8, f"{M}x{N}")
    r = coslam_amd.ncc_match_between(small[0], *pts[0], small[1], *pts[1], scale, F, 50.0, 0.3)
    epi, ncc = _check(r, small[0], *pts[0], small[1], *pts[1], scale, F, 50.0, 0.3, f"{M}x{N} loose")
    if M >= 700:
        kept = ncc != -1
        assert kept.sum() > 100 and (r["valid1"] == 0).any()
        # true correspondences (the same scene point seen by both cameras) sit on their epipolar lines
        n_true = min(M, N) // 2
        d = np.abs(np.diag(oracle.ncc_epi_mat(F, *pts[0], r["blocks1"], r["abc1"], r["valid1"], *pts[1], r["blocks2"], r["abc2"],
                                              r["valid2"], 1e9, -2.0)[0])[:n_true // 4])
        assert np.median(d) < 1.0


def test_ncc_degenerate_blocks_and_arguments(hip):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    img[:, 32:] = 255                                           # saturated half: flat blocks (C = inf), largest byte products
    flat = np.full((48, 64), 255, np.uint8)
    x = rng.uniform(0, 64 / 0.3, 130)
    y = rng.uniform(0, 48 / 0.3, 130)
    F = rng.normal(size=(3, 3))
    for a, b in ((img, img), (img, flat), (flat, flat)):
        r = coslam_amd.ncc_match_between(a, x, y, b, x[:97], y[:97], 0.3, F, 1e9, -2.0)   # every pair with both blocks is scored
        _check(r, a, x, y, b, x[:97], y[:97], 0.3, F, 1e9, -2.0)
    r = coslam_amd.ncc_match_between(img, x[:0], y[:0], img, x, y, 0.3, F, 50.0, 0.8)          # no features on one side
    assert r["ncc"].shape == (0, 130)
    with pytest.raises(coslam_amd.CoslamHipError):
        coslam_amd.ncc_blocks_dev(0, 1, 5, 5, 3, 1, 1, 0.3, 1, 1, 1)                           # image smaller than a block


@pytest.mark.parametrize("W,H,scale", [(640, 480, 0.3), (322, 250, 0.3), (200, 150, 1.0), (640, 480, 0.5), (101, 67, 0.3)])
def test_get_ncc_blocks_matches_the_restated_opencv_path(hip, W, H, scale):
    """cs_ncc_get_blocks_dev = getNCCBlocks (reference src/slam/SL_NCCBlock.cpp:79-155, what matchBetween calls): the resized
    image, every block and A / B / C / avgI bit for bit against oracle/ncc_oracle.c's restatement of cv::resize + cv::getRectSubPix;
    points inside, on the border, outside the image, with fractional positions."""
    import torch

    rng = np.random.default_rng(W + H)
    img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    img[H // 3: H // 3 + 20, W // 4: W // 4 + 30] = 200          # a flat patch: C = inf there
    n = 700
    x = rng.uniform(-25, W + 25, n)
    y = rng.uniform(-25, H + 25, n)
    x[:50], y[:50] = np.floor(x[:50]), np.floor(y[:50])             # integer positions too
    dev = torch.device("cuda:0")
    d_img, d_x, d_y = torch.from_numpy(img).to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    ws, hs = coslam_amd.ncc_scaled_dims(W, H, scale)
    d_small = torch.zeros(max(ws * hs, 1), dtype=torch.uint8, device=dev)
    d_blk = torch.zeros((n, 128), dtype=torch.uint8, device=dev)
    d_abc = torch.zeros((n, 4), dtype=torch.float64, device=dev)
    d_val = torch.zeros(n, dtype=torch.int32, device=dev)
    coslam_amd.ncc_get_blocks_dev(torch.cuda.current_stream().cuda_stream, d_img.data_ptr(), W, H, n, d_x.data_ptr(), d_y.data_ptr(),
                                  scale, d_small.data_ptr(), d_blk.data_ptr(), d_abc.data_ptr(), d_val.data_ptr())
    torch.cuda.synchronize()
    if scale != 1.0:
        small_o = oracle.resize_linear_u8(img, scale, scale)
        assert small_o.shape == (hs, ws)
        assert np.array_equal(d_small.cpu().numpy().reshape(hs, ws), small_o), "resized image differs"
    blk_o, abc_o = oracle.get_ncc_blocks(img, x, y, scale)
    assert np.array_equal(d_blk.cpu().numpy(), blk_o), "blocks differ"
    assert np.array_equal(d_abc.cpu().numpy(), abc_o), "A / B / C / avgI differ"     # (inf == inf for the flat blocks)
    assert np.all(d_val.cpu().numpy() == 1)


def test_match_between_full_is_get_ncc_blocks_plus_the_matrices(hip):
    """cs_ncc_match_between_full = matchBetween's own data path (src/app/SL_NewMapPointsInterCam.cpp:273-290): FULL images ->
    getNCCBlocks(0.3) -> getEpiNccMat; both matrices bit for bit against the oracle."""
    W, H = 640, 480
    sc = Scene(2, W, H, 3000, seed=5)
    im1, im2 = sc.render(0, 0), sc.render(1, 0)
    rng = np.random.default_rng(9)
    M, N = 300, 280
    x1, y1, x2, y2 = rng.uniform(0, W, M), rng.uniform(0, H, M), rng.uniform(0, W, N), rng.uniform(0, H, N)
    F = rng.normal(size=(3, 3))
    g = coslam_amd.ncc_match_between_full(im1, x1, y1, im2, x2, y2, 0.3, F, 1e9, 0.3)
    b1, c1 = oracle.get_ncc_blocks(im1, x1, y1, 0.3)
    b2, c2 = oracle.get_ncc_blocks(im2, x2, y2, 0.3)
    assert np.array_equal(g["blocks1"], b1) and np.array_equal(g["blocks2"], b2) and np.array_equal(g["abc1"], c1)
    ones1, ones2 = np.ones(M, np.int32), np.ones(N, np.int32)
    e_o, n_o = oracle.ncc_epi_mat(F, x1, y1, b1, c1, ones1, x2, y2, b2, c2, ones2, 1e9, 0.3)
    assert np.array_equal(g["epi"], e_o) and np.array_equal(g["ncc"], n_o) and (n_o != -1).sum() > 50


@pytest.mark.parametrize("nccMin,cap", [(0.8, 4096), (0.3, 1 << 20), (0.3, 100)])
def test_epi_pairs_is_the_list_of_the_matrices_kept_entries(hip, nccMin, cap):
    """cs_ncc_epi_pairs_dev writes only the pairs that pass both tests of getEpiNccMat (reference src/slam/SL_FeatureMatching.cpp:
    24-31): the records must be exactly the non-(-1) entries of the dense matrices of cs_ncc_epi_mat_dev -- same doubles -- in any
    order; the counter counts every passing pair, also those a too-small list drops."""
    import torch

    from coslam_amd.ncc import NCC_PAIR_DTYPE, ncc_blocks_dev, ncc_epi_mat_dev, ncc_epi_pairs_dev

    W, H, scale, M, N = 640, 480, 0.3, 900, 777
    sc = Scene(2, W, H, 3000, seed=78)
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    side = []
    for c, n in ((0, M), (1, N)):
        im = sc.render(c, 0).astype(np.float64)
        ys = (np.arange(int(H * scale)) / scale).astype(int)
        xs = (np.arange(int(W * scale)) / scale).astype(int)
        small = torch.from_numpy(np.ascontiguousarray(im[np.ix_(ys, xs)]).astype(np.uint8)).to(dev)
        uv, vis = sc.project(c, 0)
        k = np.nonzero(vis)[0][: n // 2]
        x = torch.from_numpy(np.concatenate([uv[k, 0], rng.uniform(-20, W + 20, n - len(k))])).to(dev)
        y = torch.from_numpy(np.concatenate([uv[k, 1], rng.uniform(-20, H + 20, n - len(k))])).to(dev)
        blk = torch.zeros((n, 128), dtype=torch.uint8, device=dev)
        abc = torch.zeros((n, 4), dtype=torch.float64, device=dev)
        val = torch.zeros(n, dtype=torch.int32, device=dev)
        ncc_blocks_dev(s, small.data_ptr(), small.shape[1], small.shape[0], n, x.data_ptr(), y.data_ptr(), scale, blk.data_ptr(),
                       abc.data_ptr(), val.data_ptr())
        side.append((x, y, blk, abc, val, small))
    F = _fundamental(sc, 0, 1)
    (x1, y1, b1, a1, v1, _), (x2, y2, b2, a2, v2, _) = side
    epi = torch.zeros((M, N), dtype=torch.float64, device=dev)
    ncc = torch.zeros((M, N), dtype=torch.float64, device=dev)
    ncc_epi_mat_dev(s, F, M, x1.data_ptr(), y1.data_ptr(), b1.data_ptr(), a1.data_ptr(), v1.data_ptr(), N, x2.data_ptr(), y2.data_ptr(),
                    b2.data_ptr(), a2.data_ptr(), v2.data_ptr(), 50.0, nccMin, -1.0, epi.data_ptr(), ncc.data_ptr())
    pairs = torch.zeros(max(cap, 1) * NCC_PAIR_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    cnt = torch.full((1,), 12345, dtype=torch.int32, device=dev)
    ncc_epi_pairs_dev(s, F, M, x1.data_ptr(), y1.data_ptr(), b1.data_ptr(), a1.data_ptr(), v1.data_ptr(), N, x2.data_ptr(), y2.data_ptr(),
                      b2.data_ptr(), a2.data_ptr(), v2.data_ptr(), 50.0, nccMin, pairs.data_ptr(), cap, cnt.data_ptr())
    torch.cuda.synchronize()
    epi_h, ncc_h = epi.cpu().numpy(), ncc.cpu().numpy()
    kept = np.argwhere(ncc_h != -1.0)
    n = int(cnt.item())
    assert n == len(kept) and n > (1 if nccMin > 0.5 else 300)
    rec = pairs.cpu().numpy().view(NCC_PAIR_DTYPE)[: min(n, cap)]
    if n <= cap:
        order = np.lexsort((rec["j"], rec["i"]))
        rec = rec[order]
        assert np.array_equal(np.stack([rec["i"], rec["j"]], 1), kept)
        assert np.array_equal(rec["epi"], epi_h[kept[:, 0], kept[:, 1]]) and np.array_equal(rec["ncc"], ncc_h[kept[:, 0], kept[:, 1]])
    else:   # a list that is too small: every record it holds is a kept entry, the counter says how many there were
        assert len(rec) == cap
        assert np.array_equal(rec["epi"], epi_h[rec["i"], rec["j"]]) and np.array_equal(rec["ncc"], ncc_h[rec["i"], rec["j"]])
        assert len({(int(a), int(b)) for a, b in zip(rec["i"], rec["j"])}) == cap and np.all(ncc_h[rec["i"], rec["j"]] != -1.0)


def test_group_launches_give_what_the_per_camera_calls_give(hip):
    """cs_ncc_get_blocks_group_dev + cs_ncc_epi_pairs_group_dev (a whole matching run of three cameras in three launches) against
    cs_ncc_get_blocks_dev per camera + cs_ncc_epi_pairs_dev per camera pair: the same blocks, A / B / C and pair lists."""
    import torch

    from coslam_amd.ncc import (NCC_PAIR_DTYPE, ncc_epi_pairs_dev, ncc_epi_pairs_group_dev, ncc_get_blocks_dev, ncc_get_blocks_group_dev,
                                ncc_scaled_dims)

    W, H, n, nC, scale, cap = 640, 480, 1000, 3, 0.3, 1 << 16
    sc = Scene(nC, W, H, 3000, seed=79)
    rng = np.random.default_rng(6)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    ws_, hs_ = ncc_scaled_dims(W, H, scale)
    imgs, xs, ys, val = [], [], [], []
    for c in range(nC):
        imgs.append(torch.from_numpy(sc.render(c, 0)).to(dev))
        uv, vis = sc.project(c, 0)
        k = np.nonzero(vis)[0][: n // 2]
        xs.append(torch.from_numpy(np.concatenate([uv[k, 0], rng.uniform(-20, W + 20, n - len(k))])).to(dev))
        ys.append(torch.from_numpy(np.concatenate([uv[k, 1], rng.uniform(-20, H + 20, n - len(k))])).to(dev))
        val.append(torch.from_numpy((rng.random(n) < 0.8).astype(np.int32)).to(dev))

    def bufs():
        return ([torch.zeros(ws_ * hs_, dtype=torch.uint8, device=dev) for _ in range(nC)], [torch.zeros((n, 128), dtype=torch.uint8, device=dev) for _ in range(nC)],
                [torch.zeros((n, 4), dtype=torch.float64, device=dev) for _ in range(nC)], [torch.zeros(cap * 24, dtype=torch.uint8, device=dev) for _ in range(nC - 1)],
                torch.full((nC - 1,), 7, dtype=torch.int32, device=dev))

    Fm = [_fundamental(sc, c, c + 1) for c in range(nC - 1)]
    sm1, bl1, ab1, pr1, cn1 = bufs()
    for c in range(nC):
        ncc_get_blocks_dev(s, imgs[c].data_ptr(), W, H, n, xs[c].data_ptr(), ys[c].data_ptr(), scale, sm1[c].data_ptr(), bl1[c].data_ptr(),
                           ab1[c].data_ptr(), 0)
    for c in range(nC - 1):
        ncc_epi_pairs_dev(s, Fm[c], n, xs[c].data_ptr(), ys[c].data_ptr(), bl1[c].data_ptr(), ab1[c].data_ptr(), val[c].data_ptr(), n,
                          xs[c + 1].data_ptr(), ys[c + 1].data_ptr(), bl1[c + 1].data_ptr(), ab1[c + 1].data_ptr(), val[c + 1].data_ptr(), 50.0, 0.5,
                          pr1[c].data_ptr(), cap, cn1[c:c + 1].data_ptr())
    sm2, bl2, ab2, pr2, cn2 = bufs()
    cams = [dict(img=imgs[c].data_ptr(), x=xs[c].data_ptr(), y=ys[c].data_ptr(), scaled=sm2[c].data_ptr(), blocks=bl2[c].data_ptr(),
                 abc=ab2[c].data_ptr(), valid=val[c].data_ptr()) for c in range(nC)]
    ncc_get_blocks_group_dev(s, cams, W, H, n, scale)
    ncc_epi_pairs_group_dev(s, cams, n, [dict(F=Fm[c], camA=c, camB=c + 1, pairs=pr2[c].data_ptr(), count=cn2[c:c + 1].data_ptr())
                                         for c in range(nC - 1)], 50.0, 0.5, cap)
    torch.cuda.synchronize()
    for c in range(nC):
        assert torch.equal(sm1[c], sm2[c]) and torch.equal(bl1[c], bl2[c]) and torch.equal(ab1[c].view(torch.int64), ab2[c].view(torch.int64))
        assert int(val[c].sum()) < n   # (the mask is the caller's: the group cutter leaves it alone)
    c1, c2 = cn1.cpu().numpy(), cn2.cpu().numpy()
    assert np.array_equal(c1, c2) and c1.min() > 20 and c1.max() < cap
    for c in range(nC - 1):
        a = np.sort(pr1[c].cpu().numpy().view(NCC_PAIR_DTYPE)[: c1[c]], order=("i", "j"))
        b = np.sort(pr2[c].cpu().numpy().view(NCC_PAIR_DTYPE)[: c2[c]], order=("i", "j"))
        assert np.array_equal(a, b)


def test_fundamental_matrices_from_the_poses_on_the_device(hip):
    """cs_ncc_fmats_dev (NewMapPtsNCC::matchBetween forms E and F from the cameras' CURRENT poses, reference
    src/app/SL_NewMapPointsInterCam.cpp:284-292 -- the loop must not hand it matrices of the ground truth): F of the consecutive camera
    pairs from poses in device memory, against x_A = R x_B + t, E = [t]x R, F = K^-T E K^-1 in numpy (<= 1e-12 of the matrix's size);
    and a matching run whose jobs point at those matrices (dF) gives the pair lists of a run handed the same numbers by value."""
    import torch

    from coslam_amd.ncc import NCC_PAIR_DTYPE, ncc_epi_pairs_group_dev, ncc_fmats_dev, ncc_get_blocks_group_dev, ncc_scaled_dims

    W, H, n, nC, scale, cap = 640, 480, 800, 4, 0.3, 1 << 16
    sc = Scene(nC, W, H, 3000, seed=41)
    rng = np.random.default_rng(8)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    R = np.stack([sc.pose(c, 7)[0].ravel() for c in range(nC)])
    t = np.stack([sc.pose(c, 7)[1] for c in range(nC)]) + rng.normal(0, 0.01, (nC, 3))     # (estimated poses: not the truth's)
    iK = np.linalg.inv(sc.K)
    d_R, d_t, d_iK = torch.from_numpy(R.copy()).to(dev), torch.from_numpy(t.copy()).to(dev), torch.from_numpy(iK.ravel().copy()).to(dev)
    d_F = torch.zeros((nC - 1, 9), dtype=torch.float64, device=dev)
    ncc_fmats_dev(s, nC, list(range(nC - 1)), list(range(1, nC)), [d_iK.data_ptr()] * nC, d_R.data_ptr(), d_t.data_ptr(), d_F.data_ptr())
    torch.cuda.synchronize()
    F_dev = d_F.cpu().numpy()
    for c in range(nC - 1):
        Ra, Rb = R[c].reshape(3, 3), R[c + 1].reshape(3, 3)
        Rr = Ra @ Rb.T
        tr = t[c] - Rr @ t[c + 1]
        E = np.array([[0, -tr[2], tr[1]], [tr[2], 0, -tr[0]], [-tr[1], tr[0], 0]]) @ Rr
        Fw = iK.T @ E @ iK
        assert np.abs(F_dev[c].reshape(3, 3) - Fw).max() <= 1e-12 * np.abs(Fw).max(), c
    ws_, hs_ = ncc_scaled_dims(W, H, scale)
    imgs = [torch.from_numpy(sc.render(c, 7)).to(dev) for c in range(nC)]
    xs, ys, val = [], [], []
    for c in range(nC):
        uv, vis = sc.project(c, 7)
        k = np.nonzero(vis)[0][:n]
        xs.append(torch.from_numpy(np.ascontiguousarray(uv[k, 0])).to(dev)), ys.append(torch.from_numpy(np.ascontiguousarray(uv[k, 1])).to(dev))
        val.append(torch.ones(n, dtype=torch.int32, device=dev))
    sm = [torch.zeros(ws_ * hs_, dtype=torch.uint8, device=dev) for _ in range(nC)]
    bl, ab = [torch.zeros((n, 128), dtype=torch.uint8, device=dev) for _ in range(nC)], [torch.zeros((n, 4), dtype=torch.float64, device=dev) for _ in range(nC)]
    cams = [dict(img=imgs[c].data_ptr(), x=xs[c].data_ptr(), y=ys[c].data_ptr(), scaled=sm[c].data_ptr(), blocks=bl[c].data_ptr(), abc=ab[c].data_ptr(),
                 valid=val[c].data_ptr()) for c in range(nC)]
    ncc_get_blocks_group_dev(s, cams, W, H, n, scale)
    out = []
    for by_value in (True, False):
        pr = [torch.zeros(cap * 24, dtype=torch.uint8, device=dev) for _ in range(nC - 1)]
        cn = torch.zeros(nC - 1, dtype=torch.int32, device=dev)
        jobs = [dict(camA=c, camB=c + 1, pairs=pr[c].data_ptr(), count=cn[c:c + 1].data_ptr(),
                     **(dict(F=F_dev[c]) if by_value else dict(dF=d_F[c].data_ptr()))) for c in range(nC - 1)]
        ncc_epi_pairs_group_dev(s, cams, n, jobs, 50.0, 0.5, cap)
        torch.cuda.synchronize()
        c_ = cn.cpu().numpy()
        out.append([np.sort(pr[c].cpu().numpy().view(NCC_PAIR_DTYPE)[: c_[c]], order=("i", "j")) for c in range(nC - 1)])
        assert c_.min() > 20
    for a, b in zip(*out):
        assert np.array_equal(a, b)
