"""GPU parity tests of the KLT path: libcoslam_hip.so (through the C-ABI) vs the oracle on the same
seeded synthetic inputs.  Bit-exact for the pyramid, the cornerness map, the detection set and the
slot tables (binary16/32 with a fixed evaluation order).  The gain tracker (CoSLAM's default) is ALSO
bit-exact -- positions, gains and status flags -- against the oracle in its "tree" summation mode, which
takes the window sums in the HIP kernel's fixed order (oracle/klt_oracle.c okl_track_gain_pass_tree);
against the shader's serial order the same results are within 0.02 px (tolerance from SURVEY.md 8d), and every
status flag that differs from the serial order is checked, slot by slot ON THE GPU's output, to belong to a slot that
sits within 1 % of one of the thresholds the shader tests (the oracle records that margin per slot:
okl_set_threshold_margin_buffer).  The no-gain tracker is bit-exact against its own tree mode as well
(okl_set_nogain_sum_mode: the kernel's lane / fold order).  Only gain-tracker windows wider than 15 (one wave per
feature, wave-wide folds) keep the 0.02 px tolerance against the default-order oracle."""
import os

import numpy as np
import pytest

import coslam_amd
import oracle
from coslam_amd.synth import Scene, blob_image, shift_image

pytestmark = pytest.mark.gpu

TOL_PX = 0.02


def exact_mode(cfg):
    """the gain tracker with a window the rows kernel covers, and the no-gain tracker: the oracle's tree mode (window sums in
    the HIP kernels' fixed order) must match bit for bit"""
    return (not cfg.trackWithGain) or 1 <= cfg.windowWidth // 2 <= 7


def make_pair(cfg, W, H, L, fw, fh, tap_mode=0, sum_mode=None):
    trk = coslam_amd.KLT_SequenceTracker(cfg, device=0, tap_mode=tap_mode)
    trk.allocate(W, H, L, fw, fh)
    if sum_mode is None:
        sum_mode = 1 if exact_mode(cfg) else 0
    ora = oracle.SequenceTracker(cfg, centered=tap_mode, sum_mode=sum_mode)
    ora.allocate(W, H, L, fw, fh)
    return trk, ora


def assert_dest_exact(d_g, d_o, what=""):
    assert np.array_equal(d_g["status"], d_o["status"]), f"{what}: {(d_g['status'] != d_o['status']).sum()} status flags differ"
    live = d_o["status"] >= 0
    assert np.array_equal(d_g["pos"][live], d_o["pos"][live]), f"{what}: positions differ"
    assert np.array_equal(d_g["gain"][live], d_o["gain"][live]), f"{what}: gains differ"


def cfg2(**kw):
    base = dict(nIterations=10, nLevels=4, levelSkip=1, windowWidth=7, trackWithGain=1, minCornerness=3000.0,
                convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=5)
    base.update(kw)
    return coslam_amd.KLT_SequenceTrackerConfig(**base)


def assert_serial_order_differs_only_at_thresholds(d_g, d_s, margin, W, H, what=""):
    """GPU result vs the oracle in the SHADER's serial summation order, both started from identical state: positions of
    the slots both track within 0.02 px, and a slot tracked by one and invalidated by the other must sit within 1 % of a
    threshold the shader tests (det, SSD, |dX|^2, valid region) -- `margin` is what the serial run recorded per slot."""
    both = (d_g["status"] == 0) & (d_s["status"] == 0)
    either = (d_g["status"] == 0) | (d_s["status"] == 0)
    err = np.abs(d_g["pos"][both] - d_s["pos"][both]) * np.array([W, H], dtype=np.float32)
    assert both.sum() > 0 and err.max() <= TOL_PX, (what, err.max() if err.size else None)
    differ = either & ~both
    assert np.all(margin[differ] < 0.01), (what, "status differs away from every threshold", margin[differ])
    assert differ.sum() <= 0.01 * max(either.sum(), 1) + 1, (what, int(differ.sum()), int(either.sum()))
    return int(differ.sum())


def serial_track_with_margins(cfg, W, H, levels, fw, fh, im0, im1, tap_mode=0):
    ser = oracle.SequenceTracker(cfg, centered=tap_mode, sum_mode=0)
    ser.allocate(W, H, levels, fw, fh)
    ser.detect(im0)
    ser.advanceFrame()
    margin = np.full(fw * fh, 1e30, np.float32)
    oracle.set_threshold_margin_buffer(margin)
    try:
        _, d_s = ser.track(im1)
    finally:
        oracle.set_threshold_margin_buffer(None)
    ser.close()
    return d_s, margin


def compare_dest(d_g, d_o, W, H, min_same=1.0, tol=TOL_PX):
    same = d_g["status"] == d_o["status"]
    assert same.mean() >= min_same, f"status agreement {same.mean():.4f} < {min_same}"
    live = same & (d_o["status"] >= 0)
    err = np.abs(d_g["pos"][live] - d_o["pos"][live]) * np.array([W, H], dtype=np.float32)
    return same, live, (err.max() if err.size else 0.0)


@pytest.mark.parametrize("W,H,L,tap", [(640, 480, 4, 0), (640, 480, 4, 1), (322, 250, 3, 0), (1920, 1080, 4, 0),
                                       (96, 64, 1, 0), (640, 480, 6, 0),
                                       # odd sizes at every level, ragged last tiles of the fused kernels, 2 / 5 / 6 levels
                                       (333, 241, 5, 0), (333, 241, 5, 1), (101, 67, 4, 0), (101, 67, 4, 1), (64, 48, 2, 0),
                                       (1000, 562, 6, 0), (1279, 719, 4, 1), (130, 33, 3, 0)])
def test_pyramid_bitexact(hip, W, H, L, tap):
    rng = np.random.default_rng(W * 7 + H + L)
    img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    trk = coslam_amd.KLT_SequenceTracker(cfg2(nLevels=L), 0, tap)
    trk.allocate(W, H, L, 8, 8)
    trk.build_pyramid(img)
    got = trk.read_pyramid(1)
    want = oracle.pyramid_build(img, W, H, L, centered=tap)
    for l in range(L):
        g = trk.level_view(got, l)
        w = oracle.level_view(want, W, H, L, l)
        assert np.array_equal(g, w), f"level {l}: {np.sum(g != w)} halfs differ"
    trk.close()


def test_pyramid_edge_images(hip):
    W, H, L = 128, 96, 3
    trk = coslam_amd.KLT_SequenceTracker(cfg2(nLevels=L), 0, 0)
    trk.allocate(W, H, L, 8, 8)
    for img in (np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8),
                (np.indices((H, W)).sum(0) % 2 * 255).astype(np.uint8)):
        trk.build_pyramid(img)
        assert np.array_equal(trk.read_pyramid(1), oracle.pyramid_build(img, W, H, L))
    trk.close()


@pytest.mark.parametrize("W,H,N,gain", [(640, 480, (50, 40), 1), (640, 480, (32, 32), 0), (322, 250, (16, 16), 1)])
def test_detect_bitexact(hip, W, H, N, gain):
    sc = Scene(1, W, H, 4000 if W >= 640 else 1200, seed=11)
    img = sc.render(0, 0)
    cfg = cfg2(trackWithGain=gain, nLevels=3)
    trk, ora = make_pair(cfg, W, H, 3, *N)
    n_g, d_g = trk.detect(img)
    n_o, d_o = ora.detect(img)
    assert np.array_equal(trk.read_cornerness(), ora.read_cornerness()), "non-max cornerness map differs"
    assert n_g == n_o and n_g > 100
    assert np.array_equal(d_g["status"], d_o["status"])
    live = d_o["status"] >= 0
    for f in ("pos", "gain", "fed"):
        assert np.array_equal(d_g[f][live], d_o[f][live]), f
    assert np.array_equal(trk.read_features(), ora.read_features())
    trk.close()


def test_detect_topk_when_more_corners_than_slots(hip):
    W, H = 640, 480
    sc = Scene(1, W, H, 5000, seed=5)
    img = sc.render(0, 0)
    cfg = cfg2(minDistance=3, minCornerness=500.0, nLevels=2)
    trk, ora = make_pair(cfg, W, H, 2, 16, 16)  # 256 slots, thousands of corners
    n_g, d_g = trk.detect(img)
    n_o, d_o = ora.detect(img)
    assert n_g == n_o == 256
    assert np.array_equal(d_g["pos"], d_o["pos"]) and np.array_equal(d_g["gain"], d_o["gain"])
    trk.close()


def test_detect_with_present_points(hip):
    W, H = 320, 240
    sc = Scene(1, W, H, 900, seed=3)
    img = sc.render(0, 0)
    rng = np.random.default_rng(0)
    present = np.zeros((40, 3), np.float32)
    present[:, :2] = rng.uniform(0.1, 0.9, (40, 2))
    cfg = cfg2(nLevels=2, minCornerness=1000.0)
    trk, ora = make_pair(cfg, W, H, 2, 20, 20)
    n_g, d_g = trk.detect(img, present)
    n_o, d_o = ora.detect(img, present)
    assert n_g == n_o
    assert np.array_equal(d_g["status"], d_o["status"])
    live = d_o["status"] >= 0
    for f in ("pos", "gain", "fed"):
        assert np.array_equal(d_g[f][live], d_o[f][live]), f
    assert (d_g["fed"][live] >= 0).sum() == 40
    assert np.array_equal(trk.read_features(), ora.read_features())
    trk.close()


def test_empty_image_detects_nothing(hip):
    W, H = 128, 96
    cfg = cfg2(nLevels=2)
    trk, ora = make_pair(cfg, W, H, 2, 8, 8)
    img = np.full((H, W), 77, np.uint8)
    n_g, d_g = trk.detect(img)
    assert n_g == 0 and np.all(d_g["status"] == -1)
    trk.advanceFrame()
    n_g, d_g = trk.redetect(img)
    assert n_g == 0 and np.all(d_g["status"] == -1)
    trk.close()


@pytest.mark.parametrize("gain,levels,skip,win", [(1, 4, 1, 7), (0, 4, 1, 7), (1, 6, 2, 6), (0, 3, 2, 5), (0, 4, 1, 11),
                                                  (1, 4, 1, 13), (1, 3, 1, 5), (1, 3, 1, 3), (1, 4, 1, 9), (1, 3, 1, 15),
                                                  (1, 3, 1, 17)])
def test_track_parity(hip, gain, levels, skip, win):
    W, H, fw, fh = 640, 480, 50, 40
    sc = Scene(1, W, H, 4000, seed=21)
    im0, im1 = sc.render(0, 0), sc.render(0, 1)
    cfg = cfg2(trackWithGain=gain, nLevels=levels, levelSkip=skip, windowWidth=win)
    trk, ora = make_pair(cfg, W, H, levels, fw, fh)
    n_g, _ = trk.detect(im0)
    n_o, _ = ora.detect(im0)
    assert n_g == n_o
    trk.advanceFrame()
    ora.advanceFrame()
    n_g, d_g = trk.track(im1)
    n_o, d_o = ora.track(im1)
    if exact_mode(cfg):   # tree-mode oracle: bit for bit
        assert n_g == n_o
        assert_dest_exact(d_g, d_o, "track")
        assert np.array_equal(trk.read_features(), ora.read_features())
        assert (d_o["status"] == 0).sum() > 200
        # and against the shader's serial summation order: <= 0.02 px, every status difference at a threshold
        d_s, margin = serial_track_with_margins(cfg, W, H, levels, fw, fh, im0, im1)
        assert_serial_order_differs_only_at_thresholds(d_g, d_s, margin, W, H, f"gain {gain} window {win}")
    else:
        same, live, emax = compare_dest(d_g, d_o, W, H, min_same=0.995)
        assert live.sum() > 200
        assert emax <= TOL_PX, emax
        if gain:
            assert np.max(np.abs(d_g["gain"][live] - d_o["gain"][live])) < 1e-3
        assert abs(n_g - n_o) <= (~same).sum()
    trk.close()


@pytest.mark.parametrize("case", range(int(os.environ.get("COSLAM_TEST_CASES", "10"))))
def test_random_configurations_match_the_oracle(hip, case):
    """Seeded random tracker configurations (image size, levels, skip, window, iterations, slot grid, thresholds, gain on /
    off): detect + redetect + track against the oracle, same tolerances as the fixed-configuration tests."""
    rng = np.random.default_rng(500 + case)
    W, H = int(rng.integers(180, 420)), int(rng.integers(140, 320))
    levels = int(rng.integers(2, 5))
    while (min(W, H) >> (levels - 1)) < 12:
        levels -= 1
    cfg = cfg2(trackWithGain=int(rng.integers(0, 2)), nLevels=levels, levelSkip=int(rng.integers(1, 3)),
               windowWidth=int(rng.choice([5, 6, 7, 9, 11])), nIterations=int(rng.integers(1, 13)),
               minDistance=int(rng.integers(3, 10)), minCornerness=float(rng.choice([800.0, 1500.0, 3000.0])),
               convergenceThreshold=float(rng.choice([0.5, 1.0, 2.0])), SSD_Threshold=float(rng.choice([5000.0, 20000.0])))
    fw, fh = int(rng.integers(6, 26)), int(rng.integers(5, 22))
    sc = Scene(1, W, H, 1500, seed=700 + case, sigma=1.4)
    trk, ora = make_pair(cfg, W, H, levels, fw, fh)
    img = sc.render(0, 0)
    n_g, d_g = trk.detect(img)
    n_o, d_o = ora.detect(img)
    assert n_g == n_o and np.array_equal(d_g["status"], d_o["status"])
    assert np.array_equal(d_g["pos"][d_o["status"] >= 0], d_o["pos"][d_o["status"] >= 0])
    trk.advanceFrame()
    ora.advanceFrame()
    for f, call in ((1, "redetect"), (2, "track"), (3, "redetect")):
        img = sc.render(0, f)
        n_g, d_g = getattr(trk, call)(img)
        n_o, d_o = getattr(ora, call)(img)
        if exact_mode(cfg):
            assert n_g == n_o, (case, f, call)
            assert_dest_exact(d_g, d_o, f"case {case} frame {f} {call}")
        else:
            same, live, emax = compare_dest(d_g, d_o, W, H, min_same=0.98)
            assert emax <= TOL_PX * f, (case, f, call, emax)
            assert abs(n_g - n_o) <= 2 * (~same).sum() + 2, (case, f, call)
        trk.advanceFrame()
        ora.advanceFrame()
    trk.close()


@pytest.mark.parametrize("gain", [1, 0])
def test_redetect_sequence(hip, gain):
    """The reference's per-frame call: redetect + advanceFrame (GPUKLT::next, GPUKLT.cpp:144-161)."""
    W, H, fw, fh = 640, 480, 50, 40
    sc = Scene(1, W, H, 4000, seed=33)
    cfg = cfg2(trackWithGain=gain)
    trk, ora = make_pair(cfg, W, H, 4, fw, fh)
    img = sc.render(0, 0)
    assert trk.detect(img)[0] == ora.detect(img)[0]
    trk.advanceFrame()
    ora.advanceFrame()
    for f in range(1, 6):
        img = sc.render(0, f)
        n_g, d_g = trk.redetect(img)
        n_o, d_o = ora.redetect(img)
        # tree-mode oracle (both trackers): every frame bit for bit, no drift to bound
        assert n_g == n_o
        assert_dest_exact(d_g, d_o, f"frame {f}")
        assert np.array_equal(trk.read_features(), ora.read_features())
        trk.advanceFrame()
        ora.advanceFrame()
    assert (d_o["status"] == 0).sum() > 300
    trk.close()


def test_known_shift_is_recovered(hip):
    """Known-answer: a pure translation of the image must come back as the flow of every tracked feature."""
    W, H = 320, 240
    sc = Scene(1, W, H, 700, seed=8, sigma=1.6)
    im0 = sc.render(0, 0)
    dx, dy = 1.3, -0.7
    im1 = shift_image(im0, dx, dy)
    cfg = cfg2(nLevels=3, trackWithGain=0, minCornerness=1500.0)
    trk = coslam_amd.KLT_SequenceTracker(cfg, 0)
    trk.allocate(W, H, 3, 20, 20)
    n0, d0 = trk.detect(im0)
    trk.advanceFrame()
    n1, d1 = trk.track(im1)
    ok = d1["status"] == 0
    assert ok.sum() > 0.8 * n0
    flow = (d1["pos"][ok] - d0["pos"][ok]) * np.array([W, H], np.float32)
    assert np.median(np.abs(flow[:, 0] - dx)) < 0.05 and np.median(np.abs(flow[:, 1] - dy)) < 0.05
    trk.close()


@pytest.mark.parametrize("gain", [0, 1])
def test_affine_warp_flow_is_recovered(hip, gain):
    """Known-answer: the analytic flow of an affine warp (1 % zoom, 0.35 degrees, a shift), HIP path."""
    from scipy.ndimage import affine_transform

    W, H = 320, 240
    sc = Scene(1, W, H, 700, seed=11, sigma=1.6)
    im0 = sc.render(0, 0)
    th, zoom, t = np.deg2rad(0.35), 1.01, np.array([0.6, -0.4])
    A = zoom * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    c0 = np.array([(W - 1) / 2.0, (H - 1) / 2.0])
    Ai = np.linalg.inv(A)
    off_xy = c0 - Ai @ (c0 + t)
    im1 = affine_transform(im0.astype(np.float64), Ai[::-1, ::-1], offset=off_xy[::-1], order=3, mode="nearest")
    im1 = np.clip(np.rint(im1), 0, 255).astype(np.uint8)
    trk = coslam_amd.KLT_SequenceTracker(cfg2(nLevels=3, trackWithGain=gain, minCornerness=1500.0), 0)
    trk.allocate(W, H, 3, 20, 20)
    n0, d0 = trk.detect(im0)
    trk.advanceFrame()
    n1, d1 = trk.track(im1)
    ok = d1["status"] == 0
    assert ok.sum() > 0.8 * n0
    q0 = d0["pos"][ok].astype(np.float64) * [W, H] - 0.5
    flow = (d1["pos"][ok].astype(np.float64) - d0["pos"][ok]) * [W, H]
    want = (q0 - c0) @ A.T + c0 + t - q0
    err = np.linalg.norm(flow - want, axis=1)
    assert np.median(err) < 0.1 and np.percentile(err, 90) < 0.25
    trk.close()


def test_feed_extern_points_matches_oracle(hip):
    W, H = 320, 240
    sc = Scene(1, W, H, 600, seed=4)
    img = sc.render(0, 0)
    cfg = cfg2(nLevels=2, minCornerness=1500.0)
    trk, ora = make_pair(cfg, W, H, 2, 16, 16)
    trk.detect(img)
    ora.detect(img)
    trk.advanceFrame()
    ora.advanceFrame()
    pts = np.zeros((12, 3), np.float32)
    pts[:, :2] = np.random.default_rng(2).uniform(0.2, 0.8, (12, 2))
    n_g, ids_g = trk.feedExternFeaturePoints(pts)
    n_o, ids_o = ora.feedExternFeaturePoints(pts)
    assert n_g == n_o and np.array_equal(ids_g, ids_o)
    assert np.array_equal(trk.read_features(), ora.read_features())
    trk.close()


@pytest.mark.parametrize("gain,npts", [(1, 40), (0, 700), (1, 0)])
def test_feed_on_the_device_matches_oracle(hip, gain, npts):
    """cs_klt_feed_dev: the list in HBM, one launch; more points than free slots, points on top of tracked features (the stride-2
    read of the stride-3 list decides which), none at all."""
    import torch

    W, H = 320, 240
    sc = Scene(1, W, H, 900, seed=9)
    img = sc.render(0, 0)
    cfg = cfg2(nLevels=2, minCornerness=800.0, trackWithGain=gain)
    trk, ora = make_pair(cfg, W, H, 2, 24, 20)
    trk.detect(img)
    ora.detect(img)
    trk.advanceFrame()
    ora.advanceFrame()
    feats = ora.read_features().reshape(-1, 3)
    live = feats[feats[:, 0] >= 0]
    rng = np.random.default_rng(5)
    pts = np.zeros((npts, 3), np.float32)
    if npts:
        pts[:, :2] = rng.uniform(0.1, 0.9, (npts, 2))
        # the distance loop reads floats 2k, 2k+1 of the flat list: put tracked features' positions THERE
        flat = pts.reshape(-1)
        for k in range(0, min(npts, 2 * (len(live) // 2)), 3):
            flat[2 * k], flat[2 * k + 1] = live[k % len(live), 0], live[k % len(live), 1]
    n_o, ids_o = ora.feedExternFeaturePoints(pts.copy())
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(pts.reshape(-1).copy()).to(dev) if npts else torch.zeros(3, dtype=torch.float32, device=dev)
    d_ids = torch.full((max(npts, 1),), -1, dtype=torch.int32, device=dev)
    d_n = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    trk.feed_dev(npts, d_pts.data_ptr(), d_ids.data_ptr(), d_n.data_ptr())
    got = trk.read_features()
    n_g = int(d_n.item())
    assert n_g == n_o and np.array_equal(d_ids.cpu().numpy()[:n_g], ids_o)
    if npts == 700:
        assert n_g < npts          # the list has 480 slots
    assert np.array_equal(got, ora.read_features())
    trk.close()


def test_device_resident_entry_points(hip):
    """*_dev variants: image and results stay in HBM; same answer as the host-pointer variants."""
    import torch

    W, H, fw, fh = 640, 480, 50, 40
    sc = Scene(1, W, H, 4000, seed=33)
    cfg = cfg2()
    a = coslam_amd.KLT_SequenceTracker(cfg, 0)
    a.allocate(W, H, 4, fw, fh)
    b = coslam_amd.KLT_SequenceTracker(cfg, 0)
    b.allocate(W, H, 4, fw, fh)
    dev = torch.device("cuda:0")
    d_dest = torch.zeros(fw * fh * 5, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
    b.set_stream(torch.cuda.current_stream().cuda_stream)
    for f in range(3):
        img = sc.render(0, f)
        d_img = torch.from_numpy(img).to(dev)
        if f == 0:
            n_a, dest_a = a.detect(img)
            b.detect_dev(d_img.data_ptr(), d_dest.data_ptr(), d_counts.data_ptr())
        else:
            n_a, dest_a = a.redetect(img)
            b.redetect_dev(d_img.data_ptr(), d_dest.data_ptr(), d_counts.data_ptr())
        torch.cuda.synchronize()
        dest_b = d_dest.cpu().numpy().view(coslam_amd.KLT_TrackedFeature)
        assert int(d_counts[0]) == n_a
        assert np.array_equal(dest_b["status"], dest_a["status"])
        live = dest_a["status"] >= 0
        assert np.array_equal(dest_b["pos"][live], dest_a["pos"][live])
        a.advanceFrame()
        b.advanceFrame()
    a.close()
    b.close()


def test_graph_replay_matches_eager(hip):
    """hipGraph replay of the frame schedule (all buffer-rotation states) gives the eager results bit for bit."""
    import torch

    W, H, fw, fh = 640, 480, 50, 40
    sc = Scene(1, W, H, 4000, seed=44)
    cfg = cfg2()
    dev = torch.device("cuda:0")
    res = []
    for graphs in (False, True):
        t = coslam_amd.KLT_SequenceTracker(cfg, 0)
        t.allocate(W, H, 4, fw, fh)
        t.set_stream(torch.cuda.current_stream().cuda_stream)
        t.enable_graphs(graphs)
        d_dest = torch.zeros(fw * fh * 5, dtype=torch.int32, device=dev)
        d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
        out = []
        for f in range(14):  # > 2 full rotation periods (6)
            d_img = torch.from_numpy(sc.render(0, f % 7)).to(dev)
            if f == 0:
                t.detect_dev(d_img.data_ptr(), d_dest.data_ptr(), d_counts.data_ptr())
            else:
                t.redetect_dev(d_img.data_ptr(), d_dest.data_ptr(), d_counts.data_ptr())
            t.advanceFrame()
            torch.cuda.synchronize()
            out.append((d_dest.cpu().numpy().copy(), d_counts.cpu().numpy().copy()))
        res.append(out)
        t.close()
    for (da, ca), (db, cb) in zip(*res):
        assert np.array_equal(ca, cb)
        fa, fb = da.view(coslam_amd.KLT_TrackedFeature), db.view(coslam_amd.KLT_TrackedFeature)
        assert np.array_equal(fa["status"], fb["status"])
        live = fa["status"] >= 0
        assert np.array_equal(fa["pos"][live], fb["pos"][live])


@pytest.mark.parametrize("W,H,levels,fw,fh,mind", [(640, 480, 4, 50, 40, 5), (640, 480, 6, 50, 40, 5), (333, 241, 3, 20, 15, 5),
                                                   (1920, 1080, 4, 100, 50, 5), (640, 480, 4, 20, 15, 40)])
def test_frame_front_prefetch_matches_unprefetched(hip, W, H, levels, fw, fh, mind):
    """cs_klt_prefetch_dev (next frame's pyramid + cornerness built by this frame's detector-tail launches, third pyramid
    buffer, second cornerness map) gives the unprefetched results bit for bit -- also when a prefetch names the wrong
    image, when track-only frames sit in between, and with no host synchronisation between frames."""
    import torch

    nf = 9 if W <= 640 else 4   # distinct frames (rendering 1080p scenes on the host is the slow part)
    sc = Scene(1, W, H, 4000 if W <= 640 else 12000, seed=52)
    cfg = cfg2(nLevels=levels, minDistance=mind)  # minDistance 40: the non-max tile outgrows the fused tail's LDS
    dev = torch.device("cuda:0")
    frames = [torch.from_numpy(sc.render(0, f)).to(dev) for f in range(nf)]
    n_steps = 16
    plan = ["detect"] + ["redetect"] * 6 + ["track", "track"] + ["redetect"] * 7
    res = []
    for prefetch in (False, True):
        t = coslam_amd.KLT_SequenceTracker(cfg, 0)
        t.allocate(W, H, levels, fw, fh)
        t.set_stream(torch.cuda.current_stream().cuda_stream)
        d_dests = [torch.zeros(fw * fh * 5, dtype=torch.int32, device=dev) for _ in range(n_steps)]
        d_counts = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(n_steps)]
        for f in range(n_steps):
            img = frames[f % nf]
            fn = {"detect": t.detect_dev, "redetect": t.redetect_dev, "track": t.track_dev}[plan[f]]
            if prefetch and f + 1 < n_steps:
                # f == 4: prefetch the WRONG image -- must be ignored by the next call
                t.prefetch_dev(frames[(f + 1 + (2 if f == 4 else 0)) % nf].data_ptr())
            fn(img.data_ptr(), d_dests[f].data_ptr(), d_counts[f].data_ptr())
            t.advanceFrame()
        torch.cuda.synchronize()
        res.append([(d.cpu().numpy().copy(), c.cpu().numpy().copy()) for d, c in zip(d_dests, d_counts)])
        t.close()
    for f, ((da, ca), (db, cb)) in enumerate(zip(*res)):
        assert np.array_equal(ca, cb), f
        fa, fb = da.view(coslam_amd.KLT_TrackedFeature), db.view(coslam_amd.KLT_TrackedFeature)
        assert np.array_equal(fa["status"], fb["status"]), f
        live = fa["status"] >= 0
        assert np.array_equal(fa["pos"][live], fb["pos"][live]), f
        assert np.array_equal(fa["gain"][live], fb["gain"][live]), f


@pytest.mark.parametrize("levels,skip,iters,win,grid", [(4, 1, 10, 7, (50, 40)), (6, 2, 12, 6, (32, 32)), (3, 1, 1, 7, (20, 15)),
                                                       (3, 1, 3, 11, (20, 20)), (2, 1, 5, 7, (7, 5)),
                                                       (4, 1, 10, 7, (64, 17)), (4, 1, 10, 7, (37, 53)),
                                                       (4, 1, 10, 7, (100, 50)), (4, 1, 10, 7, (90, 30))])
def test_persistent_gain_tracker_is_bit_identical_to_per_pass_launches(hip, levels, skip, iters, win, grid):
    """One persistent launch with granule hand-offs == the reference's one-launch-per-pass Jacobi schedule."""
    W, H = 640, 480
    sc = Scene(1, W, H, 4000, seed=51)
    cfg = cfg2(nLevels=levels, levelSkip=skip, nIterations=iters, windowWidth=win)
    out = []
    for fused in (0, 1):
        t = coslam_amd.KLT_SequenceTracker(cfg, 0)
        t.allocate(W, H, levels, *grid)
        t.set_fused(fused)
        t.detect(sc.render(0, 0))
        t.advanceFrame()
        frames = []
        for f in range(1, 5):
            n, d = (t.redetect if f % 2 else t.track)(sc.render(0, f))  # track-only frames exercise the stale buffer
            frames.append((n, d.copy(), t.read_features().copy()))
            t.advanceFrame()
        out.append(frames)
        t.close()
    for (n0, d0, f0), (n1, d1, f1) in zip(*out):
        assert n0 == n1
        assert np.array_equal(d0["status"], d1["status"])
        live = d0["status"] >= 0
        assert np.array_equal(d0["pos"][live], d1["pos"][live]) and np.array_equal(d0["gain"][live], d1["gain"][live])
        assert np.array_equal(f0, f1)


def test_persistent_gain_tracker_with_asymmetric_neighbour_offsets(hip):
    """A 100 x 50 slot grid makes the reference's betaN1 offsets asymmetric ((1,1) without (-1,-1)): a wave may run
    passes ahead of a slot that reads it.  Every granule row is written once per frame under a frame-unique tag, so the
    result must still be the per-pass schedule's, bit for bit, frame after frame (1080p, 5000 slots, 20 frames)."""
    import torch

    W, H, fw, fh = 1920, 1080, 100, 50
    sc = Scene(1, W, H, 12000, seed=52)
    cfg = cfg2()
    dev = torch.device("cuda:0")
    frames = [torch.from_numpy(sc.render(0, f)).to(dev) for f in range(4)]
    res = []
    for fused in (0, 1, 1):
        t = coslam_amd.KLT_SequenceTracker(cfg, 0)
        t.allocate(W, H, 4, fw, fh)
        t.set_stream(torch.cuda.current_stream().cuda_stream)
        t.set_fused(fused)
        n = 21
        d_dests = [torch.zeros(fw * fh * 5, dtype=torch.int32, device=dev) for _ in range(n)]
        d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
        for f in range(n):
            (t.detect_dev if f == 0 else t.redetect_dev)(frames[f % 4].data_ptr(), d_dests[f].data_ptr(), d_counts.data_ptr())
            t.advanceFrame()
        torch.cuda.synchronize()
        res.append([d.cpu().numpy().view(coslam_amd.KLT_TrackedFeature).copy() for d in d_dests])
        t.close()
    for other in (1, 2):
        for f, (a, b) in enumerate(zip(res[0], res[other])):
            assert np.array_equal(a["status"], b["status"]), f
            live = a["status"] >= 0
            assert np.array_equal(a["pos"][live], b["pos"][live]), f
            assert np.array_equal(a["gain"][live], b["gain"][live]), f


def test_persistent_gain_tracker_under_uneven_foreign_load(hip):
    """The granule hand-off must not depend on timing or placement: with other streams hammering the chip (large
    GEMMs, copies, many tiny launches: uneven load, L1-warm consumers) every frame of the persistent tracker must still
    be bit-identical to the per-pass schedule run on a quiet chip."""
    import torch

    W, H, L, grid = 640, 480, 4, (50, 40)
    sc = Scene(1, W, H, 6000, seed=77)
    cfg = cfg2()
    frames = [sc.render(0, f) for f in range(9)]

    def run(fused, load):
        t = coslam_amd.KLT_SequenceTracker(cfg, 0)
        t.allocate(W, H, L, *grid)
        t.set_fused(fused)
        t.detect(frames[0])
        t.advanceFrame()
        dev = torch.device("cuda:0")
        bg = [torch.cuda.Stream(device=dev) for _ in range(3)] if load else []
        A = torch.randn(2048, 2048, device=dev) if load else None
        x = torch.zeros(1 << 22, device=dev) if load else None
        y = torch.zeros(4096, device=dev) if load else None
        out = []
        for f in range(1, 9):
            if load:  # keep the background queues full while the tracker frame runs
                with torch.cuda.stream(bg[0]):
                    for _ in range(6):
                        A @ A
                with torch.cuda.stream(bg[1]):
                    for _ in range(8):
                        x.add_(1.0)
                with torch.cuda.stream(bg[2]):
                    for _ in range(200):
                        y.mul_(1.0)
            n, d = t.redetect(frames[f])
            out.append((n, d.copy(), t.read_features().copy()))
            t.advanceFrame()
        if load:
            torch.cuda.synchronize()
        t.close()
        return out

    quiet = run(0, False)
    loaded = run(1, True)
    for (n0, d0, f0), (n1, d1, f1) in zip(quiet, loaded):
        assert n0 == n1
        assert np.array_equal(d0["status"], d1["status"])
        live = d0["status"] >= 0
        assert np.array_equal(d0["pos"][live], d1["pos"][live]) and np.array_equal(d0["gain"][live], d1["gain"][live])
        assert np.array_equal(f0, f1)


@pytest.mark.parametrize("W,H", [(333, 241), (101, 67), (640, 480)])
def test_cornerness_of_fused_front_end_is_bitexact_on_ragged_sizes(hip, W, H):
    """The detector's cornerness map comes out of the level-0 pyramid kernel's LDS tile (halo recomputed, CLAMP_TO_EDGE
    folded in): it must equal the oracle's two-pass result bit for bit, image borders and ragged tiles included."""
    rng = np.random.default_rng(W + H)
    img = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    cfg = cfg2(nLevels=3, minCornerness=50.0, minDistance=3)
    trk, ora = make_pair(cfg, W, H, 3, 16, 12)
    n_g, d_g = trk.detect(img)
    n_o, d_o = ora.detect(img)
    assert np.array_equal(trk.read_cornerness(), ora.read_cornerness())
    assert n_g == n_o
    assert np.array_equal(d_g["status"], d_o["status"])
    live = d_o["status"] >= 0
    assert np.array_equal(d_g["pos"][live], d_o["pos"][live])
    trk.close()


def test_cu_masked_stream_and_budget(hip):
    """cs_stream_create_cu_range + cs_klt_set_cu_count: the tracker on a CU-masked stream gives the same result, and a
    budget too small for the grid falls back to the launch-per-pass schedule instead of hanging."""
    import ctypes as C

    import torch

    W, H, L, grid = 640, 480, 4, (50, 40)
    sc = Scene(1, W, H, 5000, seed=88)
    frames = [sc.render(0, f) for f in range(4)]
    lib = coslam_amd.lib()
    lib.cs_stream_create_cu_range.restype = C.c_void_p
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count

    def run(stream_ptr, cu_count):
        t = coslam_amd.KLT_SequenceTracker(cfg2(), 0)
        t.allocate(W, H, L, *grid)
        if stream_ptr:
            t.set_stream(stream_ptr)
        if cu_count:
            t.set_cu_count(cu_count)
        t.detect(frames[0])
        t.advanceFrame()
        out = []
        for f in range(1, 4):
            n, d = t.redetect(frames[f])
            out.append((n, d.copy()))
            t.advanceFrame()
        t.close()
        return out

    ref = run(None, 0)
    h = lib.cs_stream_create_cu_range(0, 0, n_cus - 64)
    assert h, lib.cs_last_error()
    masked = run(h, n_cus - 64)
    tiny = run(None, 4)  # 4 CUs cannot hold 250 resident waves: must take the per-pass schedule, same numbers
    for other in (masked, tiny):
        for (n0, d0), (n1, d1) in zip(ref, other):
            assert n0 == n1 and np.array_equal(d0["status"], d1["status"])
            live = d0["status"] >= 0
            assert np.array_equal(d0["pos"][live], d1["pos"][live])
    lib.cs_stream_destroy(C.c_void_p(h))


@pytest.mark.parametrize("n_cams,prefetch", [(8, True), (3, False), (13, True), (4, False), (2, True), (1, False)])
def test_camera_group_is_bit_identical_to_single_handles(hip, n_cams, prefetch):
    """cs_klt_group_*: the frame schedule of several cameras in ONE set of launches (camera = one more grid dimension, the
    gain tracker of all cameras in one persistent launch) must give exactly what driving each handle on its own gives:
    detect, redetect and track-only frames, with and without the frame-front prefetch, dest[] / counts / feature lists.
    13 cameras (SLAM_MAX_NUM) x 2000 slots exceed the resident waves of one persistent launch -> two launches in a row.
    cs_klt_set_xcd_placement (a camera's workgroups numbered onto its own XCDs: 8 cameras one XCD each, 4 cameras two each, 13 = 8
    placed + 5 as grid rows, 3 not placed at all; 2 and 1 -- a rank's share of 8 cameras on 4 and 8 GPUs -- four and all eight XCDs each) changes where workgroups run and nothing else: the same bits."""
    import torch

    W, H, L, fw, fh = 640, 480, 4, 50, 40
    dev = torch.device("cuda:0")
    cfg = cfg2()
    sc = Scene(min(n_cams, 8), W, H, 5000, seed=91)
    nf = 5
    frames = [[torch.from_numpy(sc.render(c % 8, f + (c // 8))).to(dev) for f in range(nf)] for c in range(n_cams)]
    plan = ["detect", "redetect", "redetect", "track", "redetect", "redetect", "redetect"]

    def run(grouped, placed=False):
        ts = [coslam_amd.KLT_SequenceTracker(cfg, 0) for _ in range(n_cams)]
        for t in ts:
            t.allocate(W, H, L, fw, fh)
            t.set_stream(torch.cuda.current_stream().cuda_stream)
            t.set_xcd_placement(placed)
        grp = coslam_amd.KLT_TrackerGroup(ts) if grouped else None
        if grp:
            grp.set_stream(torch.cuda.current_stream().cuda_stream)
        dests = [[torch.zeros(fw * fh * 5, dtype=torch.int32, device=dev) for _ in range(n_cams)] for _ in plan]
        counts = [[torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(n_cams)] for _ in plan]
        feats = []
        for s_, what in enumerate(plan):
            imgs = [frames[c][s_ % nf] for c in range(n_cams)]
            nxt = [frames[c][(s_ + 1) % nf] for c in range(n_cams)]
            if grouped:
                if prefetch and s_ + 1 < len(plan):
                    grp.prefetch_dev([x.data_ptr() for x in nxt])
                getattr(grp, what + "_dev")([x.data_ptr() for x in imgs], [d.data_ptr() for d in dests[s_]],
                                            [c_.data_ptr() for c_ in counts[s_]])
                grp.advanceFrame()
            else:
                for c, t in enumerate(ts):
                    getattr(t, what + "_dev")(imgs[c].data_ptr(), dests[s_][c].data_ptr(), counts[s_][c].data_ptr())
                    t.advanceFrame()
            feats.append([t.read_features().copy() for t in ts])
        if grp:
            grp.synchronize()
        torch.cuda.synchronize()
        out = [[(d.cpu().numpy().view(coslam_amd.KLT_TrackedFeature).copy(), c_.cpu().numpy().copy())
                for d, c_ in zip(ds, cs)] for ds, cs in zip(dests, counts)]
        if grp:
            grp.close()
        for t in ts:
            t.close()
        return out, feats

    (single, f_single), (group, f_group), (placed, f_placed) = run(False), run(True), run(True, placed=True)
    for other, f_other in ((group, f_group), (placed, f_placed)):
        for s_ in range(len(plan)):
            for c in range(n_cams):
                (da, ca), (db, cb) = single[s_][c], other[s_][c]
                assert np.array_equal(ca, cb), (s_, c, ca, cb)
                assert np.array_equal(da["status"], db["status"]), (s_, c)
                live = da["status"] >= 0
                assert np.array_equal(da["pos"][live], db["pos"][live]), (s_, c)
                assert np.array_equal(da["gain"][live], db["gain"][live]), (s_, c)
                assert np.array_equal(f_single[s_][c], f_other[s_][c]), (s_, c)
    assert (single[-1][0][0]["status"] == 0).sum() > 300


def test_camera_group_rejects_mismatched_handles(hip):
    a = coslam_amd.KLT_SequenceTracker(cfg2(), 0)
    a.allocate(640, 480, 4, 50, 40)
    b = coslam_amd.KLT_SequenceTracker(cfg2(nIterations=5), 0)
    b.allocate(640, 480, 4, 50, 40)
    c = coslam_amd.KLT_SequenceTracker(cfg2(), 0)
    c.allocate(320, 240, 4, 50, 40)
    for bad in ([a, b], [a, c], [a, a]):
        with pytest.raises(coslam_amd.CoslamHipError):
            coslam_amd.KLT_TrackerGroup(bad)
    g = coslam_amd.KLT_TrackerGroup([a])
    g.close()
    for t in (a, b, c):
        t.close()


def test_staged_host_images_and_async_host_form_match_the_device_path(hip):
    """cs_klt_group_stage_h / cs_klt_group_staged (host images into a device ring on a copy stream, two frames ahead) and
    cs_klt_redetect_async_h / cs_klt_fetch (GPUKLT::next in two halves) give bit for bit what the device-resident and the
    synchronous host entry points give."""
    import torch

    W, H, L, fw, fh, n_cams, n_frames = 320, 240, 3, 20, 15, 3, 7
    sc = Scene(n_cams, W, H, 900, seed=77)
    frames = [[sc.render(c, f) for f in range(n_frames)] for c in range(n_cams)]
    cfg = cfg2(nLevels=L, minCornerness=1500.0)
    dev = torch.device("cuda:0")

    def make():
        ts = []
        for _ in range(n_cams):
            t = coslam_amd.KLT_SequenceTracker(cfg, 0)
            t.allocate(W, H, L, fw, fh)
            ts.append(t)
        return ts, coslam_amd.KLT_TrackerGroup(ts)

    def run(staged):
        ts, grp = make()
        s = torch.cuda.Stream(device=dev)
        grp.set_stream(s.cuda_stream)
        d_img = [torch.from_numpy(np.stack(frames[c])).to(dev) for c in range(n_cams)]
        h_img = [torch.from_numpy(np.stack(frames[c])).pin_memory() for c in range(n_cams)]
        dests = [torch.zeros(fw * fh * 5, dtype=torch.int32, device=dev) for _ in range(n_cams)]
        cnts = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(n_cams)]
        dp, cp = [d.data_ptr() for d in dests], [c.data_ptr() for c in cnts]
        slots, out = {}, []
        if staged:
            for f in (0, 1):
                slots[f] = grp.stage_h([h_img[c][f].data_ptr() for c in range(n_cams)])
        for f in range(n_frames):
            if staged:
                if f + 2 < n_frames:
                    slots[f + 2] = grp.stage_h([h_img[c][f + 2].data_ptr() for c in range(n_cams)])
                cur = grp.staged(slots[f])
                nxt = grp.staged(slots[f + 1]) if f + 1 < n_frames else None
            else:
                cur = [d_img[c][f].data_ptr() for c in range(n_cams)]
                nxt = [d_img[c][f + 1].data_ptr() for c in range(n_cams)] if f + 1 < n_frames else None
            if nxt is not None:
                grp.prefetch_dev(nxt)
            (grp.detect_dev if f == 0 else grp.redetect_dev)(cur, dp, cp)
            grp.advanceFrame()
            grp.synchronize()
            out.append([d.cpu().numpy().copy() for d in dests])
        grp.close()
        for t in ts:
            t.close()
        return out

    a, b = run(False), run(True)
    for f in range(n_frames):
        for c in range(n_cams):
            assert np.array_equal(a[f][c], b[f][c]), (f, c)
    # the asynchronous host form: all cameras' frames in flight together == the synchronous calls
    ts_sync, _g1 = make()
    ts_async, _g2 = make()
    for c in range(n_cams):
        ts_sync[c].detect(frames[c][0])
        ts_async[c].detect(frames[c][0])
        ts_sync[c].advanceFrame()
        ts_async[c].advanceFrame()
    for f in range(1, n_frames):
        for c in range(n_cams):
            ts_async[c].redetect_async(frames[c][f])
        for c in range(n_cams):
            n_s, d_s = ts_sync[c].redetect(frames[c][f])
            n_a, d_a = ts_async[c].fetch()
            assert n_s == n_a and np.array_equal(d_s.view(np.uint8), d_a.view(np.uint8)), (f, c)   # (bitwise: dead slots hold NaN)
            ts_sync[c].advanceFrame()
            ts_async[c].advanceFrame()
    with pytest.raises(coslam_amd.CoslamHipError):
        ts_async[0].fetch()           # nothing outstanding
    _g1.close()
    _g2.close()
    for t in ts_sync + ts_async:
        t.close()
