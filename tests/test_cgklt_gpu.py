"""The HIP KLT kernels against the REFERENCE's own fragment programs (tests/golden/cgklt_golden.npz: outputs of
src/tracking/CGKLT/Shaders/*.cg compiled in place, see tests/test_cgklt_cpu.py and oracle/ref_shim/cg/).

Bit for bit: pyramid texels (every level, both resolutions of the edge taps), the cornerness / non-max map, the detection list
and the slot table after detect().  The trackers sum their windows in a fixed tree instead of the shader's serial loop
(klt_track_rows.hip, klt_track.hip), so tracked positions are held to 0.02 px per frame of divergence and a slot
tracked by one side and invalidated by the other must sit within 1 % of a threshold the shader tests -- the margin the fixture
recorded for that slot while the shaders' result was produced.

Where oracle/_ref/libcgklt_ref.so travelled to the box, the shaders also run LIVE against the kernels at the bench's size."""
import hashlib
import os

import numpy as np
import pytest

import coslam_amd
import oracle
from coslam_amd.synth import Scene
from oracle import cgref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cgklt_golden.npz")
TOL_PX = 0.02


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def scene_cfg(g, si, gain):
    W, H, L, fw, fh, win, iters, skip, mind = [int(v) for v in g["scenes"][si]]
    cfg = coslam_amd.KLT_SequenceTrackerConfig(nIterations=iters, nLevels=L, levelSkip=skip, windowWidth=win, trackWithGain=gain,
                                               minCornerness=float(g["minCornerness"][si]), convergenceThreshold=1.0,
                                               SSD_Threshold=20000.0, minDistance=mind)
    return W, H, L, fw, fh, cfg


@pytest.mark.parametrize("si", [0, 1])
def test_pyramid_texels_equal_the_shaders(hip, g, si):
    W, H, L, fw, fh, cfg = scene_cfg(g, si, 1)
    for tap in (0, 1):
        t = coslam_amd.KLT_SequenceTracker(cfg, 0, tap)
        t.allocate(W, H, L, fw, fh)
        for f in range(3 if tap == 0 else 1):
            t.build_pyramid(g[f"s{si}_images"][f])
            p = t.read_pyramid(1)
            name = f"s{si}_pyr{f}" + ("_centered" if tap else "")
            if name + "_cg" in g:
                assert np.array_equal(p, g[name + "_cg"]), name
            else:
                assert np.array_equal(sha(p), g[name + "_sha_cg"]), name
        t.close()


def check_tracked(d, trk, margin, W, H, frames_apart, what):
    """d: the kernel's dest; trk: the shaders' raw tracker output (x < 0: invalidated); margin: per slot, the smallest relative
    distance to a threshold the shaders' run recorded"""
    a_s, a_g = trk[:, 0] >= 0, d["status"] == 0
    both, differ = a_s & a_g, a_s ^ a_g
    err = np.abs(d["pos"][both] - trk[both, :2]) * np.array([W, H], np.float32)
    assert both.sum() > 0 and err.max() <= TOL_PX * frames_apart, (what, float(err.max()))
    if frames_apart == 1:
        assert np.all(margin[differ] < 0.01), (what, "status differs away from every threshold", margin[differ])
    assert differ.sum() <= 0.02 * max(a_s.sum(), 1) + 1, (what, int(differ.sum()), int(a_s.sum()))
    return both, err.max()


@pytest.mark.parametrize("si,gain", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_sequence_against_the_shaders(hip, g, si, gain):
    W, H, L, fw, fh, cfg = scene_cfg(g, si, gain)
    pre = f"s{si}_g{gain}_"
    t = coslam_amd.KLT_SequenceTracker(cfg, 0)
    t.allocate(W, H, L, fw, fh)
    imgs = g[f"s{si}_images"]
    # frame 0: the detector, bit for bit
    n, d = t.detect(imgs[0])
    assert n == int(g[pre + "n0"][0])
    assert np.array_equal(bits(t.read_cornerness()), bits(g[pre + "nonmax0_cg"])), "non-max map"
    assert np.array_equal(d["status"], g[pre + "status0"])
    live = d["status"] >= 0
    assert np.array_equal(bits(d["pos"][live]), bits(g[pre + "pos0"][live]))
    lst = g[pre + "list0_cg"]
    assert set(map(tuple, d["pos"][live].tolist())) <= set(map(tuple, lst[:, :2].tolist())), "a detected corner the shaders did not list"
    assert np.array_equal(bits(t.read_features()), bits(g[pre + "provided0"])), "slot table handed to the tracker"
    in_sync = True
    for f in (1, 2):
        t.advanceFrame()
        n, d = t.redetect(imgs[f])
        trk = g[pre + f"tracked{f}_cg"]
        both, emax = check_tracked(d, trk, g[pre + f"margin{f}"], W, H, f, f"scene {si} gain {gain} frame {f}")
        same_set = np.array_equal(d["status"] == 0, trk[:, 0] >= 0)
        same_px = np.array_equal(np.floor(d["pos"][both] * [W, H]), np.floor(trk[both, :2] * [W, H]))
        if in_sync and same_set and same_px:
            # same slots alive and the same pixels suppressed => the detector's input is the shaders' input: exact again
            assert np.array_equal(bits(t.read_cornerness()), bits(g[pre + f"nonmax{f}_cg"])), (f, "non-max map")
            assert np.array_equal(d["status"], g[pre + f"status{f}"]), f
            new = d["status"] == 1
            assert np.array_equal(bits(d["pos"][new]), bits(g[pre + f"pos{f}"][new])), (f, "new corners")
            assert n == int(g[pre + f"n{f}"][0])
        else:
            in_sync = False
    t.close()


needs_cgref = pytest.mark.skipif(not cgref.have(), reason="oracle/_ref/libcgklt_ref.so did not travel to this box")


@needs_cgref
@pytest.mark.parametrize("gain", [1, 0])
def test_live_shaders_against_the_kernels_at_bench_size(hip, gain):
    """cfg2 (640 x 480, 2000 slots as 50 x 40, 4 levels, 7 x 7 window, 10 iterations): detect on frame 0, redetect on frame 1 by the
    kernels; the same two frames through the compiled shaders."""
    W, H, L, fw, fh, win, iters, skip, mind = 640, 480, 4, 50, 40, 7, 10, 1, 5
    cfg = coslam_amd.KLT_SequenceTrackerConfig(nIterations=iters, nLevels=L, levelSkip=skip, windowWidth=win, trackWithGain=gain,
                                               minCornerness=3000.0, convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=mind)
    sc = Scene(1, W, H, 4000, seed=21)
    im0, im1 = sc.render(0, 0), sc.render(0, 1)
    t = coslam_amd.KLT_SequenceTracker(cfg, 0)
    t.allocate(W, H, L, fw, fh)
    n0, d0 = t.detect(im0)
    p0 = cgref.pyramid_build(im0, W, H, L, 0)
    assert np.array_equal(t.read_pyramid(1), p0)
    c = cgref.nonmax(cgref.cornerness(oracle.level_view(p0, W, H, L, 0), W, H, 3000.0, 10.0), mind)
    assert np.array_equal(bits(t.read_cornerness()), bits(c))
    cnt, lst = cgref.extract(c, 2 * fw, 4 * fw * fh)
    live = d0["status"] >= 0
    assert n0 == min(cnt, fw * fh) and set(map(tuple, d0["pos"][live].tolist())) <= set(map(tuple, lst[:, :2].tolist()))
    prov = t.read_features()
    t.advanceFrame()
    n1, d1 = t.redetect(im1)
    p1 = cgref.pyramid_build(im1, W, H, L, 0)
    assert np.array_equal(t.read_pyramid(1), p1)
    # the serial-order oracle is bit-identical to the shaders (tests/test_cgklt_cpu.py); it records the threshold margins
    margin = np.full(fw * fh, 1e30, np.float32)
    if gain:
        trk = cgref.track_gain(p0, p1, W, H, L, skip, win // 2, iters, fw, fh, cfg.trackBorderMargin, 1.0, 20000.0, prov, prov)
    else:
        trk = cgref.track_nogain(p0, p1, W, H, L, skip, win // 2, fw, fh, cfg.trackBorderMargin, 1.0, 20000.0, prov)
    ser = oracle.SequenceTracker(cfg, sum_mode=0)
    ser.allocate(W, H, L, fw, fh)
    ser.detect(im0)
    ser.advanceFrame()
    oracle.set_threshold_margin_buffer(margin)
    try:
        _, d_s = ser.track(im1)
    finally:
        oracle.set_threshold_margin_buffer(None)
    ser.close()
    assert np.array_equal(d_s["status"] == 0, trk[:, 0] >= 0) and np.array_equal(bits(d_s["pos"][trk[:, 0] >= 0]), bits(trk[trk[:, 0] >= 0, :2]))
    both, emax = check_tracked(d1, trk, margin, W, H, 1, f"cfg2 gain {gain}")
    assert both.sum() > 500
    t.close()
