"""The KLT oracle (oracle/klt_oracle.c, serial summation mode) against the REFERENCE's own fragment programs.

tests/golden/cgklt_golden.npz holds what src/tracking/CGKLT/Shaders/*.cg compute -- compiled where they lie under
/root/reference into oracle/_ref/libcgklt_ref.so (oracle/build_cgref.sh) and run by the rasteriser in
oracle/ref_shim/cg/cgklt_driver.cpp -- on two small scenes: pyramids, cornerness, non-max maps, HistoPyramid point lists and
detect / redetect / redetect sequences with and without gain.  The oracle must reproduce every array BIT FOR BIT: that is what
"KLT arithmetic pinned to the shaders"  means.  What stays ours -- the GL texture model the shaders
sample through and the host's pass schedule -- is stated in oracle/ref_shim/cg/cg_shim.h.

Where the compiled shaders are present (the build container; they also travel to the GPU box), the same comparison runs live
on seeded inputs at the bench's sizes, and the committed fixture is regenerated and compared."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from coslam_amd.klt import KLT_SequenceTrackerConfig
from coslam_amd.synth import Scene
from oracle import cgref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cgklt_golden.npz")
needs_cgref = pytest.mark.skipif(not cgref.have(), reason="oracle/_ref/libcgklt_ref.so absent (built only where /root/reference exists)")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def scene_cfg(g, si, gain):
    W, H, L, fw, fh, win, iters, skip, mind = [int(v) for v in g["scenes"][si]]
    cfg = KLT_SequenceTrackerConfig(nIterations=iters, nLevels=L, levelSkip=skip, windowWidth=win, trackWithGain=gain,
                                    minCornerness=float(g["minCornerness"][si]), convergenceThreshold=1.0, SSD_Threshold=20000.0,
                                    minDistance=mind)
    return W, H, L, fw, fh, cfg


@pytest.mark.parametrize("si", [0, 1])
def test_pyramid_is_the_shaders_pyramid(g, si):
    W, H, L = [int(v) for v in g["scenes"][si][:3]]
    imgs = g[f"s{si}_images"]
    for f in range(3):
        p = oracle.pyramid_build(imgs[f], W, H, L, 0)
        if f"s{si}_pyr{f}_cg" in g:
            assert np.array_equal(p, g[f"s{si}_pyr{f}_cg"]), (si, f)
        else:
            assert np.array_equal(sha(p), g[f"s{si}_pyr{f}_sha_cg"]), (si, f)
    c = oracle.pyramid_build(imgs[0], W, H, L, 1)
    if si == 0:
        assert np.array_equal(c, g["s0_pyr0_centered_cg"])
    else:
        assert np.array_equal(sha(c), g[f"s{si}_pyr0_centered_sha_cg"])


@pytest.mark.parametrize("si", [0, 1])
def test_cornerness_is_the_shaders_cornerness(g, si):
    W, H, L = [int(v) for v in g["scenes"][si][:3]]
    lvl0 = oracle.level_view(g[f"s{si}_pyr0_cg"], W, H, L, 0)
    c = oracle.cornerness(lvl0, W, H, float(g["minCornerness"][si]), 10.0)
    assert np.array_equal(bits(c), bits(g[f"s{si}_cornerness0_cg"]))
    assert (c > 0).sum() > 100


@pytest.mark.parametrize("si,gain", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_sequence_reproduces_the_shaders_bit_for_bit(g, si, gain):
    """detect, redetect, redetect in the shader's summation order: tracked set, positions, gains, the non-max map and the point list
    of every frame equal what the shaders produced."""
    W, H, L, fw, fh, cfg = scene_cfg(g, si, gain)
    pre = f"s{si}_g{gain}_"
    o = oracle.SequenceTracker(cfg, sum_mode=0)
    o.allocate(W, H, L, fw, fh)
    imgs = g[f"s{si}_images"]
    tracked_total = 0
    for f in range(3):
        if f:
            o.advanceFrame()
            n, d = o.redetect(imgs[f])
            trk = g[pre + f"tracked{f}_cg"]
            alive = trk[:, 0] >= 0
            assert np.array_equal(alive, d["status"] == 0), (f, "tracked set")
            assert np.array_equal(bits(trk[alive, :2]), bits(d["pos"][alive])), (f, "tracked positions")
            assert np.array_equal(bits(trk[alive, 2]), bits(d["gain"][alive])), (f, "gain / X0.x channel")
            tracked_total += int(alive.sum())
        else:
            n, d = o.detect(imgs[0])
        assert n == int(g[pre + f"n{f}"][0])
        assert np.array_equal(bits(o.read_cornerness()), bits(g[pre + f"nonmax{f}_cg"])), (f, "non-max map")
        cnt, lst = oracle.extract(o.read_cornerness(), 4 * fw * fh)
        assert np.array_equal(bits(lst), bits(g[pre + f"list{f}_cg"])), (f, "point list")
        assert np.array_equal(d["status"], g[pre + f"status{f}"])
        assert np.array_equal(bits(o.read_features()), bits(g[pre + f"provided{f}"]))
    assert tracked_total > 0.5 * fw * fh
    o.close()


def test_gain_pass_with_dead_slots_and_thresholds(g):
    W, H, L, fw, fh, win = [int(v) for v in g["scenes"][0][:6]]
    for level in range(L):
        r = cgref.okl_track_gain_pass(g["s0_pyr0_cg"], g["s0_pyr1_cg"], W, H, L, level, fw, fh, win // 2, g["pass_feat0"], g["pass_cur"],
                                      4.0, 20000.0, g["pass_vr"], 1.0, 200.0)
        assert np.array_equal(bits(r), bits(g[f"pass_level{level}_cg"])), level
    alive = [(g[f"pass_level{level}_cg"][:, 0] >= 0).sum() for level in range(L)]
    assert 0 < min(alive) and max(alive) < fw * fh   # some slots invalidated, some kept, at every level


# ------------------------------------------------------------------------------------------------- live, where the shaders are built

@needs_cgref
def test_committed_fixture_is_what_the_shaders_produce_now():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD), "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    fresh, old = mg.cgklt_cases(), np.load(GOLD)
    assert sorted(fresh.keys()) == sorted(old.files)
    for k in old.files:
        assert np.ascontiguousarray(fresh[k]).tobytes() == np.ascontiguousarray(old[k]).tobytes(), k


LIVE = [  # name, W, H, L, skip, window, iterations, fw, fh, minDistance
    ("cfg2", 640, 480, 4, 1, 7, 10, 50, 40, 5),
    ("coslam-default", 640, 480, 6, 2, 6, 12, 32, 32, 8),
    ("cfg5-quarter", 960, 540, 4, 1, 7, 10, 50, 25, 5),     # 2:1 slot grid as cfg5's 100 x 50
    ("small-window", 320, 240, 3, 2, 5, 5, 20, 15, 3),
]


@needs_cgref
@pytest.mark.parametrize("case", LIVE, ids=[c[0] for c in LIVE])
def test_live_every_pass_equals_the_shaders(case):
    name, W, H, L, skip, win, iters, fw, fh, mind = case
    sc = Scene(1, W, H, W * H // 80, seed=len(name) + W)
    im0, im1 = sc.render(0, 0), sc.render(0, 1)
    p0, p1 = oracle.pyramid_build(im0, W, H, L, 0), oracle.pyramid_build(im1, W, H, L, 0)
    assert np.array_equal(p0, cgref.pyramid_build(im0, W, H, L, 0)) and np.array_equal(p1, cgref.pyramid_build(im1, W, H, L, 0))
    if all((W >> l) % 2 == 0 and (H >> l) % 2 == 0 for l in range(L - 1)):
        # the other resolution of taps that sit exactly on a texel edge; with an odd source size no tap sits on an edge and the
        # oracle's `centered` (tap base - 1, our own switch) is not a reading of the shader at all -- not compared there
        assert np.array_equal(oracle.pyramid_build(im0, W, H, L, 1), cgref.pyramid_build(im0, W, H, L, 1))
    lvl0 = oracle.level_view(p0, W, H, L, 0)
    c_o, c_s = oracle.cornerness(lvl0, W, H, 3000.0, 10.0), cgref.cornerness(lvl0, W, H, 3000.0, 10.0)
    assert np.array_equal(bits(c_o), bits(c_s))
    rng = np.random.default_rng(W + fw)
    present = np.concatenate([rng.random((200, 2)), np.zeros((200, 1))], 1).astype(np.float32)
    present[::7] = -1.0
    s_o, s_s = oracle.suppress_present(c_o, present), cgref.suppress_present(c_o, present)
    assert np.array_equal(bits(s_o), bits(s_s))
    n_o, n_s = oracle.nonmax(s_o, mind), cgref.nonmax(s_o, mind)
    assert np.array_equal(bits(n_o), bits(n_s)) and (n_o > 0).sum() > 50
    k_o, l_o = oracle.extract(n_o, 4 * fw * fh)
    k_s, l_s = cgref.extract(n_o, 2 * fw, 4 * fw * fh)
    assert k_o == k_s and np.array_equal(bits(l_o), bits(l_s))
    # trackers: the detected list in the slots, a tenth of them dead
    N, hw = fw * fh, win // 2
    feat = np.full((N, 3), -1.0, np.float32)
    m = min(N, len(l_o))
    feat[:m, :2] = l_o[:m, :2]
    feat[:, 2] = 1.0
    feat[rng.random(N) < 0.1] = (-1.0, -1.0, 1.0)
    t_s = cgref.track_nogain(p0, p1, W, H, L, skip, hw, fw, fh, 4.0, 1.0, 20000.0, feat)
    t_o = cgref.okl_track_nogain(p0, p1, W, H, L, skip, hw, fw, fh, 4.0, 1.0, 20000.0, feat)
    assert np.array_equal(bits(t_o), bits(t_s)) and (t_s[:, 0] >= 0).sum() > 0.3 * m
    # gain tracker: the host's whole level / iteration loop by the shaders vs the oracle's passes in the same schedule
    g_s = cgref.track_gain(p0, p1, W, H, L, skip, hw, iters, fw, fh, 4.0, 1.0, 20000.0, feat, feat)
    cur, other = feat.copy(), None
    cur[:, 2] = 1.0
    vr_off, vr_on = [-1.0, -1.0, 2.0, 2.0], [4.0 / W, 4.0 / H, 1.0 - 4.0 / W, 1.0 - 4.0 / H]
    thr = (1e6, 1e6, vr_off)
    for level in range(L - 1, -1, -skip):
        for it in range(1, iters + 1):
            if it == 1:
                thr = (1e6, 1e6, vr_off)
            elif it == iters:
                thr = (1.0, 20000.0, vr_on)
            cur = cgref.okl_track_gain_pass(p0, p1, W, H, L, level, fw, fh, hw, feat, cur, thr[0], thr[1], thr[2], 1.0, 200.0)
    assert np.array_equal(bits(cur), bits(g_s)) and (g_s[:, 0] >= 0).sum() > 0.3 * m
