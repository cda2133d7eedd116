"""ctypes bindings of oracle/_ref/libcgklt_ref.so: the reference's KLT fragment programs (src/tracking/CGKLT/Shaders/*.cg)
compiled in place by oracle/build_cgref.sh and run by oracle/ref_shim/cg/cgklt_driver.cpp.

TEST INFRASTRUCTURE ONLY (tests/ and tests/golden/make_golden.py).  The library exists only where /root/reference was present at
build time; have() says whether it does.  Same buffer layouts as the oracle module (pyramid: 4 binary16 per texel at okl_pyr_layout's
offsets; features: N x 3 binary32 in slot order), so that every function here has a twin there with the same arguments:

    cgref.pyramid_build      <->  oracle.pyramid_build            (okl_pyramid_build)
    cgref.track_nogain       <->  cgref.okl_track_nogain          (okl_track_nogain, serial sums)
    cgref.track_gain_pass    <->  cgref.okl_track_gain_pass       (okl_track_gain_pass)
    cgref.track_gain         <->  the pass loop of okl_seq_track  (run_tracker in klt_oracle.c)
    cgref.cornerness / suppress_present / nonmax / extract  <->  oracle.cornerness / suppress_present / nonmax / extract
"""
import ctypes as C
import os

import numpy as np

from . import _p, lib as _olib, pyr_layout

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libcgklt_ref.so")
_lib = None


def have():
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
    return _lib


def _f(x):
    return C.c_float(float(x))


def _u8(img):
    return np.ascontiguousarray(img, dtype=np.uint8)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def pyramid_build(img, W, H, L, centered=0):
    total, _ = pyr_layout(W, H, L)
    out = np.zeros(total * 4, dtype=np.uint16)
    img = _u8(img)
    assert img.size == W * H
    lib().cgref_pyramid_build(_p(img), W, H, L, int(centered), _p(out))
    return out


def track_nogain(pyr0, pyr1, W, H, L, levelSkip, hw, fw, fh, margin, convThr, ssdThr, featIn):
    pyr0, pyr1, featIn = _u16(pyr0), _u16(pyr1), _f32(featIn).reshape(fw * fh, 3)
    out = np.zeros((fw * fh, 3), dtype=np.float32)
    rc = lib().cgref_track_nogain(_p(pyr0), _p(pyr1), W, H, L, levelSkip, hw, fw, fh, _f(margin), _f(convThr), _f(ssdThr), _p(featIn),
                                  _p(out))
    if rc != 0:
        raise KeyError("klt_tracker.cg was not compiled for N_LEVELS=%d LEVEL_SKIP=%d HALF_WIDTH=%d" % (L, levelSkip, hw))
    return out


def okl_track_nogain(pyr0, pyr1, W, H, L, levelSkip, hw, fw, fh, margin, convThr, ssdThr, featIn, nIterShader=5, tree=False):
    pyr0, pyr1, featIn = _u16(pyr0), _u16(pyr1), _f32(featIn).reshape(fw * fh, 3)
    out = np.zeros((fw * fh, 3), dtype=np.float32)
    Lb = _olib()
    Lb.okl_set_nogain_sum_mode(1 if tree else 0)
    Lb.okl_track_nogain(_p(pyr0), _p(pyr1), W, H, L, levelSkip, hw, nIterShader, _f(margin), _f(convThr), _f(ssdThr), fw * fh,
                        _p(featIn), _p(out))
    Lb.okl_set_nogain_sum_mode(0)
    return out


def _gain_pass(fn, pyr0, pyr1, W, H, L, level, fw, fh, hw, feat0, featIn, sqrConvThr, ssdThr, validRegion, lam, delta):
    pyr0, pyr1 = _u16(pyr0), _u16(pyr1)
    feat0, featIn = _f32(feat0).reshape(fw * fh, 3), _f32(featIn).reshape(fw * fh, 3)
    vr = _f32(validRegion)
    out = np.zeros((fw * fh, 3), dtype=np.float32)
    rc = fn(_p(pyr0), _p(pyr1), W, H, L, level, fw, fh, hw, _p(feat0), _p(featIn), _p(out), _f(sqrConvThr), _f(ssdThr), _p(vr), _f(lam),
            _f(delta))
    return rc, out


def track_gain_pass(pyr0, pyr1, W, H, L, level, fw, fh, hw, feat0, featIn, sqrConvThr, ssdThr, validRegion, lam=1.0, delta=200.0):
    fn = lib().cgref_track_gain_pass
    fn.restype = C.c_int
    rc, out = _gain_pass(fn, pyr0, pyr1, W, H, L, level, fw, fh, hw, feat0, featIn, sqrConvThr, ssdThr, validRegion, lam, delta)
    if rc != 0:
        raise KeyError("klt_tracker_with_gain.cg was not compiled for HALF_WIDTH=%d" % hw)
    return out


def okl_track_gain_pass(pyr0, pyr1, W, H, L, level, fw, fh, hw, feat0, featIn, sqrConvThr, ssdThr, validRegion, lam=1.0, delta=200.0,
                        tree=False):
    Lb = _olib()
    fn = Lb.okl_track_gain_pass_tree if tree else Lb.okl_track_gain_pass
    fn.restype = None
    return _gain_pass(fn, pyr0, pyr1, W, H, L, level, fw, fh, hw, feat0, featIn, sqrConvThr, ssdThr, validRegion, lam, delta)[1]


def track_gain(pyr0, pyr1, W, H, L, levelSkip, hw, nIterations, fw, fh, margin, convThr, ssdThr, feat0, featCur):
    pyr0, pyr1 = _u16(pyr0), _u16(pyr1)
    feat0, featCur = _f32(feat0).reshape(fw * fh, 3), _f32(featCur).reshape(fw * fh, 3)
    out = np.zeros((fw * fh, 3), dtype=np.float32)
    rc = lib().cgref_track_gain(_p(pyr0), _p(pyr1), W, H, L, levelSkip, hw, nIterations, fw, fh, _f(margin), _f(convThr), _f(ssdThr),
                                _p(feat0), _p(featCur), _p(out))
    if rc != 0:
        raise KeyError("klt_tracker_with_gain.cg was not compiled for HALF_WIDTH=%d" % hw)
    return out


def cornerness(lvl0, W, H, minCornerness, margin):
    out = np.zeros((H, W), dtype=np.float32)
    lvl0 = _u16(lvl0)
    lib().cgref_cornerness(_p(lvl0), W, H, _f(minCornerness), _f(margin), _p(out))
    return out


def suppress_present(corner, present3):
    c = _f32(corner).copy()
    H, W = c.shape
    p = _f32(present3).reshape(-1, 3)
    lib().cgref_suppress_present(_p(c), W, H, p.shape[0], _p(p))
    return c


def nonmax(corner, d):
    c = _f32(corner).copy()
    H, W = c.shape
    if lib().cgref_nonmax(_p(c), W, H, int(d)) != 0:
        raise KeyError("klt_detector_nonmax.cg was not compiled for MIN_DIST=%d" % d)
    return c


def extract(corner, plw, maxOut):
    """(count, list) -- list holds min(count, maxOut) rows (s, t, cornerness) in the traversal shader's order."""
    c = _f32(corner)
    H, W = c.shape
    out = np.full((maxOut, 3), -1.0, dtype=np.float32)
    n = lib().cgref_extract(_p(c), W, H, int(plw), int(maxOut), _p(out))
    if n < 0:
        raise KeyError("klt_detector_traverse_histpyr.cg was not compiled for a %d x %d image" % (W, H))
    return n, out[: min(n, maxOut)]
