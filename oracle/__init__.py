"""ctypes bindings of oracle/liboracle.so (and oracle/_ref/libintracam_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under coslam_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libintracam_ref.so")

TrackedFeature = np.dtype([("status", "<i4"), ("pos", "<f4", (2,)), ("gain", "<f4"), ("fed", "<i4")])


class Config(C.Structure):
    _fields_ = [
        ("nIterations", C.c_int), ("nLevels", C.c_int), ("levelSkip", C.c_int), ("windowWidth", C.c_int),
        ("trackBorderMargin", C.c_float), ("convergenceThreshold", C.c_float), ("SSD_Threshold", C.c_float),
        ("trackWithGain", C.c_int), ("minDistance", C.c_int), ("minCornerness", C.c_float),
        ("detectBorderMargin", C.c_float),
    ]

    @classmethod
    def from_any(cls, cfg):
        c = cls()
        for n, _ in cls._fields_:
            setattr(c, n, getattr(cfg, n))
        return c


def build(force=False):
    """make -C oracle (liboracle.so, and _ref/ when /root/reference exists)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        L = _lib
        L.okl_pyr_layout.restype = C.c_size_t
        L.okl_pyr_layout.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.okl_seq_create.restype = C.c_void_p
        L.okl_seq_create.argtypes = [C.POINTER(Config), C.c_int]
        L.okl_seq_cur_pyramid.restype = C.c_void_p
        L.okl_seq_cornerness.restype = C.c_void_p
        L.okl_f32_to_f16.restype = C.c_uint16
        L.okl_f32_to_f16.argtypes = [C.c_float]
        L.okl_f16_to_f32.restype = C.c_float
        L.okl_f16_to_f32.argtypes = [C.c_uint16]
        L.okl_extract.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pyr_layout(W, H, L):
    off = np.zeros(L, dtype=np.int64)
    total = lib().okl_pyr_layout(W, H, L, _p(off))
    return int(total), off


def pyramid_build(img, W, H, L, centered=0):
    total, off = pyr_layout(W, H, L)
    out = np.zeros(total * 4, dtype=np.uint16)
    img = np.ascontiguousarray(img, dtype=np.uint8)
    lib().okl_pyramid_build(_p(img), W, H, L, centered, _p(out))
    return out


def level_view(pyr, W, H, L, level):
    _, off = pyr_layout(W, H, L)
    w, h = W >> level, H >> level
    return pyr.reshape(-1, 4)[off[level]: off[level] + w * h].reshape(h, w, 4)


def half_to_float(u16):
    return np.asarray(u16, dtype=np.uint16).view(np.float16).astype(np.float32)


def cornerness(lvl0, W, H, minCornerness, margin):
    out = np.zeros((H, W), dtype=np.float32)
    lvl0 = np.ascontiguousarray(lvl0, dtype=np.uint16)
    lib().okl_cornerness(_p(lvl0), W, H, C.c_float(minCornerness), C.c_float(margin), _p(out))
    return out


def nonmax(corner, d):
    c = np.ascontiguousarray(corner, dtype=np.float32).copy()
    H, W = c.shape
    lib().okl_nonmax(_p(c), W, H, d)
    return c


def suppress_present(corner, present3):
    c = np.ascontiguousarray(corner, dtype=np.float32).copy()
    H, W = c.shape
    p = np.ascontiguousarray(present3, dtype=np.float32).reshape(-1, 3)
    lib().okl_suppress_present(_p(c), W, H, p.shape[0], _p(p))
    return c


def extract(corner, maxOut):
    c = np.ascontiguousarray(corner, dtype=np.float32)
    H, W = c.shape
    out = np.zeros((maxOut, 3), dtype=np.float32)
    n = lib().okl_extract(_p(c), W, H, maxOut, _p(out))
    return n, out[: min(n, maxOut)]


def sample(lvl, Wl, Hl, s, t):
    out = np.zeros(3, dtype=np.float32)
    lvl = np.ascontiguousarray(lvl, dtype=np.uint16)
    lib().okl_sample(_p(lvl), Wl, Hl, C.c_float(s), C.c_float(t), _p(out))
    return out


def handback(features, W, H, K, kud, mapPts, slot2map, trackSpan, xy, frame, isStatic=None, nColBlk=16, nRowBlk=12,
             ptsStride=192):
    """ohb_handback: frame number `frame` of one camera.  slot2map (int32[N]), trackSpan (int32[2N]: first / last frame
    of every slot's track, -1 = empty) and xy (float64[2N]) are updated in place.
    Returns dict(state, selBlk, npts, Ms, ms, sel)."""
    L = lib()
    L.ohb_handback.restype = C.c_int
    N = len(features)
    f = np.ascontiguousarray(features, dtype=TrackedFeature)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    kud = np.ascontiguousarray(kud, dtype=np.float64).reshape(7)
    mp = np.ascontiguousarray(mapPts, dtype=np.float64).reshape(-1, 3)
    assert slot2map.dtype == np.int32 and trackSpan.dtype == np.int32 and trackSpan.size == 2 * N
    assert xy.dtype == np.float64 and xy.size == 2 * N
    st = np.zeros(N, dtype=np.int32)
    selBlk = np.zeros(nColBlk * nRowBlk, dtype=np.int32)
    Ms, ms = np.zeros((ptsStride, 3)), np.zeros((ptsStride, 2))
    sel = np.full(ptsStride, -1, dtype=np.int32)
    stat = None if isStatic is None else np.ascontiguousarray(isStatic, dtype=np.uint8)
    n = L.ohb_handback(N, W, H, int(frame), _p(f), _p(K), _p(kud), _p(mp), _p(stat) if stat is not None else None,
                       _p(slot2map), _p(trackSpan), _p(xy), _p(st), nColBlk, nRowBlk, _p(selBlk), ptsStride, _p(Ms), _p(ms),
                       _p(sel))
    return dict(state=st, selBlk=selBlk, npts=n, Ms=Ms[:n], ms=ms[:n], sel=sel[:n])


def point_features(state, slot2map, P):
    """ohb_point_features: int32[P], the slot of the camera's feature of this frame attached to every map point, or -1."""
    st = np.ascontiguousarray(state, dtype=np.int32)
    s2 = np.ascontiguousarray(slot2map, dtype=np.int32)
    out = np.zeros(P, dtype=np.int32)
    lib().ohb_point_features(len(st), _p(st), _p(s2), int(P), 1, _p(out))
    return out


def search_maha_nearest(xy, state, m, var, maxDist):
    """org_search_maha_nearest (searchMahaNearestFeatPt): xy float64[2N] (x then y), state int32[N].  Returns (slot, dmin)."""
    L = lib()
    L.org_search_maha_nearest.restype = C.c_int
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    st = np.ascontiguousarray(state, dtype=np.int32)
    m = np.ascontiguousarray(m, dtype=np.float64)
    var = np.ascontiguousarray(var, dtype=np.float64).reshape(4)
    d = C.c_double(0)
    s = L.org_search_maha_nearest(len(st), _p(xy), _p(st), _p(m), _p(var), C.c_double(maxDist), C.byref(d))
    return s, d.value


def register_search(W, H, Ks, Rs, ts, xy, state, slot2map, isDynamic, Ms, covs, pointFeat, sigmaSearch, maxDist, sigmaMerge):
    """org_register_search: per-camera lists xy (float64[2N]), state / slot2map (int32[N]), isDynamic (uint8[N] or None);
    Ms (P x 3), covs (P x 9), pointFeat (P x nCams int32).  Returns dict(slot, m, var, dist, flags), P x nCams each."""
    L = lib()
    nC = len(xy)
    N = len(state[0])
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    Rs = np.ascontiguousarray(Rs, dtype=np.float64).reshape(nC, 9)
    ts = np.ascontiguousarray(ts, dtype=np.float64).reshape(nC, 3)
    Ms = np.ascontiguousarray(Ms, dtype=np.float64).reshape(-1, 3)
    P = len(Ms)
    covs = np.ascontiguousarray(covs, dtype=np.float64).reshape(P, 9)
    pf = np.ascontiguousarray(pointFeat, dtype=np.int32).reshape(P, nC)
    xs = [np.ascontiguousarray(a, dtype=np.float64) for a in xy]
    ss = [np.ascontiguousarray(a, dtype=np.int32) for a in state]
    s2 = [np.ascontiguousarray(a, dtype=np.int32) for a in slot2map]
    dy = [None if a is None else np.ascontiguousarray(a, dtype=np.uint8) for a in isDynamic]
    vp = C.c_void_p * nC
    a_xy = vp(*[a.ctypes.data for a in xs])
    a_st = vp(*[a.ctypes.data for a in ss])
    a_s2 = vp(*[a.ctypes.data for a in s2])
    a_dy = vp(*[None if a is None else a.ctypes.data for a in dy])
    slot = np.zeros((P, nC), dtype=np.int32)
    m = np.zeros((P, nC, 2))
    var = np.zeros((P, nC, 4))
    dist = np.zeros((P, nC))
    flags = np.zeros((P, nC), dtype=np.int32)
    L.org_register_search(nC, N, int(W), int(H), _p(Ks), _p(Rs), _p(ts), a_xy, a_st, a_s2, a_dy, P, _p(Ms), _p(covs), _p(pf),
                          C.c_double(sigmaSearch), C.c_double(maxDist), C.c_double(sigmaMerge), _p(slot), _p(m), _p(var),
                          _p(dist), _p(flags))
    return dict(slot=slot, m=m, var=var, dist=dist, flags=flags)


def seq_triangulate(K, R, t, m, M, cov, sigma):
    """opu_seq_triangulate: returns the updated (M, cov)."""
    L = lib()
    K, R, t, m = (np.ascontiguousarray(a, dtype=np.float64).reshape(-1) for a in (K, R, t, m))
    M = np.array(M, dtype=np.float64).reshape(3).copy()
    cov = np.array(cov, dtype=np.float64).reshape(9).copy()
    L.opu_seq_triangulate(_p(K), _p(R), _p(t), _p(m), _p(M), _p(cov), C.c_double(sigma))
    return M, cov.reshape(3, 3)


def pose_update_gate(Ks, Rs, ts, xy, state, slot2map, mapPts, mapCov, mapFlags, largeErr, sigma, reprojErr, cams=None):
    """poseUpdate3D's second half for the cameras `cams` (default: all) ONE AFTER THE OTHER (opu_gate_camera per camera, the map
    updated in place between them).  Per-camera lists xy (float64[2N]), state / slot2map (int32[N]), reprojErr (float64[N], in /
    out); mapPts (P x 3), mapCov (P x 9), mapFlags (uint8[P]) are updated IN PLACE.  Returns (num nodes, numOut) per camera."""
    L = lib()
    L.opu_gate_camera.restype = C.c_int
    nC = len(xy)
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    Rs = np.ascontiguousarray(Rs, dtype=np.float64).reshape(nC, 9)
    ts = np.ascontiguousarray(ts, dtype=np.float64).reshape(nC, 3)
    assert mapPts.dtype == np.float64 and mapCov.dtype == np.float64 and mapFlags.dtype == np.uint8
    assert mapPts.flags.c_contiguous and mapCov.flags.c_contiguous and mapFlags.flags.c_contiguous
    out = []
    for c in (range(nC) if cams is None else cams):
        x = np.ascontiguousarray(xy[c], dtype=np.float64)
        st = np.ascontiguousarray(state[c], dtype=np.int32)
        s2 = np.ascontiguousarray(slot2map[c], dtype=np.int32)
        assert reprojErr[c].dtype == np.float64 and reprojErr[c].flags.c_contiguous
        no = C.c_int(0)
        num = L.opu_gate_camera(_p(Ks[c]), _p(Rs[c]), _p(ts[c]), len(st), _p(x), _p(st), _p(s2), len(mapFlags), _p(mapPts),
                                _p(mapCov), _p(mapFlags), int(largeErr), C.c_double(sigma), _p(reprojErr[c]), C.byref(no))
        out.append((num, no.value))
    return out


def detect_dynamic(iK, histR, histT, histXY, state, slot2map, trackSpan, mapFlags, maxLen, minLen, minOutNum, maxEpiErr, isStatic):
    """opu_detect_dynamic_camera for one camera: histR (nHist x 9), histT (nHist x 3), histXY (nHist x 2N), entry 0 = this frame;
    isStatic uint8[N] is updated in place.  Returns the number of features made dynamic."""
    L = lib()
    L.opu_detect_dynamic_camera.restype = C.c_int
    iK = np.ascontiguousarray(iK, dtype=np.float64).reshape(9)
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nH = len(histR)
    st = np.ascontiguousarray(state, dtype=np.int32)
    s2 = np.ascontiguousarray(slot2map, dtype=np.int32)
    sp = np.ascontiguousarray(trackSpan, dtype=np.int32)
    fl = np.ascontiguousarray(mapFlags, dtype=np.uint8)
    assert isStatic.dtype == np.uint8 and isStatic.flags.c_contiguous and histXY.shape == (nH, 2 * len(st))
    return L.opu_detect_dynamic_camera(_p(iK), len(st), nH, nH, _p(histR), _p(histT), _p(histXY), _p(st), _p(s2), _p(sp), len(fl),
                                       _p(fl), int(maxLen), int(minLen), int(minOutNum), C.c_double(maxEpiErr), _p(isStatic))


def update_new_poses_points(Ks, iKs, histR, histT, histXY, trackSpan, featStatic, pointFeat, mapPts, mapCov, mapFlags, sigma,
                            lastFrame=None, isCurrent=None, firstKeyFrame=-1, cmpAcos=False):
    """opu_update_new_poses_points (RobustBundleRTS::updateNewPosesPoints): histR (nC x nHist x 9), histT (nC x nHist x 3), histXY
    (nC x nHist x 2N), entry 0 = this frame; trackSpan (nC x 2N), featStatic (nC x N), pointFeat (nMap x nC); mapPts (nMap x 3) and
    mapCov (nMap x 9) are updated IN PLACE.  Returns (number re-triangulated, static ones, dynamic ones, chosen second views)."""
    L = lib()
    L.opu_update_new_poses_points.restype = C.c_int
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nC, nH = histR.shape[0], histR.shape[1]
    N = histXY.shape[2] // 2
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    iKs = np.ascontiguousarray(iKs, dtype=np.float64).reshape(nC, 9)
    sp = np.ascontiguousarray(trackSpan, dtype=np.int32).reshape(nC, 2 * N)
    fs = np.ascontiguousarray(featStatic, dtype=np.uint8).reshape(nC, N)
    pf = np.ascontiguousarray(pointFeat, dtype=np.int32)
    nMap = pf.shape[0]
    assert pf.shape == (nMap, nC) and histT.shape == (nC, nH, 3) and histXY.shape == (nC, nH, 2 * N)
    assert mapPts.dtype == np.float64 and mapCov.dtype == np.float64 and mapPts.flags.c_contiguous and mapCov.flags.c_contiguous
    fl = np.ascontiguousarray(mapFlags, dtype=np.uint8)
    lf = None if lastFrame is None else np.ascontiguousarray(lastFrame, dtype=np.int32)
    ic = None if isCurrent is None else np.ascontiguousarray(isCurrent, dtype=np.uint8)
    chosen = np.full((nMap, nC), -1, dtype=np.int32)
    ns, nd = C.c_int(0), C.c_int(0)
    n = L.opu_update_new_poses_points(nC, N, nH, _p(Ks), _p(iKs), _p(histR), _p(histT), _p(histXY), _p(sp), _p(fs), nMap, _p(pf),
                                      _p(lf) if lf is not None else None, _p(ic) if ic is not None else None, int(firstKeyFrame),
                                      _p(mapPts), _p(mapCov), _p(fl), C.c_double(sigma), int(bool(cmpAcos)), _p(chosen),
                                      C.byref(ns), C.byref(nd))
    return n, ns.value, nd.value, chosen


def refine_map_points(Ks, iKs, histR, histT, histXY, trackSpan, pointFeat, mapPts, mapCov, sigma, select=None, cmpAcos=False):
    """opu_refine_map_points (CoSLAM::refineMapPoint for the selected points; layouts as update_new_poses_points); mapPts / mapCov are
    updated IN PLACE.  Returns the number of points refined."""
    L = lib()
    L.opu_refine_map_points.restype = C.c_int
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nC, nH = histR.shape[0], histR.shape[1]
    N = histXY.shape[2] // 2
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    iKs = np.ascontiguousarray(iKs, dtype=np.float64).reshape(nC, 9)
    sp = np.ascontiguousarray(trackSpan, dtype=np.int32).reshape(nC, 2 * N)
    pf = np.ascontiguousarray(pointFeat, dtype=np.int32)
    nMap = pf.shape[0]
    assert pf.shape == (nMap, nC) and mapPts.dtype == np.float64 and mapCov.dtype == np.float64
    assert mapPts.flags.c_contiguous and mapCov.flags.c_contiguous
    sel = None if select is None else np.ascontiguousarray(select, dtype=np.uint8)
    return L.opu_refine_map_points(nC, N, nH, _p(Ks), _p(iKs), _p(histR), _p(histT), _p(histXY), _p(sp), nMap, _p(pf),
                                   _p(sel) if sel is not None else None, _p(mapPts), _p(mapCov), C.c_double(sigma), int(bool(cmpAcos)))


def map_points_classify(Ks, iKs, histR, histT, histXY, trackSpan, featStatic, pointFeat, curFrame, mapPts, mapCov, mapFlags, newPt,
                        staticFrameNum, firstFrame, pixelVar, featFrame=None, featFirst=None, slot2map=None):
    """opu_map_points_classify (CoSLAM::mapPointsClassify, one frame; layouts as update_new_poses_points).  featStatic (nC x N uint8),
    pointFeat (nMap x nC int32), mapPts, mapCov, mapFlags, newPt (uint8), staticFrameNum (int32) and slot2map (nC x N int32, optional)
    are updated IN PLACE.  Returns (points examined, points that became false)."""
    L = lib()
    L.opu_map_points_classify.restype = C.c_int
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nC, nH = histR.shape[0], histR.shape[1]
    N = histXY.shape[2] // 2
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    iKs = np.ascontiguousarray(iKs, dtype=np.float64).reshape(nC, 9)
    sp = np.ascontiguousarray(trackSpan, dtype=np.int32).reshape(nC, 2 * N)
    nMap = pointFeat.shape[0]
    for a, dt in ((featStatic, np.uint8), (pointFeat, np.int32), (mapPts, np.float64), (mapCov, np.float64), (mapFlags, np.uint8),
                  (newPt, np.uint8), (staticFrameNum, np.int32)):
        assert a.dtype == dt and a.flags.c_contiguous
    assert pointFeat.shape == (nMap, nC) and featStatic.shape == (nC, N)
    ff = None if featFrame is None else np.ascontiguousarray(featFrame, dtype=np.int32)
    f1 = None if featFirst is None else np.ascontiguousarray(featFirst, dtype=np.int32)
    fr = np.ascontiguousarray(firstFrame, dtype=np.int32)
    if slot2map is not None:
        assert slot2map.dtype == np.int32 and slot2map.flags.c_contiguous and slot2map.shape == (nC, N)
    nf = C.c_int(0)
    n = L.opu_map_points_classify(nC, N, nH, _p(Ks), _p(iKs), _p(histR), _p(histT), _p(histXY), _p(sp), _p(featStatic),
                                  _p(slot2map) if slot2map is not None else None, nMap, _p(pointFeat),
                                  _p(ff) if ff is not None else None, _p(f1) if f1 is not None else None, int(curFrame), _p(mapPts),
                                  _p(mapCov), _p(mapFlags), _p(newPt), _p(staticFrameNum), _p(fr), C.c_double(pixelVar), C.byref(nf))
    return n, nf.value


def map_points_classify_ref(Ks, iKs, histR, histT, histXY, trackSpan, featStatic, pointFeat, featRef, segPool, refStatic, curFrame, mapPts,
                            mapCov, mapFlags, newPt, staticFrameNum, firstFrame, pixelVar, slot2map=None):
    """opu_map_points_classify_ref: the classification with the points' features as references -- featRef (nMap x nC x 4 int32, in / out:
    a detached view's reference is cleared), segPool (nC x cap x 4 int32), refStatic (nMap x nC uint8, in / out) or None.  Everything
    else as map_points_classify.  Returns (points examined, points that became false)."""
    L = lib()
    L.opu_map_points_classify_ref.restype = C.c_int
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nC, nH = histR.shape[0], histR.shape[1]
    N = histXY.shape[2] // 2
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    iKs = np.ascontiguousarray(iKs, dtype=np.float64).reshape(nC, 9)
    sp = np.ascontiguousarray(trackSpan, dtype=np.int32).reshape(nC, 2 * N)
    nMap = pointFeat.shape[0]
    for a, dt in ((featStatic, np.uint8), (pointFeat, np.int32), (featRef, np.int32), (mapPts, np.float64), (mapCov, np.float64),
                  (mapFlags, np.uint8), (newPt, np.uint8), (staticFrameNum, np.int32)):
        assert a.dtype == dt and a.flags.c_contiguous
    assert pointFeat.shape == (nMap, nC) and featStatic.shape == (nC, N) and featRef.shape == (nMap, nC, 4)
    pool = np.ascontiguousarray(segPool, dtype=np.int32)
    assert pool.ndim == 3 and pool.shape[0] == nC and pool.shape[2] == 4
    if refStatic is not None:
        assert refStatic.dtype == np.uint8 and refStatic.flags.c_contiguous and refStatic.shape == (nMap, nC)
    fr = np.ascontiguousarray(firstFrame, dtype=np.int32)
    if slot2map is not None:
        assert slot2map.dtype == np.int32 and slot2map.flags.c_contiguous and slot2map.shape == (nC, N)
    nf = C.c_int(0)
    n = L.opu_map_points_classify_ref(nC, N, nH, _p(Ks), _p(iKs), _p(histR), _p(histT), _p(histXY), _p(sp), _p(featStatic),
                                      _p(slot2map) if slot2map is not None else None, nMap, _p(pointFeat), _p(featRef), _p(pool),
                                      int(pool.shape[1]), _p(refStatic) if refStatic is not None else None, int(curFrame), _p(mapPts),
                                      _p(mapCov), _p(mapFlags), _p(newPt), _p(staticFrameNum), _p(fr), C.c_double(pixelVar), C.byref(nf))
    return n, nf.value


def check_unify(Ks, iKs, histR, histT, histXY, trackSpan, pf1, pf2, M1, M2, sigma, cmpAcos=False):
    """opu_check_unify (CoSLAM::checkUnify): pf1 / pf2 int32[nC] = the two points' slots per camera (< 0 none), M1 / M2 their
    positions.  Returns (ok, M, cov)."""
    L = lib()
    L.opu_check_unify.restype = C.c_int
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nC, nH = histR.shape[0], histR.shape[1]
    N = histXY.shape[2] // 2
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    iKs = np.ascontiguousarray(iKs, dtype=np.float64).reshape(nC, 9)
    sp = np.ascontiguousarray(trackSpan, dtype=np.int32).reshape(nC, 2 * N)
    a, b = np.ascontiguousarray(pf1, dtype=np.int32).reshape(nC), np.ascontiguousarray(pf2, dtype=np.int32).reshape(nC)
    m1, m2 = np.ascontiguousarray(M1, dtype=np.float64).reshape(3), np.ascontiguousarray(M2, dtype=np.float64).reshape(3)
    M, cov = np.zeros(3), np.zeros(9)
    ok = L.opu_check_unify(nC, N, nH, _p(Ks), _p(iKs), _p(histR), _p(histT), _p(histXY), _p(sp), _p(a), _p(b), _p(m1), _p(m2),
                           C.c_double(sigma), int(bool(cmpAcos)), _p(M), _p(cov))
    return bool(ok), M, cov


def _ref_args(histR, histT, histXY, Ks, iKs):
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nC, nH = histR.shape[0], histR.shape[1]
    N = histXY.shape[2] // 2
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    iKs = np.ascontiguousarray(iKs, dtype=np.float64).reshape(nC, 9)
    return histR, histT, histXY, nC, nH, N, Ks, iKs


def update_new_poses_points_ref(Ks, iKs, histR, histT, histXY, featStatic, featRef, segPool, curFrame, mapPts, mapCov, mapFlags, sigma,
                                lastFrame=None, isCurrent=None, firstKeyFrame=-1, walkCap=64, cmpAcos=False, refStatic=None):
    """opu_update_new_poses_points_ref: update_new_poses_points with the features as REFERENCES -- featRef (nMap x nC x 4 int32:
    slot, frame, first, seg) and segPool (nC x cap x 4: slot, last, first, next): MapPoint::pFeatures with the preFrame chains behind them,
    stale features and re-linked tracks included (oracle/poseupdate_oracle.c, above update_points_core).  refStatic (nMap x nC uint8, optional):
    the features' types as of their own frames (None: featStatic of the slot whatever the feature's age)."""
    L = lib()
    L.opu_update_new_poses_points_ref.restype = C.c_int
    histR, histT, histXY, nC, nH, N, Ks, iKs = _ref_args(histR, histT, histXY, Ks, iKs)
    fs = np.ascontiguousarray(featStatic, dtype=np.uint8).reshape(nC, N)
    fr = np.ascontiguousarray(featRef, dtype=np.int32)
    nMap = fr.shape[0]
    sp = np.ascontiguousarray(segPool, dtype=np.int32)
    assert fr.shape == (nMap, nC, 4) and sp.shape[0] == nC and sp.shape[2] == 4
    assert mapPts.dtype == np.float64 and mapCov.dtype == np.float64 and mapPts.flags.c_contiguous and mapCov.flags.c_contiguous
    fl = np.ascontiguousarray(mapFlags, dtype=np.uint8)
    lf = None if lastFrame is None else np.ascontiguousarray(lastFrame, dtype=np.int32)
    ic = None if isCurrent is None else np.ascontiguousarray(isCurrent, dtype=np.uint8)
    chosen = np.full((nMap, nC), -1, dtype=np.int32)
    ns, nd = C.c_int(0), C.c_int(0)
    rs = None if refStatic is None else np.ascontiguousarray(refStatic, dtype=np.uint8)
    n = L.opu_update_new_poses_points_ref(nC, N, nH, _p(Ks), _p(iKs), _p(histR), _p(histT), _p(histXY), _p(fs), nMap, _p(fr),
                                          _p(rs) if rs is not None else None, _p(sp),
                                          sp.shape[1], int(curFrame), int(walkCap), _p(lf) if lf is not None else None,
                                          _p(ic) if ic is not None else None, int(firstKeyFrame), _p(mapPts), _p(mapCov), _p(fl),
                                          C.c_double(sigma), int(bool(cmpAcos)), _p(chosen), C.byref(ns), C.byref(nd))
    return n, ns.value, nd.value, chosen


def refine_map_points_ref(Ks, iKs, histR, histT, histXY, featRef, segPool, curFrame, mapPts, mapCov, sigma, select=None, walkCap=64,
                          cmpAcos=False):
    """opu_refine_map_points_ref (CoSLAM::refineMapPoint over feature references; layouts as update_new_poses_points_ref)."""
    L = lib()
    L.opu_refine_map_points_ref.restype = C.c_int
    histR, histT, histXY, nC, nH, N, Ks, iKs = _ref_args(histR, histT, histXY, Ks, iKs)
    fr = np.ascontiguousarray(featRef, dtype=np.int32)
    nMap = fr.shape[0]
    sp = np.ascontiguousarray(segPool, dtype=np.int32)
    assert fr.shape == (nMap, nC, 4) and mapPts.dtype == np.float64 and mapCov.dtype == np.float64
    sel = None if select is None else np.ascontiguousarray(select, dtype=np.uint8)
    return L.opu_refine_map_points_ref(nC, N, nH, _p(Ks), _p(iKs), _p(histR), _p(histT), _p(histXY), nMap, _p(fr), _p(sp), sp.shape[1],
                                       int(curFrame), int(walkCap), _p(sel) if sel is not None else None, _p(mapPts), _p(mapCov),
                                       C.c_double(sigma), int(bool(cmpAcos)))


def check_unify_ref(Ks, iKs, histR, histT, histXY, ref1, ref2, segPool, curFrame, M1, M2, sigma, walkCap=64, cmpAcos=False):
    """opu_check_unify_ref (CoSLAM::checkUnify): ref1 / ref2 int32[nC][4] = the two points' feature references.  Returns (ok, M, cov)."""
    L = lib()
    L.opu_check_unify_ref.restype = C.c_int
    histR, histT, histXY, nC, nH, N, Ks, iKs = _ref_args(histR, histT, histXY, Ks, iKs)
    a, b = np.ascontiguousarray(ref1, dtype=np.int32).reshape(nC, 4), np.ascontiguousarray(ref2, dtype=np.int32).reshape(nC, 4)
    sp = np.ascontiguousarray(segPool, dtype=np.int32)
    m1, m2 = np.ascontiguousarray(M1, dtype=np.float64).reshape(3), np.ascontiguousarray(M2, dtype=np.float64).reshape(3)
    M, cov = np.zeros(3), np.zeros(9)
    ok = L.opu_check_unify_ref(nC, N, nH, _p(Ks), _p(iKs), _p(histR), _p(histT), _p(histXY), _p(a), _p(b), _p(sp), sp.shape[1],
                               int(curFrame), int(walkCap), _p(m1), _p(m2), C.c_double(sigma), int(bool(cmpAcos)), _p(M), _p(cov))
    return bool(ok), M, cov


def static_check_mergability(K, histR, histT, histXY, slot, length, M, cov, pixelVar):
    """org_static_check_mergability (CoSLAM::staticCheckMergability): histR (nHist x 9), histT (nHist x 3), histXY (nHist x 2N),
    entry 0 = this frame; the track of `slot` covers the `length` newest entries.  Returns True / False."""
    L = lib()
    L.org_static_check_mergability.restype = C.c_int
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nH = len(histR)
    N = histXY.shape[1] // 2
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(3)
    cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(9)
    return bool(L.org_static_check_mergability(_p(K), nH, _p(histR), _p(histT), _p(histXY), N, int(slot), int(length), _p(M), _p(cov),
                                               C.c_double(pixelVar)))


def register_mergability_cam(K, histR, histT, histXY, trackSpan, Ms, covs, slot, pixelVar):
    """org_register_mergability_cam: staticCheckMergability for one camera's candidates (slot int32[P]); returns uint8[P]
    (1 mergeable, 0 not, 255 no candidate, 2 every frame of the history passes but the track is longer than the history)."""
    L = lib()
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    sp = np.ascontiguousarray(trackSpan, dtype=np.int32)
    Ms = np.ascontiguousarray(Ms, dtype=np.float64).reshape(-1, 3)
    covs = np.ascontiguousarray(covs, dtype=np.float64).reshape(len(Ms), 9)
    sl = np.ascontiguousarray(slot, dtype=np.int32).reshape(-1)
    out = np.zeros(len(Ms), dtype=np.uint8)
    L.org_register_mergability_cam(_p(K), len(histR), _p(histR), _p(histT), _p(histXY), len(sp) // 2, _p(sp), len(Ms), _p(Ms), _p(covs),
                                   _p(sl), 1, C.c_double(pixelVar), _p(out))
    return out


def ncc_blocks(img, x, y, scale):
    """onc_block_compute for n points: returns (blocks uint8[n,128] (121 used, pad 0x80), abc float64[n,4], valid int32[n])."""
    L = lib()
    L.onc_block_compute.restype = C.c_int
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    n = len(x)
    blocks = np.full((n, 128), 0x80, dtype=np.uint8)
    abc = np.zeros((n, 4))
    valid = np.zeros(n, dtype=np.int32)
    for i in range(n):
        valid[i] = L.onc_block_compute(_p(img), W, H, C.c_double(x[i]), C.c_double(y[i]), C.c_double(scale),
                                       blocks[i].ctypes.data_as(C.c_void_p), abc[i].ctypes.data_as(C.c_void_p))
        if not valid[i]:
            blocks[i] = 0x80
            abc[i] = 0
    return blocks, abc, valid


def resize_linear_u8(img, fx, fy):
    """onc_resize_linear_u8: cv::resize(img, Size(), fx, fy) with INTER_LINEAR on 8-bit (restated)"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    wd, hd = C.c_int(0), C.c_int(0)
    lib().onc_resize_dims(W, H, C.c_double(fx), C.c_double(fy), C.byref(wd), C.byref(hd))
    out = np.zeros((hd.value, wd.value), np.uint8)
    lib().onc_resize_linear_u8(_p(img), W, H, C.c_double(fx), C.c_double(fy), _p(out))
    return out


def get_rect_sub_pix_u8(img, cx, cy, win=11):
    """onc_get_rect_sub_pix_u8: cv::getRectSubPix(img, Size(win, win), (cx, cy)) 8u -> 8u (restated)"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    out = np.zeros((win, win), np.uint8)
    lib().onc_get_rect_sub_pix_u8(_p(img), W, H, C.c_double(cx), C.c_double(cy), int(win), _p(out))
    return out


def get_ncc_blocks(img, x, y, scale):
    """onc_get_ncc_blocks: getNCCBlocks (reference src/slam/SL_NCCBlock.cpp:79-155) -> (blocks uint8[n,128], abc float64[n,4])"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    n = len(x)
    wd, hd = C.c_int(W), C.c_int(H)
    lib().onc_resize_dims(W, H, C.c_double(scale), C.c_double(scale), C.byref(wd), C.byref(hd))
    small = np.zeros(max(wd.value * hd.value, 1), np.uint8)
    blocks, abc = np.zeros((n, 128), np.uint8), np.zeros((n, 4))
    lib().onc_get_ncc_blocks(_p(img), W, H, n, _p(x), _p(y), C.c_double(scale), _p(small), _p(blocks), _p(abc))
    return blocks, abc


def ncc_epi_mat(F, x1, y1, blk1, abc1, valid1, x2, y2, blk2, abc2, valid2, epiMax, nccMin, wNone=-1.0):
    """onc_epi_ncc_mat: returns (epiMat, nccMat), M x N float64."""
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (x1, y1, x2, y2)]
    M, N = len(a[0]), len(a[2])
    b1, b2 = np.ascontiguousarray(blk1, dtype=np.uint8), np.ascontiguousarray(blk2, dtype=np.uint8)
    c1, c2 = np.ascontiguousarray(abc1, dtype=np.float64), np.ascontiguousarray(abc2, dtype=np.float64)
    v1, v2 = np.ascontiguousarray(valid1, dtype=np.int32), np.ascontiguousarray(valid2, dtype=np.int32)
    epi, ncc = np.zeros((M, N)), np.zeros((M, N))
    lib().onc_epi_ncc_mat(_p(F), M, _p(a[0]), _p(a[1]), _p(b1), _p(c1), _p(v1), N, _p(a[2]), _p(a[3]), _p(b2), _p(c2), _p(v2),
                          C.c_double(epiMax), C.c_double(nccMin), C.c_double(wNone), _p(epi), _p(ncc))
    return epi, ncc


def posegraph_edges(nodeR, nodeT, id1, id2):
    """opg_rigid_from_to for every edge (getRigidTransFromTo of the two ends' poses): returns (edgeR [E,9], edgeT [E,3])."""
    L = lib()
    nodeR = np.ascontiguousarray(nodeR, dtype=np.float64).reshape(-1, 9)
    nodeT = np.ascontiguousarray(nodeT, dtype=np.float64).reshape(-1, 3)
    E = len(id1)
    eR, eT = np.zeros((E, 9)), np.zeros((E, 3))
    for e in range(E):
        i, j = int(id1[e]), int(id2[e])
        L.opg_rigid_from_to(_p(nodeR[i]), _p(nodeT[i]), _p(nodeR[j]), _p(nodeT[j]), eR[e].ctypes.data_as(C.c_void_p),
                            eT[e].ctypes.data_as(C.c_void_p))
    return eR, eT


def posegraph_relax(fixed, nodeR, nodeT, id1, id2, edgeR, edgeT):
    """opg_relax on ONE graph: returns (rc, newR [N,9], newT [N,3])."""
    L = lib()
    L.opg_relax.restype = C.c_int
    fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
    nodeR = np.ascontiguousarray(nodeR, dtype=np.float64).reshape(-1, 9)
    nodeT = np.ascontiguousarray(nodeT, dtype=np.float64).reshape(-1, 3)
    id1 = np.ascontiguousarray(id1, dtype=np.int32)
    id2 = np.ascontiguousarray(id2, dtype=np.int32)
    edgeR = np.ascontiguousarray(edgeR, dtype=np.float64).reshape(-1, 9)
    edgeT = np.ascontiguousarray(edgeT, dtype=np.float64).reshape(-1, 3)
    N, E = len(fixed), len(id1)
    newR, newT = np.zeros((N, 9)), np.zeros((N, 3))
    rc = L.opg_relax(N, E, _p(fixed), _p(nodeR), _p(nodeT), _p(id1), _p(id2), _p(edgeR), _p(edgeT), _p(newR), _p(newT))
    return rc, newR, newT


def approx_rotation(M):
    """opg_approx_rotation: U V^T of the 3x3 M."""
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(9)
    out = np.zeros(9)
    lib().opg_approx_rotation(_p(M), _p(out))
    return out.reshape(3, 3)


def set_threshold_margin_buffer(buf):
    """buf: float32[N] preset to a large value (kept alive by the caller), or None to switch the diagnostic off."""
    lib().okl_set_threshold_margin_buffer(_p(buf) if buf is not None else None)


class SequenceTracker:
    """okl_seq: CPU restatement of V3D_GPU::KLT_SequenceTracker."""

    def __init__(self, config, centered=0, sum_mode=0):
        """sum_mode 0: window sums serially, as the shader writes them; 1 ("tree"): in the HIP tracker's fixed order
        (okl_track_gain_pass_tree) -- tracking must then agree with the HIP path bit for bit."""
        self._L = lib()
        self.cfg = Config.from_any(config)
        self._h = C.c_void_p(self._L.okl_seq_create(C.byref(self.cfg), centered))
        self._L.okl_seq_set_sum_mode(self._h, int(sum_mode))

    def allocate(self, W, H, L, fw, fh, plw=0, plh=0):
        if plw <= 0 or plh <= 0:
            plw, plh = 2 * fw, 2 * fh
        self._L.okl_seq_allocate(self._h, W, H, L, fw, fh, plw, plh)
        self.W, self.H, self.L, self.fw, self.fh, self.N = W, H, L, fw, fh, fw * fh

    def clone(self, sum_mode=None):
        """deep copy of the whole tracker state (optionally switched to another summation mode)"""
        c = object.__new__(SequenceTracker)
        c._L, c.cfg = self._L, self.cfg
        self._L.okl_seq_clone.restype = C.c_void_p
        c._h = C.c_void_p(self._L.okl_seq_clone(self._h))
        c.W, c.H, c.L, c.fw, c.fh, c.N = self.W, self.H, self.L, self.fw, self.fh, self.N
        if sum_mode is not None:
            self._L.okl_seq_set_sum_mode(c._h, int(sum_mode))
        return c

    def close(self):
        if self._h:
            self._L.okl_seq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setBorderMargin(self, m):
        self._L.okl_seq_set_border_margin(self._h, C.c_float(m))

    def setConvergenceThreshold(self, t):
        self._L.okl_seq_set_convergence_threshold(self._h, C.c_float(t))

    def setSSD_Threshold(self, t):
        self._L.okl_seq_set_ssd_threshold(self._h, C.c_float(t))

    def _call(self, fn, image, *extra):
        img = np.ascontiguousarray(image, dtype=np.uint8)
        assert img.size == self.W * self.H
        dest = np.zeros(self.N, dtype=TrackedFeature)
        dest["status"] = -1
        dest["fed"] = -1
        n = C.c_int(0)
        fn(self._h, _p(img), C.byref(n), _p(dest), *extra)
        return n.value, dest

    def detect(self, image, present=None):
        if present is None:
            return self._call(self._L.okl_seq_detect, image)
        p = np.ascontiguousarray(present, dtype=np.float32).reshape(-1, 3)
        return self._call(self._L.okl_seq_detect_present, image, p.shape[0], _p(p))

    def redetect(self, image):
        return self._call(self._L.okl_seq_redetect, image)

    def track(self, image):
        return self._call(self._L.okl_seq_track, image)

    def feedExternFeaturePoints(self, featPts):
        p = np.ascontiguousarray(featPts, dtype=np.float32).reshape(-1, 3)
        ids = np.full(max(p.shape[0], 1), -1, dtype=np.int32)
        n = C.c_int(0)
        self._L.okl_seq_feed(self._h, p.shape[0], _p(p), _p(ids), C.byref(n))
        return n.value, ids[: n.value].copy()

    def advanceFrame(self):
        self._L.okl_seq_advance(self._h)

    def read_pyramid(self):
        total, _ = pyr_layout(self.W, self.H, self.L)
        ptr = self._L.okl_seq_cur_pyramid(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(total * 4,)).copy()

    def read_cornerness(self):
        ptr = self._L.okl_seq_cornerness(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(self.H, self.W)).copy()

    def read_features(self):
        out = np.zeros((self.N, 3), dtype=np.float32)
        self._L.okl_seq_read_features(self._h, _p(out))
        return out


# ---------------------------------------------------------------------------------------------------
# intra-camera pose (pose_oracle.c) and the reference's own code (oracle/_ref/libintracam_ref.so)
class PoseOption(C.Structure):
    _fields_ = [
        ("maxIterLM", C.c_int), ("maxIterRW", C.c_int),
        ("epsErrorChangeLM", C.c_double), ("epsParamChangeLM", C.c_double), ("epsErrorChangeRW", C.c_double),
        ("verboseLM", C.c_int), ("verboseRW", C.c_int),
        ("lambda0", C.c_double), ("lambda_", C.c_double),
        ("err0", C.c_double), ("err", C.c_double), ("errRW", C.c_double),
        ("retTypeLM", C.c_int), ("npts", C.c_int), ("nIterLM", C.c_int), ("nIterRW", C.c_int),
    ]


def _dd(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def intracam_estimate(K, R0, t0, npts, prevErrs, Ms, ms, tau):
    """okp_intracam_estimate -> (ok, R[3,3], t[3], opt)"""
    L = lib()
    K, R0, t0, Ms, ms = _dd(K).ravel(), _dd(R0).ravel(), _dd(t0).ravel(), _dd(Ms).ravel(), _dd(ms).ravel()
    pe = None if prevErrs is None else _dd(prevErrs).ravel()
    R, t = np.zeros(9), np.zeros(3)
    o = PoseOption()
    L.okp_option_default(C.byref(o))
    ok = L.okp_intracam_estimate(_p(K), _p(R0), _p(t0), int(npts), None if pe is None else _p(pe), _p(Ms), _p(ms),
                                 C.c_double(tau), _p(R), _p(t), C.byref(o))
    return bool(ok), R.reshape(3, 3), t, o


_ref = None


def have_ref():
    return os.path.exists(REF_PATH)


def ref_intracam_estimate(K, R0, t0, npts, prevErrs, Ms, ms, tau):
    """The reference's own intraCamEstimate (SL_IntraCamPose.cpp compiled in place into oracle/_ref)."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_PATH)
    K, R0, t0, Ms, ms = _dd(K).ravel(), _dd(R0).ravel(), _dd(t0).ravel(), _dd(Ms).ravel(), _dd(ms).ravel()
    pe = None if prevErrs is None else _dd(prevErrs).ravel()
    R, t, st = np.zeros(9), np.zeros(3), np.zeros(8)
    ok = _ref.ref_intraCamEstimate(_p(K), _p(R0), _p(t0), int(npts), None if pe is None else _p(pe), _p(Ms), _p(ms),
                                   C.c_double(tau), _p(R), _p(t), _p(st))
    stats = dict(err=st[0], errRW=st[1], lambda_=st[2], nIterLM=int(st[3]), nIterRW=int(st[4]), retTypeLM=int(st[5]),
                 err0=st[6], lambda0=st[7])
    return bool(ok), R.reshape(3, 3), t, stats


# ---------------------------------------------------------------------------------------------------
# robust bundle adjustment (ba_oracle.c)
class BAStats(C.Structure):
    _fields_ = [("cost0", C.c_double), ("cost", C.c_double), ("nIterTotal", C.c_int), ("nOuter", C.c_int),
                ("nOutliers", C.c_int), ("pad", C.c_int)]


def csr_by_point(P, obs_pt, obs_cam, obs_xy):
    """Group measurements by point (stable): returns obs_ptr[P+1], obs_cam, obs_xy, order."""
    obs_pt = np.asarray(obs_pt, dtype=np.int64)
    order = np.argsort(obs_pt, kind="stable")
    ptr = np.zeros(P + 1, dtype=np.int32)
    np.add.at(ptr, obs_pt + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    return ptr, np.ascontiguousarray(np.asarray(obs_cam, dtype=np.int32)[order]), \
        np.ascontiguousarray(np.asarray(obs_xy, dtype=np.float64)[order]), order


def ba_robust(Ks, Rs, Ts, pts, obs_ptr, obs_cam, obs_xy, nCamsCon, nPtsCon, maxErr, maxIter, innerMaxIter):
    """oba_robust on copies -> (Rs, Ts, pts, outlier, stats)"""
    L = lib()
    Cn, P = len(Rs), len(pts)
    Ks = _dd(Ks).reshape(Cn, 9)
    Rs = _dd(Rs).reshape(Cn, 9).copy()
    Ts = _dd(Ts).reshape(Cn, 3).copy()
    pts = _dd(pts).reshape(P, 3).copy()
    obs_ptr = np.ascontiguousarray(obs_ptr, dtype=np.int32)
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
    obs_xy = _dd(obs_xy)
    n = len(obs_cam)
    out = np.zeros(max(n, 1), dtype=np.int32)
    st = BAStats()
    L.oba_robust(Cn, P, n, _p(Ks), _p(Rs), _p(Ts), _p(pts), _p(obs_ptr), _p(obs_cam), _p(obs_xy), int(nCamsCon),
                 int(nPtsCon), C.c_double(maxErr), int(maxIter), int(innerMaxIter), _p(out), C.byref(st))
    return Rs.reshape(Cn, 3, 3), Ts, pts, out[:n], st


def ba_residual(K, R, t, M, m, jac=True):
    """oba_residual (oracle/ba_oracle.c): e = m - pi(K(RM+t)) and, optionally, Jc (2x6) and Jp (2x3)."""
    L = lib()
    K, R, t, M, m = (_dd(a).reshape(-1) for a in (K, R, t, M, m))
    e, Jc, Jp = np.zeros(2), np.zeros(12), np.zeros(6)
    vp = C.c_void_p
    ok = L.oba_residual(K.ctypes.data_as(vp), R.ctypes.data_as(vp), t.ctypes.data_as(vp), M.ctypes.data_as(vp),
                        m.ctypes.data_as(vp), e.ctypes.data_as(vp), Jc.ctypes.data_as(vp) if jac else None,
                        Jp.ctypes.data_as(vp) if jac else None)
    return bool(ok), e, Jc.reshape(2, 6), Jp.reshape(2, 3)


def parse_inputs_window(key_frames, map_pts, map_static=None):
    """RobustBundleRTS::addKeyFrames + addPoints + parseInputs (reference src/app/SL_CoSLAMRobustBA.cpp:37-78,109-165) restated
    over structure-of-arrays key-frame records (numpy; test infrastructure).  key_frames: oldest first, each a list over the
    cameras of dicts(xy float64[2N] (x[N] then y[N]), state int[N], slot2map int[N], K[9], R[9], t[3]).  A feature point is a
    slot with state 0 / 1 whose slot2map names a (static) map point; FeaturePoints lists are in slot order
    (GPUKLT::addToFeaturePoints), so for one (key frame, camera) a later slot of the same map point replaces an earlier one
    (vecFeatPts[camId] = fpt, :141) -- but BOTH count towards `nfpts > 1` (:120-121: the size of the point's list)...
    except that CoSLAM never maps two features of one frame and camera to one map point (MapPoint::pFeatures[camId] is one
    pointer); the device path counts a (key frame, camera) once, and so does this restatement.
    Returns dict(Ks, Rs, Ts, pts, obs_ptr, obs_cam, obs_xy, point_map)."""
    n_cams = len(key_frames[0])
    n_map = len(map_pts)
    cams = [(j, c) for j in range(len(key_frames)) for c in range(n_cams)]          # addKeyFrames: key frame x camera
    feat = np.full((len(cams), n_map), -1, dtype=np.int64)                          # (camera index, map point) -> slot
    for ci, (j, c) in enumerate(cams):
        rec = key_frames[j][c]
        for s_ in range(len(rec["state"])):                                         # list order = slot order: the last one stays
            m = int(rec["slot2map"][s_])
            if rec["state"][s_] in (0, 1) and 0 <= m < n_map and (map_static is None or map_static[m]):
                feat[ci, m] = s_
    Ks = np.stack([np.asarray(key_frames[j][c]["K"], float).reshape(9) for j, c in cams])
    Rs = np.stack([np.asarray(key_frames[j][c]["R"], float).reshape(9) for j, c in cams])
    Ts = np.stack([np.asarray(key_frames[j][c]["t"], float).reshape(3) for j, c in cams])
    pts, ptr, ocam, oxy, pmap = [], [0], [], [], []
    for m in range(n_map):                                                          # std::map<MapPoint*, ...>: address = index order
        seen = np.nonzero(feat[:, m] >= 0)[0]
        if len(seen) <= 1:                                                          # nfpts > 1 (:120-121)
            continue
        pts.append(map_pts[m])
        pmap.append(m)
        for ci in seen:                                                             # camera order (:146-151)
            j, c = cams[ci]
            rec, s_ = key_frames[j][c], int(feat[ci, m])
            n = len(rec["state"])
            ocam.append(ci)
            oxy.append((rec["xy"][s_], rec["xy"][n + s_]))
        ptr.append(len(ocam))
    return dict(Ks=Ks, Rs=Rs, Ts=Ts, pts=np.asarray(pts, float).reshape(-1, 3), obs_ptr=np.asarray(ptr, np.int32),
                obs_cam=np.asarray(ocam, np.int32), obs_xy=np.asarray(oxy, float).reshape(-1, 2), point_map=np.asarray(pmap, np.int32))


def parse_inputs_window_fast(key_frames, map_pts, map_static=None):
    """parse_inputs_window with the per-slot loops as numpy array operations (what bench.py's CPU baseline times: a Python loop over
    80 000 slots per key frame would not be a fair CPU figure).  Same arguments, same result, array for array
    (tests/test_oracle_cpu.py holds the two against each other)."""
    n_cams = len(key_frames[0])
    n_map = len(map_pts)
    cams = [(j, c) for j in range(len(key_frames)) for c in range(n_cams)]
    feat = np.full((len(cams), n_map), -1, dtype=np.int64)
    for ci, (j, c) in enumerate(cams):
        rec = key_frames[j][c]
        st, m = np.asarray(rec["state"]), np.asarray(rec["slot2map"])
        ok = ((st == 0) | (st == 1)) & (m >= 0) & (m < n_map)
        if map_static is not None:
            ok &= np.asarray(map_static)[np.clip(m, 0, n_map - 1)].astype(bool)
        idx = np.nonzero(ok)[0]
        feat[ci, m[idx]] = idx                                  # (slot order: for a repeated map point the last slot stays)
    Ks = np.stack([np.asarray(key_frames[j][c]["K"], float).reshape(9) for j, c in cams])
    Rs = np.stack([np.asarray(key_frames[j][c]["R"], float).reshape(9) for j, c in cams])
    Ts = np.stack([np.asarray(key_frames[j][c]["t"], float).reshape(3) for j, c in cams])
    seen = feat >= 0
    keep = np.nonzero(seen.sum(0) > 1)[0]                      # nfpts > 1 (:120-121), map index order
    sub = seen[:, keep]                                         # [camera][kept point]
    pi, ci = np.nonzero(sub.T)                                  # per kept point its cameras in camera order (:146-151)
    slots = feat[ci, keep[pi]]
    N = np.array([len(key_frames[j][c]["state"]) for j, c in cams])
    xy_all = [np.asarray(key_frames[j][c]["xy"], float) for j, c in cams]
    oxy = np.empty((len(ci), 2))
    for q in range(len(cams)):                                  # (one gather per camera, not per measurement)
        sel = np.nonzero(ci == q)[0]
        oxy[sel, 0], oxy[sel, 1] = xy_all[q][slots[sel]], xy_all[q][N[q] + slots[sel]]
    ptr = np.concatenate([[0], np.cumsum(sub.sum(0))]).astype(np.int32)
    return dict(Ks=Ks, Rs=Rs, Ts=Ts, pts=np.asarray(map_pts, float)[keep].reshape(-1, 3), obs_ptr=ptr, obs_cam=ci.astype(np.int32),
                obs_xy=oxy, point_map=keep.astype(np.int32))


def intercam_add_map_points(W, H, nColBlk, nRowBlk, ptsStride, xy, state, slot2map, trackSpan, isStatic, mapPts, mapFlags, newPt, pointFeat,
                            maxDyn=60):
    """InterCamPoseEstimator::addMapPoints (reference src/app/SL_InterCamPoseEstimator.cpp:18-91) restated over structure-of-arrays
    records (TEST INFRASTRUCTURE; integer / index work, numpy): per camera xy (2N: x then y), state (N: 0 / 1 = the slot's track has
    a feature in this frame), slot2map (N), trackSpan (2N: first, last frame), isStatic (N: FeaturePoint::type); mapFlags (CS_MAP_*
    bytes), newPt (MapPoint::bNewPt), pointFeat (nMap x nCams: slot of the point's feature of THIS frame or -1).
    Returns dict(pts [P,3], obs_ptr [P+1], obs_cam, obs_xy [nObs,2], point_map [P], n_static): vecPts3D / vecMeas2D flattened."""
    nC = len(xy)
    N = len(state[0])
    blkW, blkH = W // nColBlk, H // nRowBlk     # src/app/SL_SingleSLAM.cpp:270-271
    pf = np.asarray(pointFeat)
    pts, ptr, cam, oxy, pmap = [], [0], [], [], []
    for c in range(nC):
        tracks = {}                              # chooseStaticFeatPts, src/app/SL_SingleSLAM.cpp:345-397
        for s in range(N):
            if state[c][s] not in (0, 1):        # tk->empty()
                continue
            m = int(slot2map[c][s])
            certain_static = m >= 0 and (int(mapFlags[m]) & 7) == 0
            if not (isStatic[c][s] or certain_static):
                continue
            bx, by = int(xy[c][s] / blkW), int(xy[c][N + s] / blkH)
            if bx >= nColBlk or by >= nRowBlk or bx < 0 or by < 0:
                continue
            bi = by * nColBlk + bx
            f1, f2 = int(trackSpan[c][s]), int(trackSpan[c][N + s])
            ln = f2 - f1 + 1 if f1 >= 0 else 1
            if bi not in tracks:
                tracks[bi] = (s, m, ln)
            else:
                so, mo, lo = tracks[bi]
                if mo < 0 and (m >= 0 or lo < ln):   # :372-381
                    tracks[bi] = (s, m, ln)
        k = 0
        for bi in sorted(tracks):                # featPts in block order (:386-394); :39-52 static points: one measurement each
            s, m, _ = tracks[bi]
            if m < 0 or k >= ptsStride:          # `if (!fp->mpt) continue`
                continue
            pts.append(mapPts[m]), pmap.append(m)
            cam.append(c), oxy.append((xy[c][s], xy[c][N + s]))
            ptr.append(len(cam))
            k += 1
    n_static = len(pts)
    dyn = set()
    for c in range(nC):                          # chooseDynamicFeatPts, src/app/SL_SingleSLAM.cpp:398-447
        best = {}
        for s in range(N):
            if state[c][s] not in (0, 1):
                continue
            m = int(slot2map[c][s])
            if m < 0:
                continue
            nvis = int((pf[m] >= 0).sum())       # MapPoint::numVisCam as updateVisCamNum(curFrame) leaves it
            if nvis < 2:
                continue
            fl = int(mapFlags[m])
            unc = bool(fl & 4)
            certain_dyn = (not unc) and (fl & 3) == 1
            if not (certain_dyn or (unc and newPt[m])):
                continue
            bx, by = int(xy[c][s] / blkW), int(xy[c][N + s] / blkH)
            if bx >= nColBlk or by >= nRowBlk or bx < 0 or by < 0:
                continue
            bi = by * nColBlk + bx
            if bi not in best or best[bi][0] < nvis:   # `if (fpOld->mpt->numVisCam < fp->mpt->numVisCam)`: the first of the most visible
                best[bi] = (nvis, s)
        for nvis, s in best.values():
            dyn.add(int(slot2map[c][s]))
    k = 0
    for m in sorted(dyn):                        # std::map<MapPoint*, int>: address order = map order
        if k > maxDyn:                           # :72
            continue
        pts.append(mapPts[m]), pmap.append(m)
        for c in range(nC):
            s = int(pf[m, c])
            if s >= 0:                           # `fp && fp->f == curFrame`
                cam.append(c), oxy.append((xy[c][s], xy[c][N + s]))
        ptr.append(len(cam))
        k += 1
    return dict(pts=np.array(pts, dtype=np.float64).reshape(-1, 3), obs_ptr=np.array(ptr, dtype=np.int32), obs_cam=np.array(cam, dtype=np.int32),
                obs_xy=np.array(oxy, dtype=np.float64).reshape(-1, 2), point_map=np.array(pmap, dtype=np.int32), n_static=n_static)


def register_decide_static(slot, flags, mergeable, mapFlags, pointFeat, slot2map, map_base=0, kinds=1):
    """The decision half of CoSLAM::curStaticPointsRegInGroup / curStaticPointRegInGroup with bMerge == false (reference
    src/app/SL_CoSLAM.cpp:854-898, 731-830) restated over the search tables (TEST INFRASTRUCTURE; index work, plain Python): for every
    camera o in order, the certainly static points with a feature of this frame in o, in map order; each walks the cameras in order --
    skipping those where it has a feature of this frame, where the search found nothing (slot < 0) or a DYNAMIC feature -- and
    attaches the nearest feature when that is unmapped and mergeable over its whole track (staticCheckMergability; compareFeaturePt is
    always true, :546-558); a feature that already carries a map point ENDS the point's walk (`if (!bMerge) return bReg`, :789-790).
    slot / flags / mergeable: P x C tables of the search (flags bit 1: the candidate is dynamic); mapFlags [P] CS_MAP_* bytes;
    pointFeat [P][C] and slot2map [C][N] are updated IN PLACE (a track's features all take the point: slot2map is per track).
    Returns (attached [P][C] uint8, regged [P] uint8: the points refineMapPoint is called for, :889-893).
    This is the SINGLE PASS the kernels run (every walk against the tables of one search); the reference's own run is pinned by
    register_cur_static_sequential below, from which this differs only where a registered point is visited again."""
    P, C = slot.shape
    attached = np.zeros((P, C), dtype=np.uint8)
    regged = np.zeros(P, dtype=np.uint8)
    # kinds bit 0: the certainly static points (curStaticPointsRegInGroup); bit 1: behind them the certainly DYNAMIC ones
    # (curDynamicPointsRegInGroup, :904-1020: the same walk over DYNAMIC features only -- a static feature is passed by, a dynamic one that
    # carries a point ends the walk; currentMapPointsRegister's order, :834-853).  The two kinds never meet at a feature.
    for want_dyn, o in [(k, o_) for k in (0, 1) if kinds & (1 << k) for o_ in range(C)]:
        vec = [p for p in range(P) if (int(mapFlags[p]) & 7) == want_dyn and pointFeat[p, o] >= 0]   # :864-869 / :917-922
        for p in vec:
            breg = False
            for i in range(C):
                if pointFeat[p, i] >= 0:                 # :736-737
                    continue
                s = int(slot[p, i])
                if s < 0:                                # behind the camera / outside the image / no feature
                    continue
                if ((int(flags[p, i]) >> 1) & 1) != want_dyn:   # `pFeat->type != TYPE_FEATPOINT_DYNAMIC` (:757) / `== ...DYNAMIC` (:981)
                    continue
                if slot2map[i][s] < 0:                   # `pFeat->mpt == 0`, as it is NOW
                    if mergeable[p, i] == 1:
                        slot2map[i][s] = map_base + p    # the feature and its predecessors (:771-775), MapPoint::addFeature
                        pointFeat[p, i] = s
                        attached[p, i] = 1
                        breg = True
                else:
                    break                                # :789-790
            if breg:
                regged[p] = 1
    return attached, regged


def register_decide_static_c(slot, flags, mergeable, mapFlags, pointFeat, slot2map, map_base=0, kinds=1):
    """org_register_decide: register_decide_static's walks in C (the CPU baseline's leg; checked against the Python restatement).
    slot2map: ONE int32 array [nCams][N], updated in place like pointFeat.  Returns (attached, regged)."""
    P, nC = slot.shape
    N = slot2map.shape[1]
    sl, fl = np.ascontiguousarray(slot, dtype=np.int32), np.ascontiguousarray(flags, dtype=np.int32)
    mg, mf = np.ascontiguousarray(mergeable, dtype=np.uint8), np.ascontiguousarray(mapFlags, dtype=np.uint8)
    assert pointFeat.dtype == np.int32 and pointFeat.flags.c_contiguous and slot2map.dtype == np.int32 and slot2map.flags.c_contiguous
    att, reg = np.zeros((P, nC), dtype=np.uint8), np.zeros(P, dtype=np.uint8)
    L = lib()
    L.org_register_decide.restype = C.c_int
    L.org_register_decide(P, nC, N, _p(sl), _p(fl), _p(mg), _p(mf), _p(pointFeat), _p(slot2map), int(map_base), int(kinds), _p(att), _p(reg))
    return att, reg


def register_cur_static_sequential(W, H, Ks, iKs, histR, histT, histXY, trackSpan, state, isStatic, slot2map, mapPts, mapCov, mapFlags,
                                   pointFeat, pixelVar, with_dynamic=False, merge=False):
    """CoSLAM::curStaticPointsRegInGroup (bMerge == false) AS THE REFERENCE RUNS IT (src/app/SL_CoSLAM.cpp:854-898, 731-830), one point after
    the other (TEST INFRASTRUCTURE, plain Python over the pinned restatements of the search, staticCheckMergability and refineMapPoint):
    for every camera o in turn, the certainly static points that hold a feature of this frame in o -- INCLUDING features attached in an
    earlier camera's loop -- in map order; each is projected with its position AS IT STANDS (refined at the end of every loop in which it
    gained a feature, :889-893), walks the cameras, attaches the nearest unmapped, non-dynamic, mergeable feature, and stops at a feature
    that carries a point.  This is what tests/golden/decide_golden.npz pins bit for bit; cs_register_decide_static_dev /
    register_decide_static evaluate every walk against the positions at the START of the frame (one search, one decision, one refine)
    and differ where a point that gained a feature is visited again in a later camera's loop.
    histR / histT / histXY / trackSpan / pointFeat as refine_map_points takes them (entry 0 = this frame); slot2map (list per camera),
    pointFeat, mapPts, mapCov are updated IN PLACE.  Returns (features attached, the reference's return value: registrations summed
    over the camera loops)."""
    nC, nP = len(state), len(mapPts)
    xy0 = [np.ascontiguousarray(histXY[c][0]) for c in range(nC)]
    is_dyn = [(1 - np.asarray(isStatic[c])).astype(np.uint8) for c in range(nC)]
    n_att, n_reg_total = 0, 0
    # with_dynamic: curDynamicPointsRegInGroup behind the static points' loops (currentMapPointsRegister, :834-853): the certainly dynamic
    # points, maxDist 4 sigma (:973), DYNAMIC candidates only; returns the registrations of both (static, dynamic) then
    n_reg_kind = [0, 0]
    n_merged = 0
    for want_dyn, o in [(0, o_) for o_ in range(nC)] + ([(1, o_) for o_ in range(nC)] if with_dynamic else []):
        vec = [p for p in range(nP) if (int(mapFlags[p]) & 7) == want_dyn and pointFeat[p, o] >= 0]      # :864-869 / :917-922
        regged = []
        for p in vec:
            res = register_search(W, H, Ks, histR[:, 0], histT[:, 0], xy0, state, slot2map, is_dyn, mapPts[p:p + 1], mapCov[p:p + 1],
                                  pointFeat[p:p + 1], pixelVar, (4 if want_dyn else 3) * pixelVar, pixelVar)
            breg = False
            for i in range(nC):
                s = int(res["slot"][0, i])
                if pointFeat[p, i] >= 0 or s < 0 or ((int(res["flags"][0, i]) >> 1) & 1) != want_dyn:
                    continue
                if slot2map[i][s] >= 0:                                                               # :789-790
                    if not (merge and not want_dyn):
                        break
                    # bMerge (:791-826; curDynamicPointRegInGroup ignores the flag): can the two points be one?
                    q = int(slot2map[i][s])
                    if (int(mapFlags[q]) & 3) != 0 or q == p:                                         # !isLocalStatic() / itself
                        continue
                    ok, Mu, covu = check_unify(Ks, iKs, histR, histT, histXY, trackSpan, pointFeat[p], pointFeat[q], mapPts[p], mapPts[q], pixelVar)
                    if not ok:
                        continue
                    mapPts[p], mapCov[p] = Mu, covu                                                    # updatePosition
                    mapFlags[q] = (int(mapFlags[q]) & 4) | 2                                           # numVisCam = 0, setFalse()
                    for v in range(nC):                                                               # its features where p has none ...
                        sq = int(pointFeat[q, v])
                        if sq >= 0 and pointFeat[p, v] < 0:
                            pointFeat[q, v] = -1
                            slot2map[v][sq] = p
                            pointFeat[p, v] = sq
                            if v == i and sq == s:
                                # ... up to the camera of the conflict: the loop reads `pFeat->mpt->pFeatures[v]` (:808) and pFeat->mpt IS p
                                # once pFeat itself has moved -- the other point's features in the cameras behind it stay where they are
                                break
                    n_merged += 1
                    breg = True
                    break
                sl = np.full(1, s, dtype=np.int32)
                if register_mergability_cam(Ks[i], histR[i], histT[i], histXY[i], trackSpan[i], mapPts[p:p + 1], mapCov[p:p + 1], sl, pixelVar)[0] == 1:
                    slot2map[i][s] = p
                    pointFeat[p, i] = s
                    n_att += 1
                    breg = True
            if breg:
                regged.append(p)
        for p in regged:                                                                              # :889-893
            sel = np.zeros(nP, dtype=np.uint8)
            sel[p] = 1
            refine_map_points(Ks, iKs, histR, histT, histXY, trackSpan, pointFeat, mapPts, mapCov, pixelVar, select=sel)
        n_reg_total += len(regged)
        n_reg_kind[want_dyn] += len(regged)
    if merge:
        return n_att, n_reg_kind[0], n_reg_kind[1], n_merged
    return (n_att, n_reg_total) if not with_dynamic else (n_att, n_reg_kind[0], n_reg_kind[1])


def new_map_points_from_pairs(N, pairs, Ks, iKs, Rs, ts, xy, state, slot2map, isStatic, mapPts, mapCov, mapFlags, newPt, firstFrame, pointFeat,
                              map_count, cur_frame, max_disp=80.0, max_rp_err=3.0, sigma=10.0, min_len=2, max_seeds=512, reproj=None, W=640, H=480):
    """NewMapPtsNCC::run + output behind getEpiNccMat (reference src/app/SL_NewMapPointsInterCam.cpp:150-161, 163-192, 194-270, 295-316,
    631-690) restated over structure-of-arrays records (TEST INFRASTRUCTURE, plain Python / numpy, operation for operation what
    coslam_amd/csrc/newpts.hip does): per consecutive camera pair the candidate list (i, j, epi, ncc) of the NCC stage -> seeds,
    disparity guide, greedy matches; the matches chained into tracks; every track of >= min_len views triangulated, gated and appended
    to the map arrays IN PLACE behind map_count.  greedyNCCMatch / greedyGuidedNCCMatch / getDisparityMat are un-vendored: the
    definitions are newpts.hip's header's.  Returns dict(matches [nC-1][N], tracks (list of [(cam, slot)]), new (indices of the new
    points), map_count)."""
    import math

    nC = len(xy)
    cap = len(mapPts)
    pf = pointFeat
    match = np.full((nC - 1, N), -1, dtype=np.int32)
    has_in = np.zeros((nC, N), dtype=bool)
    for a in range(nC - 1):
        b = a + 1
        seeds = []                                       # getSeedsBetween (:97-127), map order
        for m in range(min(map_count, cap)):
            if int(mapFlags[m]) & 6:                     # isFalse() / isUncertain()
                continue
            s1, s2 = int(pf[m, a]), int(pf[m, b])
            if s1 >= 0 and s2 >= 0 and len(seeds) < max_seeds:
                x1, y1 = xy[a][s1], xy[a][N + s1]
                seeds.append((x1, y1, xy[b][s2] - x1, xy[b][N + s2] - y1))
        cand = []
        for (i, j, epi, ncc) in pairs[a]:
            i, j = int(i), int(j)
            if seeds:                                    # getDisparityMat + the guide
                x1, y1 = xy[a][i], xy[a][N + i]
                best, bk = 1.0e300, 0
                for k, sd in enumerate(seeds):
                    dx, dy = sd[0] - x1, sd[1] - y1
                    d2 = dx * dx + dy * dy
                    if d2 < best:
                        best, bk = d2, k
                ex = (xy[b][j] - x1) - seeds[bk][2]
                ey = (xy[b][N + j] - y1) - seeds[bk][3]
                if not math.sqrt(ex * ex + ey * ey) <= max_disp:
                    continue
            cand.append((-float(ncc), i, j))
        cand.sort()                                      # falling score, then rising row, then rising column
        rows, cols = set(), set()
        for _, i, j in cand:
            if i in rows or j in cols:
                continue
            rows.add(i), cols.add(j)
            match[a, i] = j
            has_in[b, j] = True
    tracks = []                                          # featTracksFromMatches (:631-690): numbered by (pair, feature)
    for a in range(nC - 1):
        for i in range(N):
            if match[a, i] < 0 or (a > 0 and has_in[a, i]):
                continue
            tk, c, s = [(a, i)], a, i
            while c < nC - 1 and match[c, s] >= 0:
                s = int(match[c, s])
                c += 1
                tk.append((c, s))
            tracks.append(tk)
    new = []
    for tk in tracks:                                    # reconstructTracks (:194-270)
        if len(tk) < min_len:
            continue
        Nn, g = [0.0] * 6, [0.0] * 3
        for c, s in tk:
            iK, R, t = iKs[c].reshape(9), Rs[c].reshape(9), ts[c]
            mx, my = xy[c][s], xy[c][N + s]
            w = (iK[6] * mx + iK[7] * my) + iK[8]
            x, y = ((iK[0] * mx + iK[1] * my) + iK[2]) / w, ((iK[3] * mx + iK[4] * my) + iK[5]) / w
            a0 = [R[0] - x * R[6], R[1] - x * R[7], R[2] - x * R[8]]
            a1 = [R[3] - y * R[6], R[4] - y * R[7], R[5] - y * R[8]]
            b0, b1 = x * t[2] - t[0], y * t[2] - t[1]
            Nn[0] = Nn[0] + (a0[0] * a0[0] + a1[0] * a1[0])
            Nn[1] = Nn[1] + (a0[0] * a0[1] + a1[0] * a1[1])
            Nn[2] = Nn[2] + (a0[0] * a0[2] + a1[0] * a1[2])
            Nn[3] = Nn[3] + (a0[1] * a0[1] + a1[1] * a1[1])
            Nn[4] = Nn[4] + (a0[1] * a0[2] + a1[1] * a1[2])
            Nn[5] = Nn[5] + (a0[2] * a0[2] + a1[2] * a1[2])
            for q in range(3):
                g[q] = g[q] + (a0[q] * b0 + a1[q] * b1)

        def cof(S):
            c_ = [S[3] * S[5] - S[4] * S[4], S[2] * S[4] - S[1] * S[5], S[1] * S[4] - S[2] * S[3], S[0] * S[5] - S[2] * S[2],
                  S[1] * S[2] - S[0] * S[4], S[0] * S[3] - S[1] * S[1]]
            return c_, (S[0] * c_[0] + S[1] * c_[1]) + S[2] * c_[2]

        with np.errstate(all="ignore"):
            cf, det = cof(Nn)
            M = [np.float64((cf[0] * g[0] + cf[1] * g[1]) + cf[2] * g[2]) / det, np.float64((cf[1] * g[0] + cf[3] * g[1]) + cf[4] * g[2]) / det,
                 np.float64((cf[2] * g[0] + cf[4] * g[1]) + cf[5] * g[2]) / det]
            outlier, S, errs = False, [0.0] * 6, []
            for c, s in tk:
                K, R, t = Ks[c].reshape(9), Rs[c].reshape(9), ts[c]
                X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0]
                Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1]
                Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2]
                u, v, w = (K[0] * X + K[1] * Y) + K[2] * Z, (K[3] * X + K[4] * Y) + K[5] * Z, (K[6] * X + K[7] * Y) + K[8] * Z
                dx, dy = xy[c][s] - u / w, xy[c][N + s] - v / w
                e = np.sqrt(dx * dx + dy * dy)
                errs.append(e)
                if e > max_rp_err or Z < 0:
                    outlier = True
                KR = [(K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j]) + K[3 * i + 2] * R[6 + j] for i in range(3) for j in range(3)]
                ww = w * w
                J = [(KR[j] * w - u * KR[6 + j]) / ww for j in range(3)] + [(KR[3 + j] * w - v * KR[6 + j]) / ww for j in range(3)]
                S[0] = S[0] + (J[0] * J[0] + J[3] * J[3])
                S[1] = S[1] + (J[0] * J[1] + J[3] * J[4])
                S[2] = S[2] + (J[0] * J[2] + J[3] * J[5])
                S[3] = S[3] + (J[1] * J[1] + J[4] * J[4])
                S[4] = S[4] + (J[1] * J[2] + J[4] * J[5])
                S[5] = S[5] + (J[2] * J[2] + J[5] * J[5])
            if outlier:
                continue
            cf, dS = cof(S)
            s2 = sigma * sigma
            cov = [(cf[0] / dS) * s2, (cf[1] / dS) * s2, (cf[2] / dS) * s2, 0, (cf[3] / dS) * s2, (cf[4] / dS) * s2, 0, 0, (cf[5] / dS) * s2]
            cov[3], cov[6], cov[7] = cov[1], cov[2], cov[5]
        if map_count >= cap:
            continue
        m = map_count
        map_count += 1
        mapPts[m], mapCov[m] = M, cov
        n_dyn = sum(1 for c, s in tk if isStatic is not None and not isStatic[c][s])
        mapFlags[m] = 1 if n_dyn > 1 else 4            # setLocalDynamic / setUncertain (:253-264); decidePointType below
        newPt[m], firstFrame[m] = 1, cur_frame
        pf[m, :] = -1
        for k, (c, s) in enumerate(tk):
            pf[m, c] = s
            slot2map[c][s] = m
            if reproj is not None:
                reproj[c][s] = errs[k]
        new.append(m)
    # decidePointType (:25-91): the features of this frame on CERTAIN dynamic map points -- this run's dynamic points included,
    # addFeature gave their features the point already (src/slam/SL_MapPoint.cpp:58-69) -- mark 41 x 41 squares; a new uncertain point
    # with no feature inside one becomes certain static (setLocalStatic() clears bUncertain, SL_MapPoint.cpp:104-109)
    if any(int(mapFlags[m]) == 4 for m in new):
        dyn = [[] for _ in range(nC)]
        for c in range(nC):
            for s_ in range(N):
                m = int(slot2map[c][s_])
                if int(state[c][s_]) in (0, 1) and 0 <= m < cap and int(mapFlags[m]) == 1:
                    dyn[c].append((int(xy[c][s_] + 0.5), int(xy[c][N + s_] + 0.5)))
        for m in new:
            if int(mapFlags[m]) != 4:
                continue
            is_static = True
            for c in range(nC):
                s_ = int(pf[m, c])
                if s_ < 0:
                    continue
                x, y = int(xy[c][s_] + 0.5), int(xy[c][N + s_] + 0.5)
                if not (0 <= x < W and 0 <= y < H):
                    continue
                if any(abs(x - dx) <= 20 and abs(y - dy) <= 20 for dx, dy in dyn[c]):
                    is_static = False
                    break
            if is_static:
                mapFlags[m] = 0
    return dict(matches=match, tracks=tracks, new=new, map_count=map_count)


def keyframe_ready(state, slot2map, R, t, selfR, selfT, keyFrame, keyMapped, mapPts, mapFlags, firstFrame, ratio, minViewAngleDeg, minTranslation):
    """CoSLAM::IsReadyForKeyFrame (/root/reference/src/app/SL_CoSLAM.cpp:1269-1279) for ONE camera, restated in numpy: state / slot2map
    int32[N] (the hand-back's records in slot = feature-list order), the current pose, the last self-motion key pose, the last key pose's
    frame and nMappedPts.  Returns (code, m_nMappedStaticPts, num, center): getCurMapCenterViewFrom (:1224-1247) and
    IsMappedPtsDecreaseBelow (:1249-1268) stop BEFORE the frame's last feature (`fp && fp != pTail`), getNumMappedStaticPts
    (SL_SingleSLAM.cpp:121-136) includes it; the centre is summed in list order as the reference does; angles by acos with the
    reference's PI = 3.14 (SL_SLAMHelper.cpp:208-217).  TEST INFRASTRUCTURE."""
    has = np.nonzero((state == 0) | (state == 1))[0]
    m_all = slot2map[has]
    fl_all = np.where(m_all >= 0, mapFlags[np.clip(m_all, 0, len(mapFlags) - 1)], 0)
    n_static = int(((m_all >= 0) & ((fl_all & 7) == 0)).sum())
    inner = has[:-1] if len(has) else has
    m = slot2map[inner]
    m = m[m >= 0]
    num = int((firstFrame[m] <= keyFrame).sum())
    cen, n = np.zeros(3), 0
    for q in m:
        if not (mapFlags[q] & 2):
            cen = cen + mapPts[q]
            n += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        cen = cen / n
    if num < keyMapped * ratio or num < 30:
        return 1, n_static, num, cen
    C0, C1 = -(selfR.reshape(3, 3).T @ selfT), -(R.reshape(3, 3).T @ t)
    a, b = C0 - cen, C1 - cen
    with np.errstate(invalid="ignore"):
        ang = abs(np.arccos(a @ b / np.sqrt((a @ a) * (b @ b)))) / 3.14 * 180.0
    if ang > minViewAngleDeg:
        return 2, n_static, num, cen
    if np.sqrt(((C0 - C1) ** 2).sum()) > minTranslation:
        return 3, n_static, num, cen
    return 0, n_static, num, cen


def intracam_new_points(K, iK, histR, histT, histXY, state, slot2map, trackSpan, isStatic, minTrackLen, maxEpiErr, sigma, cmpAcos=False):
    """opu_intracam_new_points (SingleSLAM::newMapPoints for one camera): histR (nHist x 9), histT (nHist x 3), histXY (nHist x 2N), entry 0 =
    this frame; state / slot2map int32[N], trackSpan int32[2N], isStatic uint8[N].  Returns (slots, firstFrames, M [n x 3], cov [n x 9])
    in slot order."""
    L = lib()
    L.opu_intracam_new_points.restype = C.c_int
    histR = np.ascontiguousarray(histR, dtype=np.float64)
    histT = np.ascontiguousarray(histT, dtype=np.float64)
    histXY = np.ascontiguousarray(histXY, dtype=np.float64)
    nH, N = histR.shape[0], histXY.shape[1] // 2
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    iK = np.ascontiguousarray(iK, dtype=np.float64).reshape(9)
    st, s2m = np.ascontiguousarray(state, dtype=np.int32), np.ascontiguousarray(slot2map, dtype=np.int32)
    sp, fs = np.ascontiguousarray(trackSpan, dtype=np.int32), np.ascontiguousarray(isStatic, dtype=np.uint8)
    assert histT.shape == (nH, 3) and histXY.shape == (nH, 2 * N) and len(st) == len(s2m) == len(fs) == N and len(sp) == 2 * N
    slot, first = np.zeros(N, np.int32), np.zeros(N, np.int32)
    M, cov = np.zeros((N, 3)), np.zeros((N, 9))
    n = L.opu_intracam_new_points(_p(K), _p(iK), N, nH, _p(histR), _p(histT), _p(histXY), _p(st), _p(s2m), _p(sp), _p(fs), int(minTrackLen),
                                  C.c_double(maxEpiErr), C.c_double(sigma), int(bool(cmpAcos)), _p(slot), _p(first), _p(M), _p(cov))
    return slot[:n].copy(), first[:n].copy(), M[:n].copy(), cov[:n].copy()


def new_map_points_from_pairs_c(N, pairs, Ks, iKs, Rs, ts, xy, state, slot2map, isStatic, mapPts, mapCov, mapFlags, newPt, firstFrame, pointFeat,
                                map_count, cur_frame, max_disp=80.0, max_rp_err=3.0, sigma=10.0, min_len=2, max_seeds=512, W=640, H=480):
    """onc_new_points_from_pairs: new_map_points_from_pairs in C (what bench.py's CPU baseline times).  Same arguments (reproj is not
    kept); the map arrays and slot2map (a list of int32 arrays, one per camera) are updated IN PLACE.  Returns dict(matches, new, map_count)."""
    L = lib()
    L.onc_new_points_from_pairs.restype = C.c_int
    nC = len(xy)
    cap = len(mapPts)
    pl = [np.ascontiguousarray(np.asarray(p, dtype=np.float64).reshape(-1, 4)) for p in pairs]
    pp = (C.c_void_p * max(nC - 1, 1))(*[p.ctypes.data for p in pl])
    npairs = np.asarray([len(p) for p in pl] + [0] * (max(nC - 1, 1) - len(pl)), dtype=np.int32)
    K_ = np.ascontiguousarray(np.stack([np.asarray(k, float).reshape(9) for k in Ks]))
    iK_ = np.ascontiguousarray(np.stack([np.asarray(k, float).reshape(9) for k in iKs]))
    R_ = np.ascontiguousarray(np.stack([np.asarray(r, float).reshape(9) for r in Rs]))
    t_ = np.ascontiguousarray(np.stack([np.asarray(t, float).reshape(3) for t in ts]))
    xy_ = np.ascontiguousarray(np.stack([np.asarray(x, float)[:2 * N] if len(x) == 2 * N else np.concatenate([np.asarray(x, float)[:N], np.asarray(x, float)[len(x) // 2:len(x) // 2 + N]]) for x in xy]))
    st_ = np.ascontiguousarray(np.stack([np.asarray(q, np.int32)[:N] for q in state]))
    s2m_ = np.ascontiguousarray(np.stack([np.asarray(q, np.int32)[:N] for q in slot2map]))
    is_ = None if isStatic is None else np.ascontiguousarray(np.stack([np.asarray(q, np.uint8)[:N] for q in isStatic]))
    for a, dt in ((mapPts, np.float64), (mapCov, np.float64), (mapFlags, np.uint8), (newPt, np.uint8), (firstFrame, np.int32), (pointFeat, np.int32)):
        assert a.dtype == dt and a.flags.c_contiguous
    assert pointFeat.shape == (cap, nC)
    mc = C.c_int(int(map_count))
    match = np.full((max(nC - 1, 1), N), -1, dtype=np.int32)
    n = L.onc_new_points_from_pairs(nC, N, pp, _p(npairs), _p(K_), _p(iK_), _p(R_), _p(t_), _p(xy_), _p(st_), _p(s2m_),
                                    _p(is_) if is_ is not None else None, _p(mapPts), _p(mapCov), _p(mapFlags), _p(newPt), _p(firstFrame),
                                    _p(pointFeat), cap, C.byref(mc), int(cur_frame), C.c_double(max_disp), C.c_double(max_rp_err),
                                    C.c_double(sigma), int(min_len), int(max_seeds), int(W), int(H), _p(match))
    for c in range(nC):
        slot2map[c][:N] = s2m_[c]
    return dict(matches=match[:nC - 1], new=list(range(int(map_count), int(map_count) + n)), map_count=mc.value)
