"""NCC blocks and the epipolar / NCC matrices of CoSLAM's inter-camera matching (cs_ncc_*): NCCBlock::compute,
matchNCCBlock (reference src/slam/SL_NCCBlock.cpp:15-54, 258-264) and getEpiNccMat (src/slam/SL_FeatureMatching.cpp:3-46) --
what NewMapPtsNCC::matchBetween (src/app/SL_NewMapPointsInterCam.cpp:273-317) computes before its greedy matcher."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def ncc_blocks_dev(stream_ptr, d_img, W, H, n, d_x, d_y, scale, d_blocks, d_abc, d_valid, device=0):
    vp = C.c_void_p
    check(lib().cs_ncc_blocks_dev(int(device), vp(stream_ptr), vp(d_img), int(W), int(H), int(n), vp(d_x), vp(d_y),
                                  C.c_double(scale), vp(d_blocks), vp(d_abc), vp(d_valid)), "cs_ncc_blocks_dev")


def ncc_epi_mat_dev(stream_ptr, F, M, d_x1, d_y1, d_blocks1, d_abc1, d_valid1, N, d_x2, d_y2, d_blocks2, d_abc2, d_valid2, epiMax,
                    nccMin, wNone, d_epiMat, d_nccMat, device=0):
    vp = C.c_void_p
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
    check(lib().cs_ncc_epi_mat_dev(int(device), vp(stream_ptr), vp(F.ctypes.data), int(M), vp(d_x1), vp(d_y1), vp(d_blocks1),
                                   vp(d_abc1), vp(d_valid1), int(N), vp(d_x2), vp(d_y2), vp(d_blocks2), vp(d_abc2), vp(d_valid2),
                                   C.c_double(epiMax), C.c_double(nccMin), C.c_double(wNone), vp(d_epiMat), vp(d_nccMat)),
          "cs_ncc_epi_mat_dev")


NCC_PAIR_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("epi", "<f8"), ("ncc", "<f8")])   # == cs_ncc_pair


def ncc_epi_pairs_dev(stream_ptr, F, M, d_x1, d_y1, d_blocks1, d_abc1, d_valid1, N, d_x2, d_y2, d_blocks2, d_abc2, d_valid2, epiMax,
                      nccMin, d_pairs, pairCap, d_pairCount, device=0):
    """cs_ncc_epi_pairs_dev: only the pairs that pass both tests, as NCC_PAIR_DTYPE records (24 bytes each) + their count."""
    vp = C.c_void_p
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
    check(lib().cs_ncc_epi_pairs_dev(int(device), vp(stream_ptr), vp(F.ctypes.data), int(M), vp(d_x1), vp(d_y1), vp(d_blocks1),
                                     vp(d_abc1), vp(d_valid1), int(N), vp(d_x2), vp(d_y2), vp(d_blocks2), vp(d_abc2), vp(d_valid2),
                                     C.c_double(epiMax), C.c_double(nccMin), vp(d_pairs), int(pairCap), vp(d_pairCount)),
          "cs_ncc_epi_pairs_dev")


class NccCam(C.Structure):
    """== cs_ncc_cam (include/coslam_hip.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("img", "x", "y", "scaled", "blocks", "abc", "valid")]


class NccPairJob(C.Structure):
    """== cs_ncc_pair_job."""

    _fields_ = [("F", C.c_double * 9), ("camA", C.c_int), ("camB", C.c_int), ("pairs", C.c_void_p), ("count", C.c_void_p), ("dF", C.c_void_p)]


def ncc_cams(cams):
    """list of dicts of DEVICE pointers (ints) with the field names of cs_ncc_cam -> the ctypes array"""
    arr = (NccCam * len(cams))()
    for a, c in zip(arr, cams):
        for n, _ in NccCam._fields_:
            v = c.get(n)
            setattr(a, n, int(v) if v else None)
    return arr


def ncc_pair_jobs(jobs):
    """list of dicts(F (9 floats), camA, camB, pairs, count) -> the ctypes array"""
    arr = (NccPairJob * len(jobs))()
    for a, q in zip(arr, jobs):
        if q.get("F") is not None:
            F = np.ascontiguousarray(q["F"], dtype=np.float64).reshape(9)
            for k in range(9):
                a.F[k] = float(F[k])
        a.camA, a.camB, a.pairs, a.count = int(q["camA"]), int(q["camB"]), int(q["pairs"]), int(q["count"])
        if q.get("dF"):   # the matrix in device memory (ncc_fmats_dev), used instead of F
            a.dF = int(q["dF"])
    return arr


def ncc_fmats_dev(stream_ptr, nCams, camA, camB, d_iK, d_R, d_t, d_F, device=0):
    """cs_ncc_fmats_dev: the fundamental matrices of the camera pairs (camA[k], camB[k]) from the cameras' CURRENT poses, on the device
    (NewMapPtsNCC::matchBetween forms them from the poses the frame has just solved); d_iK: one device pointer per camera"""
    n = len(camA)
    ca, cb = (C.c_int * n)(*[int(v) for v in camA]), (C.c_int * n)(*[int(v) for v in camB])
    ik = (C.c_void_p * nCams)(*[int(v) for v in d_iK])
    check(lib().cs_ncc_fmats_dev(int(device), C.c_void_p(stream_ptr), int(nCams), n, ca, cb, ik, C.c_void_p(d_R), C.c_void_p(d_t), C.c_void_p(d_F)),
          "cs_ncc_fmats_dev")


def ncc_get_blocks_group_dev(stream_ptr, cams, W, H, n, scale, device=0):
    """getNCCBlocks of every camera of a group: one resize launch + one cutter launch (cams: ncc_cams(...) or a list of dicts)"""
    arr = cams if isinstance(cams, C.Array) else ncc_cams(cams)
    check(lib().cs_ncc_get_blocks_group_dev(int(device), C.c_void_p(stream_ptr), len(arr), arr, int(W), int(H), int(n), C.c_double(scale)),
          "cs_ncc_get_blocks_group_dev")


def ncc_epi_pairs_group_dev(stream_ptr, cams, n, jobs, epiMax, nccMin, pairCap, device=0):
    """cs_ncc_epi_pairs_dev of every camera pair of a matching run in one launch (jobs: ncc_pair_jobs(...) or a list of dicts)"""
    arr = cams if isinstance(cams, C.Array) else ncc_cams(cams)
    jb = jobs if isinstance(jobs, C.Array) else ncc_pair_jobs(jobs)
    check(lib().cs_ncc_epi_pairs_group_dev(int(device), C.c_void_p(stream_ptr), len(arr), arr, int(n), len(jb), jb, C.c_double(epiMax),
                                           C.c_double(nccMin), int(pairCap)), "cs_ncc_epi_pairs_group_dev")


def ncc_match_between(img1, x1, y1, img2, x2, y2, scale, F, epiMax, nccMin, wNone=-1.0, device=0):
    """Host arrays in and out (cs_ncc_match_between).  Returns dict(epi, ncc (M x N), blocks1/2 (n x 128 uint8), abc1/2
    (n x 4), valid1/2)."""
    img1 = np.ascontiguousarray(img1, dtype=np.uint8)
    img2 = np.ascontiguousarray(img2, dtype=np.uint8)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (x1, y1, x2, y2)]
    M, N = len(a[0]), len(a[2])
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
    epi, ncc = np.zeros((M, N)), np.zeros((M, N))
    b1, b2 = np.zeros((M, 128), np.uint8), np.zeros((N, 128), np.uint8)
    c1, c2 = np.zeros((M, 4)), np.zeros((N, 4))
    v1, v2 = np.zeros(M, np.int32), np.zeros(N, np.int32)
    p = lambda v: C.c_void_p(v.ctypes.data)  # noqa: E731
    check(lib().cs_ncc_match_between(int(device), p(img1), img1.shape[1], img1.shape[0], M, p(a[0]), p(a[1]), p(img2),
                                     img2.shape[1], img2.shape[0], N, p(a[2]), p(a[3]), C.c_double(scale), p(F),
                                     C.c_double(epiMax), C.c_double(nccMin), C.c_double(wNone), p(epi), p(ncc), p(b1), p(c1),
                                     p(v1), p(b2), p(c2), p(v2)), "cs_ncc_match_between")
    return dict(epi=epi, ncc=ncc, blocks1=b1, abc1=c1, valid1=v1, blocks2=b2, abc2=c2, valid2=v2)


def ncc_get_blocks_dev(stream_ptr, d_img, W, H, n, d_x, d_y, scale, d_scaled, d_blocks, d_abc, d_valid=0, device=0):
    """cs_ncc_get_blocks_dev: getNCCBlocks (reference src/slam/SL_NCCBlock.cpp:79-155) for n points, device pointers"""
    vp = C.c_void_p
    check(lib().cs_ncc_get_blocks_dev(int(device), vp(stream_ptr), vp(d_img), int(W), int(H), int(n), vp(d_x), vp(d_y),
                                      C.c_double(scale), vp(d_scaled), vp(d_blocks), vp(d_abc), vp(d_valid)), "cs_ncc_get_blocks_dev")


def ncc_scaled_dims(W, H, scale):
    ws, hs = C.c_int(0), C.c_int(0)
    check(lib().cs_ncc_scaled_dims(int(W), int(H), C.c_double(scale), C.byref(ws), C.byref(hs)), "cs_ncc_scaled_dims")
    return ws.value, hs.value


def ncc_match_between_full(img1, x1, y1, img2, x2, y2, scale, F, epiMax, nccMin, wNone=-1.0, device=0):
    """NewMapPtsNCC::matchBetween's own data path (cs_ncc_match_between_full): FULL images, getNCCBlocks with blockScale, the
    two matrices.  Returns dict(epi, ncc (M x N), blocks1/2 (n x 128), abc1/2 (n x 4))."""
    img1 = np.ascontiguousarray(img1, dtype=np.uint8)
    img2 = np.ascontiguousarray(img2, dtype=np.uint8)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (x1, y1, x2, y2)]
    M, N = len(a[0]), len(a[2])
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
    epi, ncc = np.zeros((M, N)), np.zeros((M, N))
    b1, b2 = np.zeros((M, 128), np.uint8), np.zeros((N, 128), np.uint8)
    c1, c2 = np.zeros((M, 4)), np.zeros((N, 4))
    p = lambda v: C.c_void_p(v.ctypes.data)  # noqa: E731
    check(lib().cs_ncc_match_between_full(int(device), p(img1), img1.shape[1], img1.shape[0], M, p(a[0]), p(a[1]), p(img2),
                                          img2.shape[1], img2.shape[0], N, p(a[2]), p(a[3]), C.c_double(scale), p(F),
                                          C.c_double(epiMax), C.c_double(nccMin), C.c_double(wNone), p(epi), p(ncc), p(b1), p(c1),
                                          p(b2), p(c2)), "cs_ncc_match_between_full")
    return dict(epi=epi, ncc=ncc, blocks1=b1, abc1=c1, blocks2=b2, abc2=c2)
