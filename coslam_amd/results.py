"""Result text files of a run (cs_export_results_v1): what CoSLAM::exportResultsVer1 writes (reference
src/app/SL_CoSLAM.cpp:1914-2028) -- input_videos.txt, mappts.txt, <c>_campose.txt, <c>_featpts.txt -- from arrays.  Host code
only; no GPU is needed to call it."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class ExportCam(C.Structure):
    """== cs_export_cam (include/coslam_hip.h)."""

    _fields_ = [("videoFilePath", C.c_char_p), ("K", C.c_void_p), ("kc", C.c_void_p), ("W", C.c_int), ("H", C.c_int),
                ("startFrameInVideo", C.c_int), ("nPoses", C.c_int), ("poseFrame", C.c_void_p), ("poseR", C.c_void_p),
                ("poseT", C.c_void_p), ("featPtr", C.c_void_p), ("featPointId", C.c_void_p), ("featXY", C.c_void_p)]


def export_results_v1(dir_path, cams, cur_frame, pt_id, pt_M, pt_cov, cov_as_reference=True):
    """cams: list of dicts(videoFilePath, K[9], kc[5], W, H, startFrameInVideo, poseFrame[n], poseR[n,9], poseT[n,3],
    featPtr[curFrame - poseFrame[0] + 2], featPointId[m], featXY[m,2]); pt_id / pt_M / pt_cov: the static map points in the
    order to list them.  cov_as_reference: reproduce the reference's 9 numbers per point (see include/coslam_hip.h)."""
    keep, arr = [], (ExportCam * len(cams))()
    for a, c in zip(arr, cams):
        v = dict(K=np.ascontiguousarray(c["K"], np.float64).reshape(9), kc=np.ascontiguousarray(c["kc"], np.float64).reshape(5),
                 poseFrame=np.ascontiguousarray(c["poseFrame"], np.int32), poseR=np.ascontiguousarray(c["poseR"], np.float64),
                 poseT=np.ascontiguousarray(c["poseT"], np.float64), featPtr=np.ascontiguousarray(c["featPtr"], np.int32),
                 featPointId=np.ascontiguousarray(c["featPointId"], np.int64), featXY=np.ascontiguousarray(c["featXY"], np.float64))
        assert v["poseR"].size == 9 * len(v["poseFrame"]) and v["poseT"].size == 3 * len(v["poseFrame"])
        assert len(v["featPtr"]) == int(cur_frame) - int(v["poseFrame"][0]) + 2 and v["featXY"].size == 2 * len(v["featPointId"])
        keep.append(v)
        path = c["videoFilePath"]
        a.videoFilePath = path if isinstance(path, bytes) else str(path).encode()
        a.W, a.H, a.startFrameInVideo, a.nPoses = int(c["W"]), int(c["H"]), int(c["startFrameInVideo"]), len(v["poseFrame"])
        for n in ("K", "kc", "poseFrame", "poseR", "poseT", "featPtr", "featPointId", "featXY"):
            setattr(a, n, v[n].ctypes.data)
    pt_id = np.ascontiguousarray(pt_id, np.int64)
    pt_M = np.ascontiguousarray(pt_M, np.float64).reshape(-1, 3)
    pt_cov = np.ascontiguousarray(pt_cov, np.float64).reshape(-1, 9)
    assert len(pt_M) == len(pt_id) == len(pt_cov)
    p = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    path = dir_path if isinstance(dir_path, bytes) else str(dir_path).encode()
    check(lib().cs_export_results_v1(path, len(cams), arr, int(cur_frame), len(pt_id), p(pt_id), p(pt_M), p(pt_cov),
                                     1 if cov_as_reference else 0), "cs_export_results_v1")
