"""ctypes mirror of cs_keyframe_ready_dev (include/coslam_hip.h): CoSLAM::IsReadyForKeyFrame for every camera in one launch, the summary
of genNewMapPoints' camera loop and, on request, addKeyFrame's effect on the cameras' key-pose state (reference
src/app/SL_CoSLAM.cpp:1269-1309, :1280-1293)."""
import ctypes as C

from ._lib import check, lib


class KeyframeCam(C.Structure):
    _fields_ = [("state", C.c_void_p), ("slot2map", C.c_void_p), ("R", C.c_void_p), ("t", C.c_void_p), ("keyFrame", C.c_void_p),
                ("keyMapped", C.c_void_p), ("selfR", C.c_void_p), ("selfT", C.c_void_p)]


def keyframe_cams(cams):
    """cams: dicts with the cs_keyframe_cam fields as device pointers"""
    arr = (KeyframeCam * len(cams))()
    for i, c in enumerate(cams):
        for k, _ in KeyframeCam._fields_:
            setattr(arr[i], k, c[k])
    return arr


def keyframe_ready_dev(stream_ptr, cams, N, nMap, d_mapPts, d_mapFlags, d_firstFrame, curFrame, minTranslation, d_ready, d_mapped, d_center,
                       ratio=0.93, minViewAngleDeg=5.0, addKeyFrame=False, d_stats=None, device=0):
    """cs_keyframe_ready_dev; cams: a keyframe_cams() array or a list of dicts"""
    arr = cams if not isinstance(cams, (list, tuple)) else keyframe_cams(cams)
    vp = C.c_void_p
    check(lib().cs_keyframe_ready_dev(int(device), vp(stream_ptr), len(arr), int(N), arr, int(nMap), vp(d_mapPts), vp(d_mapFlags), vp(d_firstFrame),
                                      int(curFrame), C.c_double(ratio), C.c_double(minViewAngleDeg), C.c_double(minTranslation),
                                      int(bool(addKeyFrame)), vp(d_ready), vp(d_mapped), vp(d_center), vp(d_stats)), "cs_keyframe_ready_dev")
