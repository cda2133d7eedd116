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


def keyframe_snapshot_dev(stream_ptr, nCams, N, d_xy, d_state, d_slot2map, d_R, d_t, d_word, d_xyOut, d_stateOut, d_slot2mapOut, d_ROut, d_tOut, h_word,
                          device=0):
    """cs_keyframe_snapshot_dev: a frame's records, poses and decision word (into pinned host memory h_word) into a slot of a ring, one launch"""
    vp = C.c_void_p
    check(lib().cs_keyframe_snapshot_dev(int(device), vp(stream_ptr), int(nCams), int(N), vp(d_xy), vp(d_state), vp(d_slot2map), vp(d_R), vp(d_t), vp(d_word),
                                         vp(d_xyOut), vp(d_stateOut), vp(d_slot2mapOut), vp(d_ROut), vp(d_tOut), vp(h_word)), "cs_keyframe_snapshot_dev")
