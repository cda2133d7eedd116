"""On-device hand-back of the tracker's dest[] to the pose stage (cs_klt_handback_dev): what GPUKLT::addToFeaturePoints
(reference src/tracking/GPUKLT.cpp:36-60), SingleSLAM::chooseStaticFeatPts (src/app/SL_SingleSLAM.cpp:345-397) and the
Ms / ms packing of SingleSLAM::poseUpdate3D (:620-640) do on the host with pointer lists, for all cameras in one launch."""
import ctypes as C

from ._lib import check, lib
from .pose import IntraCamPoseOption  # noqa: F401  (layout of the `opt` record)


class HandbackCam(C.Structure):
    """== cs_handback_cam (include/coslam_hip.h): device pointers of one camera."""

    _fields_ = [(n, C.c_void_p) for n in ("dest", "K", "kud", "mapPts", "isStatic", "slot2map", "trackSpan", "xy", "state",
                                          "selBlk", "Ms", "ms", "sel", "npts", "opt", "pointFeat")] + \
               [("pointFeatStride", C.c_int), ("nPointFeat", C.c_int)]


def handback_cams(cams):
    """list of dicts of device pointers (ints; missing / None = NULL) with the field names of cs_handback_cam -> the ctypes
    array cs_klt_handback_dev takes (build it once when the buffers do not change from frame to frame)"""
    arr = (HandbackCam * len(cams))()
    for a, c in zip(arr, cams):
        for n, ty in HandbackCam._fields_:
            v = c.get(n)
            setattr(a, n, int(v or 0) if ty is C.c_int else (int(v) if v else None))
    return arr


def handback_dev(stream_ptr, cams, N, W, H, nColBlk=16, nRowBlk=12, ptsStride=192, device=0, frame=0):
    """cams: a list of dicts (see handback_cams) or the array handback_cams returned;
    frame: GPUKLT::m_frame of this call (the tracks' frame spans are kept in trackSpan)."""
    arr = cams if isinstance(cams, C.Array) else handback_cams(cams)
    check(lib().cs_klt_handback_dev(int(device), C.c_void_p(stream_ptr), len(arr), arr, int(N), int(W), int(H),
                                    int(nColBlk), int(nRowBlk), int(ptsStride), int(frame)), "cs_klt_handback_dev")
