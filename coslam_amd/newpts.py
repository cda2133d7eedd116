"""New map points from the inter-camera NCC candidates (cs_newpts_from_pairs_dev): what NewMapPtsNCC::run / output do behind
getEpiNccMat (reference src/app/SL_NewMapPointsInterCam.cpp:150-192, 194-270, 295-316, 631-690) -- greedy guided matches per
camera pair, the matches chained into tracks, every track triangulated, gated and appended to the map, on the device."""
import ctypes as C

from ._lib import check, lib
from .poseupdate import poseupdate_cams


def newpts_scratch_bytes(nCams, N):
    L = lib()
    L.cs_newpts_scratch_bytes.restype = C.c_size_t
    return int(L.cs_newpts_scratch_bytes(int(nCams), int(N)))


def ncc_candidate_mask_dev(stream_ptr, nCams, N, d_state, d_slot2map, d_trackSpan, d_mapFlags, mapCap, d_valid, minTrack=3, device=0,
                           validStride=0):
    """validStride: ints from one camera's mask to the next (0: N, back to back)"""
    vp = C.c_void_p
    check(lib().cs_ncc_candidate_mask_dev(int(device), vp(stream_ptr), int(nCams), int(N), vp(d_state), vp(d_slot2map), vp(d_trackSpan),
                                          vp(d_mapFlags), int(mapCap), int(minTrack), vp(d_valid), C.c_size_t(int(validStride))),
          "cs_ncc_candidate_mask_dev")


class NewPtsJob:
    """the argument tables of cs_newpts_from_pairs_dev, built once (device pointers as ints)"""

    def __init__(self, cams, d_pairs, d_pair_counts):
        self.cams = poseupdate_cams(cams)
        n = len(self.cams)
        assert len(d_pairs) == n - 1 and len(d_pair_counts) == n - 1
        self.pairs = (C.c_void_p * (n - 1))(*[int(p) for p in d_pairs])
        self.counts = (C.c_void_p * (n - 1))(*[int(p) for p in d_pair_counts])
        self.n = n


def newpts_from_pairs_dev(stream_ptr, job, N, pairCap, d_R, d_t, d_mapPts, d_mapCov, d_mapFlags, d_newPt, d_firstFrame, d_pointFeat, mapCap,
                          d_mapCount, curFrame, d_scratch, d_counts=0, maxDisp=80.0, maxRpErr=3.0, pixelErrVar=10.0, minLen=2, device=0, W=640, H=480):
    vp = C.c_void_p
    check(lib().cs_newpts_from_pairs_dev(int(device), vp(stream_ptr), job.n, int(N), job.cams, job.pairs, job.counts, int(pairCap), vp(d_R), vp(d_t),
                                         vp(d_mapPts), vp(d_mapCov), vp(d_mapFlags), vp(d_newPt), vp(d_firstFrame), vp(d_pointFeat), int(mapCap),
                                         vp(d_mapCount), int(curFrame), C.c_double(maxDisp), C.c_double(maxRpErr), C.c_double(pixelErrVar),
                                         int(minLen), int(W), int(H), vp(d_scratch), vp(d_counts)), "cs_newpts_from_pairs_dev")
