"""Search step of CoSLAM's map-point registration (cs_register_search*): for every (map point, camera) pair what
CoSLAM::curStaticPointRegInGroup / curDynamicPointRegInGroup / activeMapPointRegisterInGroup (reference
src/app/SL_CoSLAM.cpp:731-757, 955-980, 1118-1145) compute before they decide -- projection, projected covariance, the
Mahalanobis-nearest feature of the camera's current frame (searchMahaNearestFeatPt, src/app/SL_SingleSLAM.cpp:1141-1164) --
for all points and cameras in one launch."""
import ctypes as C

import numpy as np

from ._lib import check, lib

SKIP_HAS_FEATURE, SKIP_BEHIND, SKIP_OUTSIDE, SKIP_NO_FEATURES = -1, -2, -3, -4
FLAG_UNMAPPED, FLAG_DYNAMIC, FLAG_MERGEABLE = 1, 2, 4


class RegisterCam(C.Structure):
    """== cs_register_cam (include/coslam_hip.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("K", "R", "t", "xy", "state", "slot2map", "isDynamic", "isStatic")]


def register_cams(cams):
    """list of dicts of pointers (ints) with the field names of cs_register_cam -> the ctypes array (build once)"""
    if isinstance(cams, C.Array):
        return cams
    arr = (RegisterCam * len(cams))()
    for a, c in zip(arr, cams):
        for n, _ in RegisterCam._fields_:
            v = c.get(n)
            setattr(a, n, int(v) if v else None)
    return arr


def register_search_dev(stream_ptr, cams, N, W, H, P, d_M, d_cov, d_pointFeat, sigmaSearch, maxDist, sigmaMerge, d_slot, d_m,
                        d_var, d_dist, d_flags, device=0):
    """cams: list of dicts of DEVICE pointers (ints) with the field names of cs_register_cam; outputs P x nCams tables."""
    vp = C.c_void_p
    check(lib().cs_register_search_dev(int(device), vp(stream_ptr), len(cams), register_cams(cams), int(N), int(W), int(H), int(P),
                                       vp(d_M), vp(d_cov), vp(d_pointFeat), C.c_double(sigmaSearch), C.c_double(maxDist),
                                       C.c_double(sigmaMerge), vp(d_slot), vp(d_m), vp(d_var), vp(d_dist), vp(d_flags)),
          "cs_register_search_dev")


class RegisterPass(C.Structure):
    """== cs_register_pass (include/coslam_hip.h)."""

    _fields_ = [("P", C.c_int), ("sigmaSearch", C.c_double), ("maxDist", C.c_double), ("sigmaMerge", C.c_double)] + \
               [(n, C.c_void_p) for n in ("M", "cov", "pointFeat", "slot", "m", "var", "dist", "flags", "mapFlags")] + [("maxDistDynamic", C.c_double)] + \
               [("list", C.c_void_p)]


def register_passes(passes):
    """list of dicts with the field names of cs_register_pass (pointers as ints) -> the ctypes array (build once)"""
    if isinstance(passes, C.Array):
        return passes
    arr = (RegisterPass * len(passes))()
    for a, q in zip(arr, passes):
        a.P, a.sigmaSearch, a.maxDist, a.sigmaMerge = int(q["P"]), float(q["sigmaSearch"]), float(q["maxDist"]), float(q["sigmaMerge"])
        for n in ("M", "cov", "pointFeat", "slot", "m", "var", "dist", "flags"):
            setattr(a, n, int(q[n]))
        if q.get("mapFlags"):   # optional: the certainly dynamic points of the pass are searched with their own scale
            a.mapFlags, a.maxDistDynamic = int(q["mapFlags"]), float(q["maxDistDynamic"])
        if q.get("list"):       # optional: the pass's points as a list of P map indices; the tables are then whole-map tables
            a.list = int(q["list"])
    return arr


def register_list_current_dev(stream_ptr, nCams, nMap, d_mapCount, d_pointFeat, d_mapFlags, d_list, d_listCount=0, d_slotTable=0, device=0,
                              listCap=None, d_overflow=0):
    """cs_register_list_current_cap_dev: the frame's current map points (a feature of this frame in some camera, below the live count, not
    false) as a compact list in map order -- what CoSLAM::currentMapPointsRegister walks (curMapPts).  listCap: at most that many are
    listed; the rest lose their candidate rows for this frame and are counted into d_overflow."""
    vp = C.c_void_p
    check(lib().cs_register_list_current_cap_dev(int(device), vp(stream_ptr), int(nCams), int(nMap), vp(d_mapCount), vp(d_pointFeat), vp(d_mapFlags),
                                                 vp(d_list), vp(d_listCount), vp(d_slotTable), int(nMap if listCap is None else listCap), vp(d_overflow)),
          "cs_register_list_current_cap_dev")


def register_search_passes_dev(stream_ptr, cams, N, W, H, passes, device=0, cam0=0, nCamsRun=None):
    """The registration passes of a frame (1 or 2) in ONE launch; cams / passes: lists of dicts or the prebuilt ctypes arrays.
    cam0 / nCamsRun: only these cameras' columns of the len(cams)-wide tables (cs_register_search_passes_range_dev)."""
    arr = register_passes(passes)
    ca = register_cams(cams)
    check(lib().cs_register_search_passes_range_dev(int(device), C.c_void_p(stream_ptr), len(ca), int(cam0),
                                                    int(len(ca) - cam0 if nCamsRun is None else nCamsRun), ca, int(N),
                                                    int(W), int(H), len(arr), arr), "cs_register_search_passes_range_dev")


def register_search(W, H, Ks, Rs, ts, xy, state, slot2map, isDynamic, Ms, covs, pointFeat, sigmaSearch, maxDist, sigmaMerge,
                    device=0):
    """Host arrays in and out (cs_register_search).  xy / state / slot2map / isDynamic: one array per camera (isDynamic
    entries may be None).  Returns dict(slot, m, var, dist, flags), P x nCams each."""
    nC = len(xy)
    N = len(state[0])
    Ks = np.ascontiguousarray(Ks, dtype=np.float64).reshape(nC, 9)
    Rs = np.ascontiguousarray(Rs, dtype=np.float64).reshape(nC, 9)
    ts = np.ascontiguousarray(ts, dtype=np.float64).reshape(nC, 3)
    Ms = np.ascontiguousarray(Ms, dtype=np.float64).reshape(-1, 3)
    P = len(Ms)
    covs = np.ascontiguousarray(covs, dtype=np.float64).reshape(P, 9)
    pf = np.ascontiguousarray(pointFeat, dtype=np.int32).reshape(P, nC)
    keep = []
    cams = []
    for c in range(nC):
        x = np.ascontiguousarray(xy[c], dtype=np.float64)
        s = np.ascontiguousarray(state[c], dtype=np.int32)
        m2 = np.ascontiguousarray(slot2map[c], dtype=np.int32)
        dy = None if isDynamic is None or isDynamic[c] is None else np.ascontiguousarray(isDynamic[c], dtype=np.uint8)
        assert x.size == 2 * N and s.size == N and m2.size == N
        keep += [x, s, m2, dy]
        cams.append(dict(K=Ks[c].ctypes.data, R=Rs[c].ctypes.data, t=ts[c].ctypes.data, xy=x.ctypes.data, state=s.ctypes.data,
                         slot2map=m2.ctypes.data, isDynamic=None if dy is None else dy.ctypes.data))
    slot = np.zeros((P, nC), dtype=np.int32)
    m = np.zeros((P, nC, 2))
    var = np.zeros((P, nC, 4))
    dist = np.zeros((P, nC))
    flags = np.zeros((P, nC), dtype=np.int32)
    vp = C.c_void_p
    check(lib().cs_register_search(int(device), nC, register_cams(cams), int(N), int(W), int(H), int(P), vp(Ms.ctypes.data),
                                   vp(covs.ctypes.data), vp(pf.ctypes.data), C.c_double(sigmaSearch), C.c_double(maxDist),
                                   C.c_double(sigmaMerge), vp(slot.ctypes.data), vp(m.ctypes.data), vp(var.ctypes.data),
                                   vp(dist.ctypes.data), vp(flags.ctypes.data)), "cs_register_search")
    return dict(slot=slot, m=m, var=var, dist=dist, flags=flags)


def register_decide_scratch_bytes(nCams, N, P):
    L = lib()
    L.cs_register_decide_scratch_bytes.restype = C.c_size_t
    return int(L.cs_register_decide_scratch_bytes(int(nCams), int(N), int(P)))


def register_decide_static_dev(stream_ptr, nCams, N, P, mapBase, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, d_slot2map, d_attached,
                               d_regged, d_scratch, d_counts=0, device=0, n_sweeps=3, only_cam=-1, kinds=1):
    """cs_register_decide_kinds_dev (only_cam >= 0: ONE camera's loop of the reference; kinds 1: the certainly static points, 2: the
    certainly dynamic ones, 3: both): d_slot2map = list of nCams device pointers (or the prebuilt c_void_p array)"""
    vp = C.c_void_p
    arr = d_slot2map if isinstance(d_slot2map, C.Array) else (C.c_void_p * nCams)(*[int(x) for x in d_slot2map])
    check(lib().cs_register_decide_kinds_dev(int(device), vp(stream_ptr), int(nCams), int(N), int(P), int(mapBase), vp(d_slot), vp(d_flags),
                                             vp(d_mergeable), vp(d_mapFlags), vp(d_pointFeat), arr, vp(d_attached), vp(d_regged),
                                             vp(d_scratch), int(n_sweeps), vp(d_counts), int(only_cam), int(kinds)), "cs_register_decide_kinds_dev")
    return arr


def register_revisit_list_dev(stream_ptr, nCams, P, cap, first_round, d_pointFeat, d_attached, d_regIn, keep_in, d_visitLoop, d_nextLoop, d_list,
                              d_counts=0, device=0, d_regOutClear=0):
    """cs_register_revisit_list_dev: the points that registered in the previous round and are visited again in a later camera's loop"""
    vp = C.c_void_p
    check(lib().cs_register_revisit_list_dev(int(device), vp(stream_ptr), int(nCams), int(P), int(cap), int(bool(first_round)), vp(d_pointFeat),
                                             vp(d_attached), vp(d_regIn), int(bool(keep_in)), vp(d_regOutClear), vp(d_visitLoop), vp(d_nextLoop), vp(d_list), vp(d_counts)),
          "cs_register_revisit_list_dev")


def register_revisit_decide_dev(stream_ptr, nCams, N, P, cap, mapBase, kinds, d_list, d_nextLoop, d_visitLoop, d_slot, d_flags, d_mergeable, d_mapFlags,
                                d_pointFeat, d_slot2map, d_attached, d_regOut, d_decideScratch, d_curList, d_curCount, curCap, d_counts=0, device=0, d_listCount=0):
    """cs_register_revisit_decide_dev: the listed points' walks in their next loop (see include/coslam_hip.h)"""
    vp = C.c_void_p
    arr = d_slot2map if isinstance(d_slot2map, C.Array) else (C.c_void_p * nCams)(*[int(x) for x in d_slot2map])
    check(lib().cs_register_revisit_decide_dev(int(device), vp(stream_ptr), int(nCams), int(N), int(P), int(cap), int(mapBase), int(kinds), vp(d_list),
                                               vp(d_nextLoop), vp(d_visitLoop), vp(d_slot), vp(d_flags), vp(d_mergeable), vp(d_mapFlags),
                                               vp(d_pointFeat), arr, vp(d_attached), vp(d_regOut), vp(d_decideScratch), vp(d_curList), vp(d_curCount),
                                               int(curCap), vp(d_counts), vp(d_listCount)), "cs_register_revisit_decide_dev")
    return arr


def register_decide_kinds_rounds_dev(stream_ptr, nCams, N, P, mapBase, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, d_slot2map, d_attached,
                                     d_regged, d_scratch, d_rvLists, rvCap, nRounds, d_rvCounts, d_visitLoop, d_nextLoop, d_counts=0, device=0, kinds=3):
    """cs_register_decide_kinds_rounds_dev: the single pass (self-settling launch, all cameras' loops) whose walks build the second visits'
    first list themselves -- d_rvLists [nRounds][rvCap] (cleared here), d_rvCounts [nRounds + 1] (the last entry, points beyond the lists, is
    not cleared), d_visitLoop / d_nextLoop [P]"""
    vp = C.c_void_p
    arr = d_slot2map if isinstance(d_slot2map, C.Array) else (C.c_void_p * nCams)(*[int(x) for x in d_slot2map])
    check(lib().cs_register_decide_kinds_rounds_dev(int(device), vp(stream_ptr), int(nCams), int(N), int(P), int(mapBase), vp(d_slot), vp(d_flags),
                                                    vp(d_mergeable), vp(d_mapFlags), vp(d_pointFeat), arr, vp(d_attached), vp(d_regged), vp(d_scratch), 0,
                                                    vp(d_counts), -1, int(kinds), vp(d_rvLists), int(rvCap), int(nRounds), vp(d_rvCounts), vp(d_visitLoop),
                                                    vp(d_nextLoop)), "cs_register_decide_kinds_rounds_dev")
    return arr


def register_revisit_decide_next_dev(stream_ptr, nCams, N, P, cap, mapBase, kinds, d_list, d_nextLoop, d_visitLoop, d_slot, d_flags, d_mergeable, d_mapFlags,
                                     d_pointFeat, d_slot2map, d_attached, d_regOut, d_decideScratch, d_curList, d_curCount, curCap, d_counts=0, device=0,
                                     d_listCount=0, d_nextList=0, d_nextCount=0, d_overflow=0):
    """cs_register_revisit_decide_next_dev: cs_register_revisit_decide_dev whose walks append the points that registered again to the next
    round's list (d_nextList / d_nextCount: both or neither)"""
    vp = C.c_void_p
    arr = d_slot2map if isinstance(d_slot2map, C.Array) else (C.c_void_p * nCams)(*[int(x) for x in d_slot2map])
    check(lib().cs_register_revisit_decide_next_dev(int(device), vp(stream_ptr), int(nCams), int(N), int(P), int(cap), int(mapBase), int(kinds), vp(d_list),
                                                    vp(d_nextLoop), vp(d_visitLoop), vp(d_slot), vp(d_flags), vp(d_mergeable), vp(d_mapFlags),
                                                    vp(d_pointFeat), arr, vp(d_attached), vp(d_regOut), vp(d_decideScratch), vp(d_curList), vp(d_curCount),
                                                    int(curCap), vp(d_counts), vp(d_listCount), vp(d_nextList), vp(d_nextCount), vp(d_overflow)),
          "cs_register_revisit_decide_next_dev")
    return arr


def register_cur_static_sequential_dev(stream_ptr, history, pu_cams, reg_cams, N, W, H, search_pass, P, d_slot, d_flags, d_mergeable, d_mapFlags,
                                       d_pointFeat, d_slot2map, d_attached, d_regged, d_scratch, d_mapPts, d_mapCov, pixelVar, d_counts=0,
                                       after_loop=None, device=0, n_sweeps=6, with_dynamic=False, merge=False, d_merge_scratch=0, mergability=None,
                                       d_featRef=0, d_refStatic=0, curFrame=None):
    """CoSLAM::curStaticPointsRegInGroup (bMerge == false) AS THE REFERENCE RUNS IT (src/app/SL_CoSLAM.cpp:854-898), camera loop after
    camera loop, on the device: for o = 0 .. nCams - 1 -- the search from the points as they stand (cs_register_search_passes_dev with
    the ONE pass `search_pass`, whose tables are d_slot / d_flags), staticCheckMergability of its candidates (history: a TrackHistory),
    the walks of the points that hold a feature in camera o (cs_register_decide_static_cam_dev), refineMapPoint of those that gained one
    (:889-893).  nCams times the frame loop's launches: the parity mode.  after_loop(o): called behind every camera's
    loop (e.g. to read d_counts -- features attached, points registered, sweeps, settled -- of that loop: the reference's return value is
    the registrations summed over the loops)."""
    nC = len(reg_cams)
    arr = None
    # with_dynamic: curDynamicPointsRegInGroup's loops behind the static ones (currentMapPointsRegister, :834-853); search_pass must then
    # carry mapFlags / maxDistDynamic (the dynamic points' own scale).  after_loop(o) is called with o = nCams .. 2 nCams - 1 for them.
    # merge: bMerge == true (every 50th frame) -- the static loops through cs_register_decide_merge_dev (pu_cams' slot2map = d_slot2map's
    # arrays; d_merge_scratch: P bytes; d_mapFlags is then written)
    for kind, o in [(1, o_) for o_ in range(nC)] + ([(2, o_) for o_ in range(nC)] if with_dynamic else []):
        register_search_passes_dev(stream_ptr, reg_cams, N, W, H, search_pass, device=device)
        if mergability is not None:   # (the caller's own mergability launch: e.g. the running whole-track verdict over a list of rows)
            mergability(stream_ptr)
        else:
            history.register_mergability_dev(stream_ptr, pu_cams, P, d_mapPts, d_mapCov, d_slot, pixelVar, d_mergeable)
        if merge and kind == 1:   # bMerge: the static points' walks one after the other, checkUnify at a conflict (the dynamic loops ignore bMerge);
            # d_counts then reads: features attached, points registered, points unified away, checkUnify calls
            history.register_decide_merge_dev(stream_ptr, pu_cams, P, 0, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, d_mapPts, d_mapCov,
                                              pixelVar, d_attached, d_regged, d_merge_scratch, d_counts, only_cam=o)
        else:
            arr = register_decide_static_dev(stream_ptr, nC, N, P, 0, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, arr or d_slot2map,
                                             d_attached, d_regged, d_scratch, d_counts, device=device, n_sweeps=n_sweeps, only_cam=o, kinds=kind)
        if d_featRef:   # MapPoint::pFeatures as references: brought up to the loop's attachments (re-links, hand-overs), then refineMapPoint over them
            history.feat_ref_advance_dev(stream_ptr, pu_cams, P, d_pointFeat, curFrame, d_featRef, d_refStatic=d_refStatic or None)
            history.refine_map_points_ref_dev(stream_ptr, pu_cams, d_featRef, P, d_mapPts, d_mapCov, pixelVar, d_select=d_regged)
        else:
            history.refine_map_points_dev(stream_ptr, pu_cams, d_pointFeat, P, d_mapPts, d_mapCov, pixelVar, d_select=d_regged)
        if after_loop is not None:
            after_loop(o + (nC if kind == 2 else 0))
    return arr
