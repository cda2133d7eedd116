"""What a frame does with the cameras' new poses (cs_pose_update3d_dev / cs_detect_dynamic_dev / cs_pose_update_frame_dev): the
gate + seqTriangulate loop of SingleSLAM::poseUpdate3D (reference src/app/SL_SingleSLAM.cpp:672-708) and
SingleSLAM::detectDynamicFeaturePoints (:784-824) for every camera of a group, on the device."""
import ctypes as C

from ._lib import check, lib

MAP_DYNAMIC, MAP_FALSE, MAP_UNCERTAIN = 1, 2, 4


class PoseUpdateCam(C.Structure):
    """== cs_poseupdate_cam (include/coslam_hip.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("K", "iK", "xy", "state", "slot2map", "trackSpan", "reprojErr", "isStatic")]


def poseupdate_cams(cams):
    """list of dicts of DEVICE pointers (ints) with the field names of cs_poseupdate_cam -> the ctypes array (build once)"""
    if isinstance(cams, C.Array):
        return cams
    arr = (PoseUpdateCam * len(cams))()
    for a, c in zip(arr, cams):
        for n, _ in PoseUpdateCam._fields_:
            v = c.get(n)
            setattr(a, n, int(v) if v else None)
    return arr


def pose_update3d_dev(stream_ptr, cams, N, d_pointFeat, nMap, d_R, d_t, d_mapPts, d_mapCov, d_mapFlags, largeErr, pixelErrVar,
                      d_numNodes=None, d_numOut=None, cam0=0, nCamsRun=None, device=0):
    """The gate loop for cameras cam0 .. cam0 + nCamsRun - 1 (default: all), in camera order per map point."""
    vp = C.c_void_p
    arr = poseupdate_cams(cams)
    n = len(arr)
    check(lib().cs_pose_update3d_dev(int(device), vp(stream_ptr), n, int(cam0), int(n - cam0 if nCamsRun is None else nCamsRun), arr,
                                     int(N), vp(d_pointFeat), int(nMap), vp(d_R), vp(d_t), vp(d_mapPts), vp(d_mapCov),
                                     vp(d_mapFlags), int(largeErr), C.c_double(pixelErrVar), vp(d_numNodes), vp(d_numOut)),
          "cs_pose_update3d_dev")


class TrackHistory:
    """cs_track_history: the ring of the last histLen frames' hand-back pixels and poses the dynamic test walks."""

    def __init__(self, nCams, N, histLen, device=0, storeLen=0):
        """storeLen > histLen: that many frames are KEPT (cs_track_history_create_ex) while the walks stay histLen deep -- what the
        running mergability verdict rebuilds a cached tail from"""
        L = lib()
        L.cs_track_history_create_ex.restype = C.c_void_p
        self._L = L
        self.nCams, self.N, self.histLen = int(nCams), int(N), int(histLen)
        self.storeLen = max(int(storeLen), self.histLen)
        self._h = L.cs_track_history_create_ex(int(device), self.nCams, self.N, self.histLen, self.storeLen)
        if not self._h:
            check(-1, "cs_track_history_create_ex")

    def close(self):
        if self._h:
            self._L.cs_track_history_destroy(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def frames(self):
        return self._L.cs_track_history_frames(C.c_void_p(self._h))

    def detect_dynamic_dev(self, stream_ptr, cams, d_R, d_t, nMap, d_mapFlags, frame, maxLen=20, minLen=5, minOutNum=3,
                           maxEpiErr=6.0, d_numDyn=None, cam0=0, nCamsRun=None):
        """detectDynamicFeaturePoints(20, 5, 3, Const::MAX_EPI_ERR) (reference src/app/SL_CoSLAM.cpp:361, 404-405)."""
        vp = C.c_void_p
        arr = poseupdate_cams(cams)
        check(self._L.cs_detect_dynamic_dev(vp(self._h), vp(stream_ptr), int(cam0),
                                            int(self.nCams - cam0 if nCamsRun is None else nCamsRun), arr, vp(d_R), vp(d_t), int(nMap),
                                            vp(d_mapFlags), int(frame), int(maxLen), int(minLen), int(minOutNum),
                                            C.c_double(maxEpiErr), vp(d_numDyn)), "cs_detect_dynamic_dev")

    def register_mergability_dev(self, stream_ptr, cams, P, d_M, d_cov, d_slot, pixelErrVar, d_mergeable, cam0=0, nCamsRun=None):
        """CoSLAM::staticCheckMergability for every candidate of a registration search (d_slot: P x nCams), full track history;
        cam0 / nCamsRun: only these cameras' columns."""
        vp = C.c_void_p
        check(self._L.cs_register_mergability_range_dev(vp(self._h), vp(stream_ptr), int(cam0),
                                                        int(self.nCams - cam0 if nCamsRun is None else nCamsRun), poseupdate_cams(cams),
                                                        int(P), vp(d_M), vp(d_cov), vp(d_slot), C.c_double(pixelErrVar), vp(d_mergeable)),
              "cs_register_mergability_range_dev")

    def mergability_cache_bytes(self, P):
        self._L.cs_register_mergability_cache_bytes.restype = C.c_size_t
        return int(self._L.cs_register_mergability_cache_bytes(int(P), self.nCams))

    def register_mergability_running_dev(self, stream_ptr, cams, P, d_M, d_cov, d_slot, pixelErrVar, d_cache, d_mergeable, tolPix=0.5,
                                         d_counts=None, cam0=0, nCamsRun=None, d_list=None, nList=0, d_flags=None):
        """staticCheckMergability over WHOLE tracks as a running verdict (cs_register_mergability_running_dev): the newest histLen
        frames walked as they stand, the older ones' verdict cached per (point, camera) in d_cache (zero-filled, mergability_cache_bytes(P)).
        d_list / nList: only the rows d_list[0 .. nList) of the P-row tables (cs_register_list_current_dev's list); d_flags: the search's
        flags table -- candidates that already carry a map point are not judged"""
        vp = C.c_void_p
        check(self._L.cs_register_mergability_running_list_dev(vp(self._h), vp(stream_ptr), int(cam0),
                                                               int(self.nCams - cam0 if nCamsRun is None else nCamsRun), poseupdate_cams(cams),
                                                               int(P), vp(d_list), int(nList), vp(d_M), vp(d_cov), vp(d_slot), vp(d_flags),
                                                               C.c_double(pixelErrVar), C.c_double(tolPix), vp(d_cache), vp(d_mergeable),
                                                               vp(d_counts)), "cs_register_mergability_running_list_dev")

    def pose_update_frame_dev(self, stream_ptr, cams, d_pointFeat, nMap, d_R, d_t, d_mapPts, d_mapCov, d_mapFlags, largeErr,
                              pixelErrVar, frame, maxLen=20, minLen=5, minOutNum=3, maxEpiErr=6.0, d_numNodes=None, d_numOut=None,
                              d_numDyn=None):
        """Gate + dynamic test of all cameras in one launch."""
        vp = C.c_void_p
        arr = poseupdate_cams(cams)
        check(self._L.cs_pose_update_frame_dev(vp(self._h), vp(stream_ptr), arr, vp(d_pointFeat), int(nMap), vp(d_R), vp(d_t),
                                               vp(d_mapPts), vp(d_mapCov), vp(d_mapFlags), int(largeErr), C.c_double(pixelErrVar),
                                               int(frame), int(maxLen), int(minLen), int(minOutNum), C.c_double(maxEpiErr),
                                               vp(d_numNodes), vp(d_numOut), vp(d_numDyn)), "cs_pose_update_frame_dev")

    def pose_update_classify_frame_dev(self, stream_ptr, cams, d_pointFeat, nMap, d_R, d_t, d_mapPts, d_mapCov, d_mapFlags, largeErr, pixelErrVar,
                                       frame, d_newPt, d_staticFrameNum, d_firstFrame, pixelVarClassify=12.0, maxLen=20, minLen=5, minOutNum=3,
                                       maxEpiErr=6.0, d_numNodes=None, d_numOut=None, d_numDyn=None, d_featFrame=None, d_featFirst=None,
                                       d_counts=None):
        """pose_update_frame_dev + map_points_classify_dev of the same frame in two launches (cs_pose_update_classify_frame_dev)."""
        vp = C.c_void_p
        check(self._L.cs_pose_update_classify_frame_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), vp(d_pointFeat), int(nMap), vp(d_R),
                                                        vp(d_t), vp(d_mapPts), vp(d_mapCov), vp(d_mapFlags), int(largeErr), C.c_double(pixelErrVar),
                                                        int(frame), int(maxLen), int(minLen), int(minOutNum), C.c_double(maxEpiErr),
                                                        vp(d_numNodes), vp(d_numOut), vp(d_numDyn), vp(d_featFrame), vp(d_featFirst), vp(d_newPt),
                                                        vp(d_staticFrameNum), vp(d_firstFrame), C.c_double(pixelVarClassify), vp(d_counts)),
              "cs_pose_update_classify_frame_dev")

    def set_poses_dev(self, stream_ptr, n, d_cam, d_frame, d_R, d_t):
        """Poses of n (camera, frame) pairs into the ring (RobustBundleRTS::output(): adjusted key poses and relaxed non-key poses,
        reference src/app/SL_CoSLAMRobustBA.cpp:283-285, 239-244); pairs the ring does not hold are skipped."""
        vp = C.c_void_p
        check(self._L.cs_track_history_set_poses_dev(vp(self._h), vp(stream_ptr), int(n), vp(d_cam), vp(d_frame), vp(d_R), vp(d_t)),
              "cs_track_history_set_poses_dev")

    def get_span_dev(self, stream_ptr, first_frame, n_frames, d_R, d_t):
        """poses of frames first_frame .. first_frame + n_frames - 1 of every camera out of the ring, [nCams][n_frames][9] / [3]"""
        vp = C.c_void_p
        check(self._L.cs_track_history_get_span_dev(vp(self._h), vp(stream_ptr), int(first_frame), int(n_frames), vp(d_R), vp(d_t)),
              "cs_track_history_get_span_dev")

    def set_span_dev(self, stream_ptr, first_frame, n_frames, d_R, d_t):
        vp = C.c_void_p
        check(self._L.cs_track_history_set_span_dev(vp(self._h), vp(stream_ptr), int(first_frame), int(n_frames), vp(d_R), vp(d_t)),
              "cs_track_history_set_span_dev")

    @property
    def newest_frame(self):
        return self._L.cs_track_history_newest_frame(C.c_void_p(self._h))

    def update_new_poses_points_dev(self, stream_ptr, cams, d_pointFeat, nMap, d_mapPts, d_mapCov, d_mapFlags, pixelErrVar,
                                    d_lastFrame=None, d_isCurrent=None, firstKeyFrame=-1, d_counts=None):
        """RobustBundleRTS::updateNewPosesPoints (reference src/app/SL_CoSLAMRobustBA.cpp:248-271): every map point seen after the
        window's first key frame is triangulated again from the ring's (adjusted) poses, in place."""
        vp = C.c_void_p
        check(self._L.cs_update_new_poses_points_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), vp(d_pointFeat), int(nMap),
                                                     vp(d_lastFrame), vp(d_isCurrent), int(firstKeyFrame), vp(d_mapPts), vp(d_mapCov),
                                                     vp(d_mapFlags), C.c_double(pixelErrVar), vp(d_counts)),
              "cs_update_new_poses_points_dev")

    def refine_map_points_dev(self, stream_ptr, cams, d_pointFeat, nMap, d_mapPts, d_mapCov, pixelErrVar, d_select=None, d_count=None):
        """CoSLAM::refineMapPoint (reference src/app/SL_CoSLAM.cpp:666-713) for the selected map points (uint8 mask; None = all), in
        place: what the registration loops call on a point that has just gained a feature."""
        vp = C.c_void_p
        check(self._L.cs_refine_map_points_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), vp(d_pointFeat), int(nMap), vp(d_select),
                                               vp(d_mapPts), vp(d_mapCov), C.c_double(pixelErrVar), vp(d_count)),
              "cs_refine_map_points_dev")

    def check_unify_dev(self, stream_ptr, cams, nPairs, d_pf1, d_pf2, d_M1, d_M2, pixelErrVar, d_ok, d_M, d_cov):
        """CoSLAM::checkUnify (reference src/app/SL_CoSLAM.cpp:561-665) for nPairs pairs of map points, a wave per pair"""
        vp = C.c_void_p
        check(self._L.cs_check_unify_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), int(nPairs), vp(d_pf1), vp(d_pf2), vp(d_M1), vp(d_M2),
                                         C.c_double(pixelErrVar), vp(d_ok), vp(d_M), vp(d_cov)), "cs_check_unify_dev")

    def newpts_intracam_scratch_bytes(self):
        self._L.cs_newpts_intracam_scratch_bytes.restype = C.c_size_t
        return int(self._L.cs_newpts_intracam_scratch_bytes(self.nCams, self.N))

    def newpts_intracam_dev(self, stream_ptr, cams, d_mapPts, d_mapCov, d_mapFlags, d_newPt, d_firstFrame, d_pointFeat, mapCap, d_mapCount, d_scratch,
                            pixelErrVar, d_ready=None, readyMin=2, minTrackLen=20, maxWalk=1024, maxEpiErr=2.0, d_counts=None):
        """cs_newpts_intracam_dev: SingleSLAM::newMapPoints (reference src/app/SL_SingleSLAM.cpp:922-1004) for the cameras d_ready selects"""
        vp = C.c_void_p
        check(self._L.cs_newpts_intracam_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), vp(d_ready), int(readyMin), int(minTrackLen),
                                             int(maxWalk), C.c_double(maxEpiErr), C.c_double(pixelErrVar), vp(d_mapPts), vp(d_mapCov),
                                             vp(d_mapFlags), vp(d_newPt), vp(d_firstFrame), vp(d_pointFeat), int(mapCap), vp(d_mapCount),
                                             vp(d_scratch), vp(d_counts)), "cs_newpts_intracam_dev")

    # ---- MapPoint::pFeatures as feature references (cs_feat_ref / cs_feat_seg, include/coslam_hip.h) --------------------------------
    def set_classify_refs(self, d_featRef, d_refStatic=None):
        """cs_track_history_set_classify_refs: the classification (map_points_classify_dev, pose_update_classify_frame_dev) reads the points'
        features as references from now on (stale features are views, the walks follow the links); d_featRef None: pointFeat alone again"""
        vp = C.c_void_p
        check(self._L.cs_track_history_set_classify_refs(vp(self._h), vp(d_featRef), vp(d_refStatic)), "cs_track_history_set_classify_refs")

    def feat_ref_advance_dev(self, stream_ptr, cams, nMap, d_pointFeat, curFrame, d_featRef, d_refStatic=None, d_counts=None, d_list=None, nList=0):
        """cs_feat_ref_advance_(list_)dev: every frame behind the registration's decisions -- tracked on / first feature / re-linked behind an
        older one (reference src/app/SL_CoSLAM.cpp:775-779) / stale / detached.  d_list / nList: a further call within the frame over those rows only"""
        vp = C.c_void_p
        check(self._L.cs_feat_ref_advance_list_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), int(nMap), vp(d_pointFeat), int(curFrame),
                                                   vp(d_featRef), vp(d_refStatic), vp(d_counts), vp(d_list), int(nList)), "cs_feat_ref_advance_list_dev")

    def load_segments(self, segs):
        """cs_track_history_load_segments: segs int32 [nCams][n][4] = {slot, last, first, next} from the host into the pools"""
        import numpy as np

        a = np.ascontiguousarray(segs, dtype=np.int32)
        assert a.ndim == 3 and a.shape[0] == self.nCams and a.shape[2] == 4
        check(self._L.cs_track_history_load_segments(C.c_void_p(self._h), a.ctypes.data_as(C.c_void_p), int(a.shape[1])),
              "cs_track_history_load_segments")

    def segments(self):
        """the cameras' pools of linked segments as they stand: (nCams, n, 4) int32 {slot, last, first, next}, n = the fullest pool's count"""
        import numpy as np

        cnt, _ = self.segment_counts()
        n = max(int(cnt.max()), 1)
        out = np.full((self.nCams, n, 4), -1, dtype=np.int32)
        check(self._L.cs_track_history_download_segments(C.c_void_p(self._h), out.ctypes.data_as(C.c_void_p), n), "cs_track_history_download_segments")
        return out

    def set_merge_refs(self, d_featRef, d_refStatic=None):
        """cs_track_history_set_merge_refs: the bMerge walks (register_decide_merge_dev) take the points' features as references from now on --
        checkUnify over stale features and linked chains, a new feature behind a stale one linked at once, the hand-over of a unification as the
        reference's loop does it (a stale feature blocks it in its camera, the other point's stale features move too); None: pointFeat alone"""
        vp = C.c_void_p
        check(self._L.cs_track_history_set_merge_refs(vp(self._h), vp(d_featRef), vp(d_refStatic)), "cs_track_history_set_merge_refs")

    def segment_counts(self):
        """cs_track_history_segment_counts: how many linked segments every camera's pool holds (a synchronous read) and the capacity"""
        import numpy as np

        out, cap = np.zeros(self.nCams, dtype=np.int32), C.c_int()
        check(self._L.cs_track_history_segments(C.c_void_p(self._h), None, C.byref(cap), None), "cs_track_history_segments")
        check(self._L.cs_track_history_segment_counts(C.c_void_p(self._h), out.ctypes.data_as(C.c_void_p)), "cs_track_history_segment_counts")
        return out, int(cap.value)

    def feat_ref_advance_refine_dev(self, stream_ptr, cams, nMap, d_pointFeat, curFrame, d_featRef, d_refStatic, d_list, nList, advanceAll, d_select,
                                    clearSelect, d_mapPts, d_mapCov, pixelErrVar, d_counts=None):
        """cs_feat_ref_advance_refine_dev: the references' advance and refineMapPoint as ONE launch -- the listed rows with d_select set are
        advanced and refined, every other row (of the whole map when advanceAll, else of the list) is advanced only; clearSelect: the marks
        are consumed"""
        vp = C.c_void_p
        self._L.cs_feat_ref_advance_refine_dev.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_double]
        check(self._L.cs_feat_ref_advance_refine_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), int(nMap), vp(d_pointFeat), int(curFrame),
                                                     vp(d_featRef), vp(d_refStatic), vp(d_counts), vp(d_list), int(nList), int(bool(advanceAll)),
                                                     vp(d_select), int(bool(clearSelect)), vp(d_mapPts), vp(d_mapCov), float(pixelErrVar)),
              "cs_feat_ref_advance_refine_dev")

    def update_new_poses_points_ref_dev(self, stream_ptr, cams, d_featRef, nMap, d_mapPts, d_mapCov, d_mapFlags, pixelErrVar,
                                        d_refStatic=None, d_lastFrame=None, d_isCurrent=None, firstKeyFrame=-1, d_counts=None):
        """cs_update_new_poses_points_ref_dev: updateNewPosesPoints with stale features as views and walks that follow the links"""
        vp = C.c_void_p
        check(self._L.cs_update_new_poses_points_ref_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), vp(d_featRef), vp(d_refStatic),
                                                         int(nMap), vp(d_lastFrame), vp(d_isCurrent), int(firstKeyFrame), vp(d_mapPts),
                                                         vp(d_mapCov), vp(d_mapFlags), C.c_double(pixelErrVar), vp(d_counts)),
              "cs_update_new_poses_points_ref_dev")

    def refine_map_points_ref_dev(self, stream_ptr, cams, d_featRef, nMap, d_mapPts, d_mapCov, pixelErrVar, d_select=None, d_count=None):
        """cs_refine_map_points_ref_dev: CoSLAM::refineMapPoint over feature references"""
        vp = C.c_void_p
        check(self._L.cs_refine_map_points_ref_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), vp(d_featRef), int(nMap), vp(d_select),
                                                   vp(d_mapPts), vp(d_mapCov), C.c_double(pixelErrVar), vp(d_count)),
              "cs_refine_map_points_ref_dev")

    def check_unify_ref_dev(self, stream_ptr, cams, nPairs, d_ref1, d_ref2, d_M1, d_M2, pixelErrVar, d_ok, d_M, d_cov):
        """cs_check_unify_ref_dev: CoSLAM::checkUnify with the two points' features as references ([nPairs][nCams] cs_feat_ref each)"""
        vp = C.c_void_p
        check(self._L.cs_check_unify_ref_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), int(nPairs), vp(d_ref1), vp(d_ref2), vp(d_M1),
                                             vp(d_M2), C.c_double(pixelErrVar), vp(d_ok), vp(d_M), vp(d_cov)), "cs_check_unify_ref_dev")

    def decide_merge_scratch_bytes(self, P, nList):
        """d_scratch of register_decide_merge_dev with a list (without: P bytes)"""
        self._L.cs_register_decide_merge_scratch_bytes.restype = C.c_size_t
        return int(self._L.cs_register_decide_merge_scratch_bytes(int(P), int(nList), self.nCams))

    def register_decide_merge_dev(self, stream_ptr, cams, P, mapBase, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, d_mapPts, d_mapCov,
                                  pixelErrVar, d_attached, d_regged, d_scratch, d_counts=0, only_cam=-1, d_list=None, nList=0):
        """curStaticPointsRegInGroup with bMerge == true (reference src/app/SL_CoSLAM.cpp:854-898, 731-830), the walks in the reference's
        order on one wave: attach, or ask checkUnify at a feature of another static point and unify on a yes (cs_register_decide_merge_dev)"""
        vp = C.c_void_p
        check(self._L.cs_register_decide_merge_list_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), int(P), int(mapBase), vp(d_list),
                                                        int(nList), vp(d_slot), vp(d_flags), vp(d_mergeable), vp(d_mapFlags), vp(d_pointFeat),
                                                        vp(d_mapPts), vp(d_mapCov), C.c_double(pixelErrVar), vp(d_attached), vp(d_regged),
                                                        vp(d_scratch), vp(d_counts), int(only_cam)), "cs_register_decide_merge_list_dev")

    def map_points_classify_dev(self, stream_ptr, cams, d_pointFeat, nMap, curFrame, d_mapPts, d_mapCov, d_mapFlags, d_newPt,
                                d_staticFrameNum, d_firstFrame, pixelVar=12.0, d_featFrame=None, d_featFirst=None, d_counts=None):
        """CoSLAM::mapPointsClassify (reference src/app/SL_CoSLAM.cpp:418-520; CoSLAM::poseUpdate calls it with 12.0 every frame):
        the uncertain and the dynamic map points of this frame decided again, in place."""
        vp = C.c_void_p
        check(self._L.cs_map_points_classify_dev(vp(self._h), vp(stream_ptr), poseupdate_cams(cams), vp(d_pointFeat), int(nMap),
                                                 vp(d_featFrame), vp(d_featFirst), int(curFrame), vp(d_mapPts), vp(d_mapCov),
                                                 vp(d_mapFlags), vp(d_newPt), vp(d_staticFrameNum), vp(d_firstFrame),
                                                 C.c_double(pixelVar), vp(d_counts)), "cs_map_points_classify_dev")
