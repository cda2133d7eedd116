"""The body of CoSLAM's main loop over device-resident state, one process per GPU.

What the reference runs per frame on its tracking thread (src/gui/CoSLAMThread.cpp:95-125: grabReadFrame, featureTracking, poseUpdate,
activeMapPointsRegister, genNewMapPoints, currentMapPointsRegister) with the bundle adjuster on a second thread
(src/app/SL_CoSLAM.cpp:1702-1784), as streams of libcoslam_hip calls:

    tracker stream   redetect of the rank's OWN cameras (one camera group)                          [featureTracking -> GPUKLT::next]
    pose stream      BA output due this frame (cs_ba_output_apply_dev)                              [RobustBundleRTS::output]
                     hand-back + intraCamEstimate of the own cameras                                [poseUpdate3D, first half]
                     N > 1: ONE all-gather of every camera's {dest[], R, t}; hand-back of the OTHER ranks' cameras from the gathered
                            dest[] -- every rank then holds every camera's records and poses, bit for bit
                     gate + seqTriangulate + dynamic test over ALL cameras, mapPointsClassify          [poseUpdate3D second half, :381-385]
                     registration search + staticCheckMergability for the own cameras' columns       [activeMapPointsRegister, currentMapPointsRegister]
                     every 4th frame: NCC matching of the own consecutive camera pairs              [genNewMapPoints -> NewMapPtsNCC]
    key frames       every rank pushes the key frame into its window ring (identical on every rank); window k is SOLVED by rank
                     k mod N (cs_ba_solve_window_flags_async on that rank's worker thread), the inter-camera solve by rank
                     (k + N / 2) mod N; `ba_lag` key-frame intervals later the solving rank broadcasts the packed result and EVERY rank
                     applies it to its replica of the map, the pose history and the window ring at the same frame.

Because every step that writes the map (gate, classify, BA output) runs on every rank over all cameras in the same order on the same
inputs, the replicas never diverge: there is ONE map, held N times.  What is sharded is what is per camera and costs the time: tracking,
the pose solve, the registration search, the NCC cutter.  What added GPUs buy on the key-frame side is TIME: a rank solves every N-th
window and has N key-frame intervals for it, so the solve's latency chain leaves the frame's critical path.
The apply lag makes a run reproducible (the reference applies a result whenever its BA thread gets the lock, :1713-1720).
"""
import ctypes as C

import numpy as np

PIXEL_ERR_VAR = 10.0      # Const::PIXEL_ERR_VAR, reference src/app/SL_GlobParam.cpp:37
MAX_EPI_ERR = 6.0         # Const::MAX_EPI_ERR, :36


class LoopConfig:
    def __init__(self, **kw):
        self.n_cams, self.W, self.H, self.levels, self.fw, self.fh = 8, 640, 480, 4, 50, 40
        self.pts_stride, self.n_col_blk, self.n_row_blk = 192, 16, 12    # reference src/app/SL_SingleSLAM.h:36-37
        self.key_every, self.n_key_frames = 5, 5                         # requestForBA(5, 2, 2, 30), SL_CoSLAM.cpp:1345
        self.ic_workers = 1        # workspaces (worker threads) the inter-camera solves of this rank rotate over (2 on one GPU: measured slower)
        self.ba_lag = 0            # key-frame intervals between a window's key frame and the frame its result is applied; 0 = min(max(N, 2), 4)
        self.p_reg = 4096          # cap on the frame's CURRENT map points (the registration's list: cs_register_list_current_dev)
        self.hist = 64             # depth of the bounded walks (dynamic test, classification, re-triangulation; the mergability walk's exact window)
        self.hist_store = 4096     # frames of pixels + poses KEPT behind them (1 GB for 8 x 2000 slots): what the running whole-track
                                   # mergability verdict rebuilds a cached tail from (cs_register_mergability_running_dev)
        self.verdict_check_every = 100   # diagnostic: every n-th frame the running verdict's candidates are judged again without cached tails (0: never)
        self.merge_tol_pix = 0.5   # a point that moved further than this in a camera's image has its cached tail judged again
        self.pixel_err_reading = "variance"   # Const::PIXEL_ERR_VAR = 10 (src/app/SL_GlobParam.cpp:37) reaches getProjectionCovMat / seqTriangulate /
        # getTriangulateCovMat (un-vendored LibVisualSLAM) as their last argument.  This library's definitions of them take a STANDARD
        # DEVIATION (J cov J^T + s^2 I).  "variance": the constant is what its name says -- the reference's own retired define reads
        # `SLAM_PIXEL_ERR_VAR 4 //2 pixels error` (src/slam/SL_Define.h:16) -- so the loop hands the helpers sqrt(10) = 3.16 px;
        # "std": the constant handed over as it is (a 10 px gate: what rounds 1-4 ran).  DESIGN.md 5.1.
        self.with_active_search = False   # the search half of activeMapPointsRegister: OFF -- the reference's own attach loop cannot be
        # reached (every point of actMapPts has numVisCam == 0, src/app/SL_CoSLAM.cpp:1114 asks for > 0: tests/cxx/ref_active_test.cpp runs
        # the reference's code to say so), so the tables this pass used to fill fed nothing
        self.ncc_every, self.ncc_pair_cap = 4, 1 << 16
        self.map_spare = 8192      # room for new map points behind the initial map
        self.klt_cams_per_launch = 0
        self.klt_xcd_placement = True
        self.intracam_mapping = False    # SingleSLAM::newMapPoints for the cameras the key-frame decision calls ready (genNewMapPoints' first half,
        # reference src/app/SL_CoSLAM.cpp:1294-1346): every unmapped feature on a track of 20+ frames becomes a map point from its own track
        # (cs_newpts_intracam_dev).  Implies keyframe_decision.  Off in the headline: ~50 new points per frame need a map that recycles its
        # points; with map_spare raised the loop runs on it for as long as the capacity lasts
        self.keyframe_decision = False   # CoSLAM::IsReadyForKeyFrame + addKeyFrame's key-pose state per frame on the device (cs_keyframe_ready_dev),
        # REPORTED (FrameLoop.keyframe_stats); the key frames themselves stay on the fixed key_every cadence unless keyframe_drives
        self.keyframe_drives = False     # the decision PLACES the key frames (genNewMapPoints :1331-1346: `decrease` -> addKeyFrame, requestForBA):
        # step()'s key_frame argument is ignored, the host reads `decrease` back every frame (a wait per frame, as the reference's one
        # thread has it), the windows' key frames fall where they fall and their results are applied through cs_ba_output_apply_frames_dev.
        # A window whose first key frame has left the pose history (hist_store frames) by the time its result is due is not applied (counted:
        # FrameLoop.keyframe_stats()["windows_not_applied_history_too_short"]).  One rank only.  Implies keyframe_decision.  Off in the headline:
        # in the bench's world the decision never says `decrease` -- there would be no window bundle adjustment to measure
        self.keyframe_lag = 0            # keyframe_drives WITHOUT a host wait per frame: D > 0 = the host acts on the decision of frame i - D, which it reads
        # from pinned memory behind that frame's event (complete long before: the host is never more than D frames ahead of what it reads, the
        # device never waits for the host) -- the key frame's records and poses are kept in a ring of D + 1 snapshots, its window is requested D
        # frames late (the request reads the map as it stands then) and applied at the same frame as before.  0: the read-back of rounds 5-6
        self.keyframe_ratio = 0.93       # m_mappedPtsReduceRatio (reference src/app/SL_CoSLAM.cpp:42)
        self.fused_registration = True   # the registration's launches fused as tools/cxx/frame_loop.cpp runs them: the second visits' lists built by the
        # walks (cs_register_decide_kinds_rounds_dev, cs_register_revisit_decide_next_dev), advance + refine as one launch
        # (cs_feat_ref_advance_refine_dev); needs feature_chains.  False: a launch per step (the form DESIGN.md 3.13.1 describes first)
        self.merge_refs = True       # the bMerge walks read and write the references too (checkUnify over stale features and chains, the hand-over as the
        # reference's loop over pFeatures does it: cs_track_history_set_merge_refs)
        self.classify_refs = True    # mapPointsClassify reads the references too (stale features, linked segments: cs_track_history_set_classify_refs)
        self.feature_chains = True   # MapPoint::pFeatures kept as feature references (cs_feat_ref): a camera that lost a point still contributes
        # its last feature to refineMapPoint / updateNewPosesPoints, and a point registered to a new track where it held an older feature has
        # the old chain linked behind it (reference src/app/SL_CoSLAM.cpp:775-779); False: the features of this frame on their own tracks
        self.klt_fused = True      # False: one launch per Gauss-Newton pass (bit-identical): for SEVERAL processes sharing one GPU, where the
                                   # persistent tracker's co-residency budget does not hold
        self.prefetch = True
        self.with_pose_update = self.with_classify = self.with_register = self.with_mergability = self.with_ncc = True
        self.with_joint = self.with_intercam = True
        self.with_decide = True    # the registration decision (who attaches which feature) + refineMapPoint of the points that gained one
        self.merge_every = 50      # bMerge: every 50th frame the static points' walks may UNIFY two points (CoSLAMThread.cpp:117-118:
        # `i % 50 == 0`; cs_register_decide_merge_dev, sequential); 0: never
        self.revisit_rounds = 2   # the reference's SECOND VISITS behind the single-pass registration: a point that registered is refined and
        #                           visited again in its next camera's loop (SL_CoSLAM.cpp:864-869, :889-893) -- rounds of list + search +
        #                           mergability + walks + refine over just those points (cs_register_revisit_*).  0: the single pass alone
        self.sequential_registration = False   # the decision camera loop after camera loop with a search + refine per loop, as the
        # reference runs it (register_cur_static_sequential_dev: bit-identical to the reference's run, nCams x the launches; one rank only)
        self.native_comm = True
        self.klt_cus = 0           # > 0: the tracker's stream is confined to CU-mask bits [0, klt_cus) (the same klt_cus / 8 CUs of every XCD)
        self.pose_cus = 0          # > 0: the pose stream is confined to the LAST pose_cus mask bits (with klt_cus + pose_cus <= 256: disjoint)
        self.klt_after_intracam = False   # the tracker of frame i + 1 starts behind frame i's intraCamEstimate: the pose solve -- one
        # workgroup per camera, a latency chain of ~20 LM steps -- then runs on an otherwise empty chip instead of beside the
        # persistent tracker's resident waves ("what the pose stream pays for co-residency")
        self.device_wait = True    # the BA result's apply waits for the solve on the device (cs_ba_output_wait_dev), not on the host
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError(f"LoopConfig: unknown field {k}")
            setattr(self, k, v)

    @property
    def n_feat(self):
        return self.fw * self.fh


class FrameLoop:
    """Device state of ONE rank and the enqueue of one frame.  `video`: dict camera -> uint8 tensor [T, H, W] on the device for
    the rank's own cameras; `scene`: K, points (the map), pose(cam, frame) (first frame's poses, F matrices of the NCC leg);
    `ic`: None = the inter-camera problem is built on the device from the frame's records (cs_ba_solve_intercam_async), else a
    pre-baked problem (coslam_amd.synth.make_intercam_problem) re-solved from the current poses; `klt_cfg`: KLT_SequenceTrackerConfig."""

    def __init__(self, cfg, scene, video, ic, klt_cfg, map_cov, rank=0, world=1, device=0, dist_backend="nccl", associate=None):
        import torch

        import coslam_amd
        from coslam_amd.ba import BAOutput, BAWindow, BAWorkspace
        from coslam_amd.handback import handback_cams
        from coslam_amd.poseupdate import TrackHistory, poseupdate_cams
        from coslam_amd.register import register_cams, register_passes

        self.torch = torch
        self.cfg, self.sc, self.ic = cfg, scene, ic
        if cfg.pixel_err_reading not in ("variance", "std"):
            raise ValueError("LoopConfig.pixel_err_reading: 'variance' or 'std'")
        self.sig = (lambda v: float(np.sqrt(v))) if cfg.pixel_err_reading == "variance" else (lambda v: float(v))
        self.sig_pix = self.sig(PIXEL_ERR_VAR)   # what the helpers get where the reference passes Const::PIXEL_ERR_VAR
        self.rank, self.world, self.device = rank, world, device
        NA, N = cfg.n_cams, cfg.n_feat
        if NA % world:
            raise ValueError(f"{NA} cameras do not shard over {world} ranks")
        if cfg.keyframe_drives and world > 1 and cfg.keyframe_lag <= 0:
            raise ValueError("LoopConfig.keyframe_drives: one rank only (the host reads the decision back every frame); keyframe_lag > 0 has every "
                             "rank read its replica's decision")
        self.nc = nc = NA // world
        self.c0 = c0 = rank * nc
        self.my_cams = list(range(c0, c0 + nc))
        self.lag = cfg.ba_lag if cfg.ba_lag > 0 else min(max(world, 2), 4)
        if (cfg.n_key_frames - 1 + self.lag) * cfg.key_every + 1 > cfg.hist:
            raise ValueError("the pose history is shorter than a window + its apply lag")
        dev = self.dev = torch.device("cuda", device)
        self.T = int(next(iter(video.values())).shape[0])
        self.video = video
        self.associate = associate
        # the map: structure-of-arrays with spare capacity (new map points are appended: cs_newpts_from_pairs_dev); unused entries carry
        # no feature anywhere, which makes them inert for every kernel
        n_pts = len(scene.points)
        n_map = self.n_map = n_pts + cfg.map_spare
        f64, i32, u8 = torch.float64, torch.int32, torch.uint8
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)   # noqa: E731
        K = np.ascontiguousarray(scene.K, dtype=np.float64)
        self.d_K = torch.from_numpy(np.tile(K.ravel(), NA)).to(dev)
        self.d_K1 = torch.from_numpy(K.ravel().copy()).to(dev)
        self.d_iK1 = torch.from_numpy(np.linalg.inv(K).ravel().copy()).to(dev)
        self.d_kud = z(7, f64)
        # ---- the ONE map (a replica per rank) and every camera's records
        self.d_map = z((n_map, 3), f64)
        self.d_map[:n_pts] = torch.from_numpy(np.ascontiguousarray(scene.points, dtype=np.float64).copy()).to(dev)
        self.d_cov = z(n_map * 9, f64)
        self.d_cov[:n_pts * 9] = torch.from_numpy(np.ascontiguousarray(map_cov, dtype=np.float64).reshape(-1)[:n_pts * 9].copy()).to(dev)
        self.d_mapcount = torch.tensor([n_pts], dtype=i32, device=dev)
        self.n_pts0 = n_pts
        self.d_mapflags = z(n_map, u8)
        self.d_newpt, self.d_sfn, self.d_firstfrm = z(n_map, u8), z(n_map, i32), z(n_map, i32)
        self.d_slot2map = torch.full((NA, N), -1, dtype=i32, device=dev)
        self.d_trackspan = torch.full((NA, 2 * N), -1, dtype=i32, device=dev)
        self.d_xy, self.d_state = z((NA, 2 * N), f64), z((NA, N), i32)
        self.d_Ms, self.d_ms, self.d_sel = z((NA, cfg.pts_stride, 3), f64), z((NA, cfg.pts_stride, 2), f64), z((NA, cfg.pts_stride), i32)
        self.d_npts, self.d_opt, self.d_ok = z(NA, i32), z((NA, 96), u8), z(NA, i32)
        self.d_isstatic, self.d_reproj = torch.ones((NA, N), dtype=u8, device=dev), z((NA, N), f64)
        self.d_pf = torch.full((n_map, NA), -1, dtype=i32, device=dev)          # MapPoint::pFeatures of this frame (hand-back)
        R0 = np.stack([scene.pose(c, 0)[0].ravel() for c in range(NA)])
        t0 = np.stack([scene.pose(c, 0)[1] for c in range(NA)])
        self.d_R = [torch.from_numpy(R0.copy()).to(dev), torch.from_numpy(R0.copy()).to(dev)]   # pose ping-pong: frame i reads [(i+1)&1]
        self.d_t = [torch.from_numpy(t0.copy()).to(dev), torch.from_numpy(t0.copy()).to(dev)]
        self.d_dests = [[z(N * 5, i32) for _ in range(nc)] for _ in range(2)]
        self.d_counts = [z(4, i32) for _ in range(nc)]
        self.d_cls_counts, self.d_apply_counts = z(2, i32), z(3, i32)
        # the registration's tables: indexed by the MAP index (whole-map tables); the frame's current points as a compact list
        self.d_mergeable = z((n_map, NA), u8)
        self.reg_out = dict(slot=torch.full((n_map, NA), -1, dtype=i32, device=dev), m=z((n_map, NA, 2), f64), var=z((n_map, NA, 4), f64),
                            dist=z((n_map, NA), f64), flags=z((n_map, NA), i32))
        self.d_curlist, self.d_curcount = torch.full((n_map,), -1, dtype=i32, device=dev), z(1, i32)
        self.d_curoverflow = z(1, i32)   # current points beyond the list's cap (left out of that frame's registration; 0 in every bench configuration)
        self.d_merge_counts = z(4, i32)   # running mergability: cache hits, full tail walks, verdicts 2, tail terms (summed over the run)
        # ---- streams
        self.klt_s, self.pose_s = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)   # (equal priorities: a high-priority pose or
        # tracker stream halves the rate, 2188 -> 942 / 906 frames/s: profiles/r04_ab_runs.txt)
        if cfg.klt_cus > 0 or cfg.pose_cus > 0:
            # CU-partitioned streams (cs_stream_create_cu_range = hipExtStreamCreateWithCUMask): VERDICT r04 item 3
            L_ = coslam_amd.lib()
            L_.cs_stream_create_cu_range.restype = C.c_void_p
            L_.cs_stream_create_cu_range.argtypes = [C.c_int, C.c_int, C.c_int]
            if cfg.klt_cus > 0:
                h_ = L_.cs_stream_create_cu_range(device, 0, int(cfg.klt_cus))
                if not h_:
                    raise coslam_amd.CoslamHipError("cs_stream_create_cu_range: " + L_.cs_last_error().decode())
                self.klt_s = torch.cuda.ExternalStream(h_, device=dev)
            if cfg.pose_cus > 0:
                h_ = L_.cs_stream_create_cu_range(device, 256 - int(cfg.pose_cus), int(cfg.pose_cus))
                if not h_:
                    raise coslam_amd.CoslamHipError("cs_stream_create_cu_range: " + L_.cs_last_error().decode())
                self.pose_s = torch.cuda.ExternalStream(h_, device=dev)
        self.intracam_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.klt_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.dest_free = [torch.cuda.Event(), torch.cuda.Event()]
        # ---- trackers of the own cameras
        self.trks = []
        for _ in self.my_cams:
            t = coslam_amd.KLT_SequenceTracker(klt_cfg, device=device)
            t.allocate(cfg.W, cfg.H, cfg.levels, cfg.fw, cfg.fh)
            if not cfg.klt_fused:
                t.set_fused(False)
            self.trks.append(t)
        self.grp = coslam_amd.KLT_TrackerGroup(self.trks)
        self.grp.set_stream(self.klt_s.cuda_stream)
        for t in self.trks:       # a camera's tracker workgroups on that camera's own XCD (one L2 per pyramid pair)
            t.set_xcd_placement(cfg.klt_xcd_placement)
        if cfg.klt_cams_per_launch > 0:
            for t in self.trks:   # co-residency budget of the persistent tracker = that many cameras per launch
                t.set_cu_count(min(256, (250 * cfg.klt_cams_per_launch + 60) // 8 + 5))
        elif cfg.klt_cus > 0:
            for t in self.trks:   # the persistent tracker's co-residency budget = the CUs its stream may use
                t.set_cu_count(int(cfg.klt_cus))
        # ---- multi-GPU exchange
        self.xchg = self.native = None
        if world > 1:
            from coslam_amd import multicam

            self.native_fallback = None
            if cfg.native_comm and dist_backend == "nccl":
                # the library's own RCCL communicators; if ANY rank cannot bring them up, ALL ranks use torch.distributed's collectives
                # instead (agreed through one all-reduce) -- said loudly: stderr here, `collectives` in bench.py's line
                import torch.distributed as dist_

                err = None
                try:
                    self.native = multicam.NativeComm(world, rank, device)
                except coslam_amd.CoslamHipError as ex:
                    err = str(ex)
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
                dist_.all_reduce(ok, op=dist_.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    if self.native is not None:
                        self.native.close()
                        self.native = None
                    self.native_fallback = err or "another rank could not create its communicator"
                    print(f"[frameloop rank {rank}] libcoslam_hip's RCCL communicator is not available ({self.native_fallback}): "
                          "torch.distributed collectives instead", file=__import__("sys").stderr, flush=True)
            self.xchg = multicam.CameraExchange(N * nc, dev, native=self.native, cams_per_rank=nc)
        # ---- per-camera argument tables (built once)
        def hb_cam(g, dest):
            return dict(dest=dest, K=self.d_K1.data_ptr(), kud=self.d_kud.data_ptr(), mapPts=self.d_map.data_ptr(),
                        slot2map=self.d_slot2map[g].data_ptr(), trackSpan=self.d_trackspan[g].data_ptr(), xy=self.d_xy[g].data_ptr(),
                        state=self.d_state[g].data_ptr(), Ms=self.d_Ms[g].data_ptr(), ms=self.d_ms[g].data_ptr(), sel=self.d_sel[g].data_ptr(),
                        npts=self.d_npts[g:g + 1].data_ptr(), opt=self.d_opt[g].data_ptr(), pointFeat=self.d_pf.data_ptr() + 4 * g,
                        pointFeatStride=NA, nPointFeat=n_map, isStatic=(self.d_isstatic[g].data_ptr() if cfg.with_pose_update else 0))

        self.hb_own = [handback_cams([hb_cam(c0 + i, self.d_dests[b][i].data_ptr()) for i in range(nc)]) for b in range(2)]
        self.hb_other = None
        if world > 1:
            others = [g for g in range(NA) if g not in self.my_cams]
            self.hb_other = handback_cams([hb_cam(g, self.xchg.record_ptr(g, device)) for g in others])
        # (the window's push reads xy / state / slot2map of every camera; dest is not looked at)
        self.hb_all = handback_cams([hb_cam(g, self.d_dests[0][0].data_ptr()) for g in range(NA)])
        self.dest_ptrs = [[d.data_ptr() for d in self.d_dests[b]] for b in range(2)]
        self.cnt_ptrs = [c.data_ptr() for c in self.d_counts]
        self.img_ptrs = [[self.video[c][f].data_ptr() for c in self.my_cams] for f in range(self.T)]
        self.pose_upd = self.d_fref = None
        if cfg.with_pose_update:
            self.pose_upd = TrackHistory(NA, N, cfg.hist, device=device, storeLen=max(cfg.hist_store, cfg.hist))
            self.d_merge_cache = z(self.pose_upd.mergability_cache_bytes(n_map), u8)
            self.d_fref = torch.full((n_map, NA, 4), -1, dtype=i32, device=dev) if cfg.feature_chains else None
            self.d_rstat = z((n_map, NA), u8) if cfg.feature_chains else None
            self.d_fref_counts = z(5, i32)   # tracked on, first features, re-linked, links dropped (pool full), detached -- summed over the run
            if self.d_fref is not None and cfg.classify_refs:
                self.pose_upd.set_classify_refs(self.d_fref.data_ptr(), self.d_rstat.data_ptr())
            if self.d_fref is not None and cfg.merge_refs:
                self.pose_upd.set_merge_refs(self.d_fref.data_ptr(), self.d_rstat.data_ptr())
            self.pu_args = poseupdate_cams([dict(K=self.d_K1.data_ptr(), iK=self.d_iK1.data_ptr(), xy=self.d_xy[g].data_ptr(),
                                                 state=self.d_state[g].data_ptr(), slot2map=self.d_slot2map[g].data_ptr(),
                                                 trackSpan=self.d_trackspan[g].data_ptr(), reprojErr=self.d_reproj[g].data_ptr(),
                                                 isStatic=self.d_isstatic[g].data_ptr()) for g in range(NA)])
        self.reg_args = [register_cams([dict(K=self.d_K1.data_ptr(), R=self.d_R[b].data_ptr() + 72 * g, t=self.d_t[b].data_ptr() + 24 * g,
                                             xy=self.d_xy[g].data_ptr(), state=self.d_state[g].data_ptr(),
                                             slot2map=self.d_slot2map[g].data_ptr(), isStatic=self.d_isstatic[g].data_ptr())
                                        for g in range(NA)]) for b in range(2)]
        # CoSLAMThread.cpp:117 currentMapPointsRegister, search step: ONE pass over the frame's current points (the list), wherever they sit
        # in the map -- the points genNewMapPoints appended this frame included; it serves the static AND the dynamic registration (the
        # certainly dynamic points are searched with their own scale, SL_CoSLAM.cpp:973)
        o_ = self.reg_out
        # (maxDist only scales every distance of a search alike -- searchMahaNearestFeatPt compares them with nothing: the reference's literal)
        cur_pass = dict(P=cfg.p_reg, sigmaSearch=self.sig_pix, maxDist=3 * PIXEL_ERR_VAR, sigmaMerge=self.sig_pix, M=self.d_map.data_ptr(),
                        cov=self.d_cov.data_ptr(), pointFeat=self.d_pf.data_ptr(), slot=o_["slot"].data_ptr(), m=o_["m"].data_ptr(),
                        var=o_["var"].data_ptr(), dist=o_["dist"].data_ptr(), flags=o_["flags"].data_ptr(), mapFlags=self.d_mapflags.data_ptr(),
                        maxDistDynamic=4 * PIXEL_ERR_VAR, list=self.d_curlist.data_ptr())
        passes = [cur_pass]
        if cfg.with_active_search:
            # (diagnostic: the search half of activeMapPointsRegister as rounds 2-4 ran it -- points p_act0 .. + 1536 searched with the
            # active loop's scale, their tables read by nothing)
            self.act_out = dict(slot=z((1536, NA), i32), m=z((1536, NA, 2), f64), var=z((1536, NA, 4), f64), dist=z((1536, NA), f64),
                                flags=z((1536, NA), i32), pf=torch.full((1536, NA), -1, dtype=i32, device=dev))
            a_ = self.act_out
            passes.append(dict(P=1536, sigmaSearch=self.sig(2.5 * PIXEL_ERR_VAR), maxDist=3 * PIXEL_ERR_VAR, sigmaMerge=self.sig_pix,
                               M=self.d_map.data_ptr() + 24 * 1536, cov=self.d_cov.data_ptr() + 72 * 1536, pointFeat=a_["pf"].data_ptr(),
                               slot=a_["slot"].data_ptr(), m=a_["m"].data_ptr(), var=a_["var"].data_ptr(), dist=a_["dist"].data_ptr(),
                               flags=a_["flags"].data_ptr()))
        self.reg_passes = register_passes(passes)
        # the second visits' rounds (cfg.revisit_rounds): their own short list, the same tables (rows of listed points only are rewritten)
        self.RV_CAP = 1024
        self.d_rvlist = torch.full((self.RV_CAP,), -1, dtype=i32, device=dev)
        self.d_rv_visit, self.d_rv_next = z(n_map, i32), z(n_map, i32)
        self.d_rv_reg = [z(n_map, torch.uint8), z(n_map, torch.uint8)]
        self.d_rv_counts, self.d_rv_listcounts = z(4, i32), z(4, i32)
        self.rv_pass = register_passes([dict(cur_pass, P=self.RV_CAP, list=self.d_rvlist.data_ptr())])
        # the fused form (LoopConfig.fused_registration): a list per round, built by the walks themselves
        nr = max(cfg.revisit_rounds, 1)
        self.d_rvlists = torch.full((nr, self.RV_CAP), -1, dtype=i32, device=dev)
        self.d_rvcounts = z(nr + 1, i32)
        self.rv_passes = [register_passes([dict(cur_pass, P=self.RV_CAP, list=self.d_rvlists[r].data_ptr())]) for r in range(nr)]
        # ---- key-frame solves
        # the inter-camera solves of consecutive key frames are independent of each other (each starts from its own frame's poses):
        # key frame k of this rank goes to workspace k mod ic_workers, each with its own worker thread, stream and staging
        self.n_ic_workers = max(cfg.ic_workers, 1)
        if ic is not None:
            self.n_ic_workers = 1
        self.ba_ws = BAWorkspace(device)
        self.ic_wss = [BAWorkspace(device) for _ in range(self.n_ic_workers)]
        self.ic_ws = self.ic_wss[0]
        self.win = self.out = None
        if cfg.with_joint and self.pose_upd is not None:
            self.win = BAWindow(NA, cfg.n_key_frames, N, n_map, device=device)
            self.win.reserve(self.ba_ws)
            # (records in flight: lag windows on the cadence; with the decision placing the key frames, up to one request per frame of the lag)
            self.out = BAOutput(NA, cfg.n_key_frames, n_map, n_slots=8 if not cfg.keyframe_drives else min(self.lag * cfg.key_every + 6, 64), device=device)
            self.out.attach(self.ba_ws)
            if self.d_fref is not None:
                self.out.set_feat_refs(self.d_fref.data_ptr(), self.d_rstat.data_ptr())
            self.recv_rec = z((2, self.out.record_bytes), u8)     # records solved by other ranks arrive here
        self.icam = None
        if ic is None:
            from coslam_amd.ba import BAInterCam, intercam_cams

            # InterCamPoseEstimator::addMapPoints from the live records of ALL cameras (reference src/app/SL_InterCamPoseEstimator.cpp:18-91)
            self.icams = [BAInterCam(NA, N, cfg.pts_stride, n_map, max_dyn=60, device=device) for _ in range(self.n_ic_workers)]
            self.icam = self.icams[0]
            self.ic_cams = intercam_cams([dict(K=self.d_K1.data_ptr(), xy=self.d_xy[g].data_ptr(), state=self.d_state[g].data_ptr(),
                                               slot2map=self.d_slot2map[g].data_ptr(), trackSpan=self.d_trackspan[g].data_ptr(),
                                               isStatic=self.d_isstatic[g].data_ptr()) for g in range(NA)])
        else:
            from coslam_amd.synth import csr_of_problem

            iptr, icam, ixy = csr_of_problem(ic)
            self.ic_ws.upload(ic["Ks"], ic["Rs0"], ic["ts0"], ic["pts0"], iptr, icam, ixy)
            self.d_iR = torch.from_numpy(ic["Rs0"].reshape(-1).copy()).to(dev)
            self.d_iT = torch.from_numpy(ic["ts0"].reshape(-1).copy()).to(dev)
            self.d_iM = torch.from_numpy(ic["pts0"].reshape(-1).copy()).to(dev)
        self.n_pushed = self.n_windows = self.n_key = self.n_my_solves = self.n_my_ic = 0
        self.skip_busy, self.n_skipped = False, 0   # (set by the caller: drop a window request while the previous solve is running)
        self.sequential_registration = bool(cfg.sequential_registration)   # (may be switched between frames)
        self.apply_at, self.my_seq = {}, {}
        self.applied, self.last_apply, self.pushed_frames = 0, None, []
        self.stage_slot, self.h_frames = {}, None
        import os as _os

        self._timing = {} if _os.environ.get("FRAMELOOP_TIMING") else None   # (diagnostic: host seconds per section)
        # (diagnostic, FRAMELOOP_GPU_SECTIONS=1: events on the pose stream at the sections' boundaries -> gpu_sections(): untraced GPU time
        # of every section of a frame, by kind of frame.  Not a valid bench line: two events per section cost a few microseconds each.)
        self._marks = [] if _os.environ.get("FRAMELOOP_GPU_SECTIONS") else None
        self._init_ncc()

    # ---------------------------------------------------------------------------------------------------------------
    def _sec(self, name):
        """diagnostic (FRAMELOOP_TIMING=1): host seconds spent inside a section of the enqueue"""
        import contextlib
        import time as _t

        if self._timing is None:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def cm():
            t0 = _t.perf_counter()
            try:
                yield
            finally:
                self._timing[name] = self._timing.get(name, 0.0) + _t.perf_counter() - t0
        return cm()

    def _mark(self, i, name):
        if self._marks is not None:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record(self.pose_s)
            self._marks.append((i, name, e))

    def gpu_sections(self, first_frame=0):
        """mean GPU microseconds between consecutive marks of a frame on the pose stream, by kind of frame (after drain())"""
        cfg, out, by = self.cfg, {}, {}
        for i, name, e in self._marks or []:
            by.setdefault(i, []).append((name, e))
        for i, ms in by.items():
            if i < first_frame:
                continue
            kind = ("ncc+" if self.ncc is not None and i % cfg.ncc_every == 0 else "") + ("key" if i % 5 == 0 else "plain")
            for (n0, e0), (n1, e1) in zip(ms[:-1], ms[1:]):
                acc = out.setdefault(kind, {}).setdefault(n0 + " -> " + n1, [0.0, 0])
                acc[0] += e0.elapsed_time(e1) * 1e3
                acc[1] += 1
            acc = out.setdefault(kind, {}).setdefault("WHOLE FRAME on the pose stream", [0.0, 0])
            acc[0] += ms[0][1].elapsed_time(ms[-1][1]) * 1e3
            acc[1] += 1
        return {k: {n: round(a / c, 1) for n, (a, c) in v.items()} | {"frames": max(c for _, c in v.values())} for k, v in out.items()}

    def vid(self, i):
        return i % self.T

    def _init_ncc(self):
        """genNewMapPoints -> NewMapPtsNCC (reference src/app/SL_CoSLAM.cpp:1366-1371, src/app/SL_NewMapPointsInterCam.cpp): every
        ncc_every-th frame.  A rank cuts the NCC blocks of its OWN cameras (it holds their images); N > 1: the blocks travel (one
        all-gather of 330 KB per camera); every rank then scores all consecutive camera pairs and turns the matches into new map points
        on its replica -- the same bytes in, the same points out."""
        torch, cfg = self.torch, self.cfg
        self.ncc = None
        NA, N = cfg.n_cams, cfg.n_feat
        if not cfg.with_ncc or NA < 2 or self.pose_upd is None:
            return
        from coslam_amd.ncc import NCC_PAIR_DTYPE, ncc_scaled_dims
        from coslam_amd.newpts import NewPtsJob, newpts_scratch_bytes

        ws_, hs_ = ncc_scaled_dims(cfg.W, cfg.H, 0.3)
        Kinv = np.linalg.inv(self.sc.K)

        dev = self.dev
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)   # noqa: E731
        rec = N * 128 + N * 32 + N * 4          # one camera's record: blocks | abc | valid
        self.ncc = dict(small=z((self.nc, ws_ * hs_), torch.uint8), rec=z((NA, rec), torch.uint8), rec_bytes=rec, runs=0,
                        send=z((self.nc, rec), torch.uint8) if self.world > 1 else None,
                        dF=z((NA - 1, 9), torch.float64),   # the pairs' fundamental matrices, formed per run from the poses just solved
                        pairs=z((NA - 1, cfg.ncc_pair_cap * NCC_PAIR_DTYPE.itemsize), torch.uint8), pair_count=z(NA - 1, torch.int32),
                        group={}, np_scr=z(newpts_scratch_bytes(NA, N), torch.uint8), np_cnt=z(4 + NA, torch.int32), new_total=0)
        r = self.ncc["rec"]
        self.ncc["blk"] = [r[g, :N * 128] for g in range(NA)]
        self.ncc["abc"] = [r[g, N * 128: N * 160] for g in range(NA)]
        self.ncc["valid"] = [r[g, N * 160:] for g in range(NA)]
        cams = [dict(K=self.d_K1.data_ptr(), iK=self.d_iK1.data_ptr(), xy=self.d_xy[g].data_ptr(), state=self.d_state[g].data_ptr(),
                     slot2map=self.d_slot2map[g].data_ptr(), isStatic=self.d_isstatic[g].data_ptr(), reprojErr=self.d_reproj[g].data_ptr())
                for g in range(NA)]
        self.ncc["job"] = NewPtsJob(cams, [self.ncc["pairs"][a].data_ptr() for a in range(NA - 1)],
                                    [self.ncc["pair_count"][a:a + 1].data_ptr() for a in range(NA - 1)])

    def _ncc_leg(self, i, f, dst):
        import coslam_amd
        from coslam_amd._lib import check
        from coslam_amd.ncc import ncc_cams, ncc_epi_pairs_group_dev, ncc_fmats_dev, ncc_get_blocks_group_dev, ncc_pair_jobs
        from coslam_amd.newpts import ncc_candidate_mask_dev, newpts_from_pairs_dev

        cfg, nc, c0, N, NA, ncc = self.cfg, self.nc, self.c0, self.cfg.n_feat, self.cfg.n_cams, self.ncc
        s_ = self.pose_s.cuda_stream
        # NewMapPtsNCC::addSlam's features: this frame's, on tracks of more than three frames, unmapped or on a false point -- of the
        # own cameras, straight into their records' `valid` part
        ncc_candidate_mask_dev(s_, nc, N, self.d_state[c0].data_ptr(), self.d_slot2map[c0].data_ptr(), self.d_trackspan[c0].data_ptr(),
                               self.d_mapflags.data_ptr(), self.n_map, ncc["valid"][c0].data_ptr(), device=self.device,
                               validStride=ncc["rec"].stride(0) // 4)
        if f not in ncc["group"]:
            own = ncc_cams([dict(img=self.img_ptrs[f][k], x=self.d_xy[c0 + k].data_ptr(), y=self.d_xy[c0 + k].data_ptr() + 8 * N,
                                 scaled=ncc["small"][k].data_ptr(), blocks=ncc["blk"][c0 + k].data_ptr(), abc=ncc["abc"][c0 + k].data_ptr(),
                                 valid=ncc["valid"][c0 + k].data_ptr()) for k in range(nc)])
            allc = ncc_cams([dict(img=0, x=self.d_xy[g].data_ptr(), y=self.d_xy[g].data_ptr() + 8 * N, scaled=0, blocks=ncc["blk"][g].data_ptr(),
                                  abc=ncc["abc"][g].data_ptr(), valid=ncc["valid"][g].data_ptr()) for g in range(NA)])
            jobs_ = ncc_pair_jobs([dict(dF=ncc["dF"][a].data_ptr(), camA=a, camB=a + 1, pairs=ncc["pairs"][a].data_ptr(),
                                        count=ncc["pair_count"][a:a + 1].data_ptr()) for a in range(NA - 1)])
            ncc["group"][f] = (own, allc, jobs_)
        own, allc, jobs_ = ncc["group"][f]
        ncc_get_blocks_group_dev(s_, own, cfg.W, cfg.H, N, 0.3, device=self.device)
        if self.world > 1:
            self._gather_ncc_records()
        # E and F of the consecutive camera pairs from the poses this frame has solved (matchBetween, SL_NewMapPointsInterCam.cpp:284-292)
        ncc_fmats_dev(s_, NA, list(range(NA - 1)), list(range(1, NA)), [self.d_iK1.data_ptr()] * NA, self.d_R[dst].data_ptr(),
                      self.d_t[dst].data_ptr(), ncc["dF"].data_ptr(), device=self.device)
        ncc_epi_pairs_group_dev(s_, allc, N, jobs_, 50.0, 0.80, cfg.ncc_pair_cap, device=self.device)   # SL_NewMapPointsInterCam.h:71-72
        # matches -> tracks -> new map points (NewMapPtsNCC::run's tail + output), appended behind d_mapcount
        newpts_from_pairs_dev(s_, ncc["job"], N, cfg.ncc_pair_cap, self.d_R[dst].data_ptr(), self.d_t[dst].data_ptr(), self.d_map.data_ptr(),
                              self.d_cov.data_ptr(), self.d_mapflags.data_ptr(), self.d_newpt.data_ptr(), self.d_firstfrm.data_ptr(),
                              self.d_pf.data_ptr(), self.n_map, self.d_mapcount.data_ptr(), i, ncc["np_scr"].data_ptr(), ncc["np_cnt"].data_ptr(),
                              maxDisp=80.0, maxRpErr=3.0, pixelErrVar=self.sig_pix, minLen=2, device=self.device, W=cfg.W, H=cfg.H)
        ncc["runs"] += 1

    def _gather_ncc_records(self):
        """the own cameras' NCC records (blocks | abc | candidate mask) to every rank: ONE all-gather, 330 KB per camera, every 4th frame"""
        import ctypes as C_

        import coslam_amd
        from coslam_amd._lib import check

        torch, ncc, nc, c0 = self.torch, self.ncc, self.nc, self.c0
        L, vp, ps = coslam_amd.lib(), C_.c_void_p, self.pose_s.cuda_stream
        with torch.cuda.stream(self.pose_s):
            ncc["send"].copy_(ncc["rec"][c0:c0 + nc], non_blocking=True)
        if self.native is not None:
            L.cs_comm_allgather_dev.argtypes = [vp, vp, vp, vp, C_.c_size_t]
            check(L.cs_comm_allgather_dev(self.native.exchange_comm, vp(ps), vp(ncc["send"].data_ptr()), vp(ncc["rec"].data_ptr()),
                                          ncc["send"].numel()), "cs_comm_allgather_dev")
        else:
            import torch.distributed as dist

            with torch.cuda.stream(self.pose_s):
                dist.all_gather_into_tensor(ncc["rec"].view(-1), ncc["send"].view(-1))

    # ---------------------------------------------------------------------------------------------------------------
    def first_frame(self):
        """GPUKLT::first (reference src/tracking/GPUKLT.cpp:133-142) of every camera, the slot -> map-point association that stands
        in for the map initialisation (out of scope), and the first hand-back."""
        torch, cfg = self.torch, self.cfg
        import coslam_amd
        from coslam_amd.handback import handback_dev

        NA, N = cfg.n_cams, cfg.n_feat
        self.grp.detect_dev(self.img_ptrs[0], self.dest_ptrs[0], self.cnt_ptrs)
        self.grp.advanceFrame()
        self.grp.synchronize()
        if self.world > 1:
            with torch.cuda.stream(self.pose_s):
                self.xchg.pack_group(self.d_dests[0], self.d_R[0][self.c0:self.c0 + self.nc], self.d_t[0][self.c0:self.c0 + self.nc], self.pose_s)
                self.xchg.all_gather(self.pose_s)
            torch.cuda.synchronize()
        dests = []
        for g in range(NA):
            if g in self.my_cams:
                w = self.d_dests[0][g - self.c0].cpu().numpy()
            else:
                w = self.xchg.unpack(g, device=self.device)[0].cpu().numpy().reshape(-1)
            dests.append(np.ascontiguousarray(w, dtype=np.int32).view(coslam_amd.KLT_TrackedFeature))
        s2m = np.stack([self.associate(self.sc, g, 0, dests[g]) for g in range(NA)])
        self.d_slot2map.copy_(torch.from_numpy(s2m))
        with torch.cuda.stream(self.pose_s):
            self._handback(0, 0)
        torch.cuda.synchronize()
        self.d_slot2map.copy_(torch.from_numpy(s2m))   # the first hand-back starts every track as new (unmapped): put the map back
        torch.cuda.synchronize()
        if self.pose_upd is not None:
            # frame 0 into the history as well (its pixels and poses: the first term of every track born in it, which a whole-track
            # mergability walk ends with); the dynamic test has nothing to say about one-frame tracks
            self.pose_upd.detect_dynamic_dev(self.pose_s.cuda_stream, self.pu_args, self.d_R[0].data_ptr(), self.d_t[0].data_ptr(), self.n_map,
                                             self.d_mapflags.data_ptr(), 0, 20, 5, 3, MAX_EPI_ERR)
            torch.cuda.synchronize()
        if cfg.keyframe_decision or cfg.intracam_mapping or cfg.keyframe_drives:
            self.enable_keyframe_decision(0, 0)

    def enable_keyframe_decision(self, frame, b):
        """the key-pose state as CoSLAM::initMap leaves it (reference src/app/SL_CoSLAM.cpp:246-256, :278-291) -- or as a key frame added at
        `frame` would: a key pose with self motion in every camera (pose buffer b), nMappedPts = the certainly static mapped features of that
        frame, m_minCamTranslation = the mean distance between the cameras / 4.5.  From the next step on cs_keyframe_ready_dev runs per frame.
        Synchronises."""
        from coslam_amd.keyframe import keyframe_cams

        torch, NA = self.torch, self.cfg.n_cams
        torch.cuda.synchronize()
        i32, f64 = torch.int32, torch.float64
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.dev)   # noqa: E731
        st, fl, s2m = self.d_state.cpu().numpy(), self.d_mapflags.cpu().numpy(), self.d_slot2map.cpu().numpy()
        km = [int((((st[g] == 0) | (st[g] == 1)) & (s2m[g] >= 0) & ((fl[np.clip(s2m[g], 0, len(fl) - 1)] & 7) == 0)).sum()) for g in range(NA)]
        self.kf = dict(frame=torch.full((NA,), int(frame), dtype=i32, device=self.dev), mapped=torch.tensor(km, dtype=i32, device=self.dev),
                       selfR=self.d_R[b].clone(), selfT=self.d_t[b].clone(), ready=z(NA + 2, i32), cnt=z(2 * NA, i32), cen=z((NA, 3), f64),
                       stats=z(5, i32), frames=0, placed=[], not_applied=0)
        Rn, tn = self.d_R[b].cpu().numpy().reshape(NA, 3, 3), self.d_t[b].cpu().numpy()
        cen = np.stack([-Rn[g].T @ tn[g] for g in range(NA)])
        dist = [np.linalg.norm(cen[a] - cen[c]) for a in range(NA) for c in range(a + 1, NA)]
        self.kf["min_translation"] = float(np.mean(dist) / 4.5) if dist else 0.1
        self.kf["cams"] = [keyframe_cams([dict(state=self.d_state[g].data_ptr(), slot2map=self.d_slot2map[g].data_ptr(),
                                               R=self.d_R[q][g].data_ptr(), t=self.d_t[q][g].data_ptr(), keyFrame=self.kf["frame"][g:].data_ptr(),
                                               keyMapped=self.kf["mapped"][g:].data_ptr(), selfR=self.kf["selfR"][g].data_ptr(),
                                               selfT=self.kf["selfT"][g].data_ptr()) for g in range(NA)]) for q in range(2)]
        torch.cuda.synchronize()

    def _kf_lag_ring(self):
        """the ring of D + 1 snapshots a lagged key-frame decision acts on: per slot the hand-back's xy / state / slot2map of all cameras, the
        poses, the decision word in pinned host memory and an event behind it"""
        k, torch, cfg, NA, N = self.kf, self.torch, self.cfg, self.cfg.n_cams, self.cfg.n_feat
        if "lag_ring" in k:
            return k["lag_ring"]
        from coslam_amd.handback import handback_cams

        ring = []
        for _ in range(cfg.keyframe_lag + 1):
            xy = torch.zeros((NA, 2 * N), dtype=torch.float64, device=self.dev)
            st = torch.zeros((NA, N), dtype=torch.int32, device=self.dev)
            s2m = torch.zeros((NA, N), dtype=torch.int32, device=self.dev)
            R, t = torch.zeros((NA, 9), dtype=torch.float64, device=self.dev), torch.zeros((NA, 3), dtype=torch.float64, device=self.dev)
            hb = handback_cams([dict(dest=self.d_dests[0][0].data_ptr(), K=self.d_K1.data_ptr(), kud=self.d_kud.data_ptr(), mapPts=self.d_map.data_ptr(),
                                     slot2map=s2m[g].data_ptr(), trackSpan=self.d_trackspan[g].data_ptr(), xy=xy[g].data_ptr(), state=st[g].data_ptr(),
                                     Ms=self.d_Ms[g].data_ptr(), ms=self.d_ms[g].data_ptr(), sel=self.d_sel[g].data_ptr(),
                                     npts=self.d_npts[g:g + 1].data_ptr(), opt=self.d_opt[g].data_ptr(), pointFeat=0, pointFeatStride=NA, nPointFeat=0,
                                     isStatic=0) for g in range(NA)])
            ring.append(dict(xy=xy, st=st, s2m=s2m, R=R, t=t, hb=hb, word=torch.zeros(1, dtype=torch.int32).pin_memory(),
                             ev=torch.cuda.Event(), frame=-1))
        torch.cuda.synchronize()   # (zero-filled on torch's stream, used on the pose stream)
        k["lag_ring"], k["lag_blocked"] = ring, 0
        return ring

    def _kf_lag_record(self, i, dst):
        """frame i's decision word, records and poses into its ring slot, on the pose stream; no host wait"""
        k, torch, NA = self.kf, self.torch, self.cfg.n_cams
        sl = self._kf_lag_ring()[i % (self.cfg.keyframe_lag + 1)]
        from coslam_amd.keyframe import keyframe_snapshot_dev

        keyframe_snapshot_dev(self.pose_s.cuda_stream, NA, self.cfg.n_feat, self.d_xy.data_ptr(), self.d_state.data_ptr(), self.d_slot2map.data_ptr(),
                              self.d_R[dst].data_ptr(), self.d_t[dst].data_ptr(), k["ready"][NA + 1:].data_ptr(), sl["xy"].data_ptr(), sl["st"].data_ptr(),
                              sl["s2m"].data_ptr(), sl["R"].data_ptr(), sl["t"].data_ptr(), sl["word"].data_ptr(), device=self.device)
        sl["ev"].record(self.pose_s)
        sl["frame"] = i

    def _kf_lag_act(self, i, dst):
        """the decision of frame i - D, read from pinned memory behind that frame's event: a key frame -> the push of THAT frame's records and
        poses and the window's request (genNewMapPoints :1331-1346), D frames late"""
        k, D = self.kf, self.cfg.keyframe_lag
        f = i - D
        sl = self._kf_lag_ring()[f % (D + 1)]
        if f < 1 or sl["frame"] != f:
            return
        if not sl["ev"].query():
            k["lag_blocked"] += 1     # (the host caught up with the device: it waits for a frame D behind, the device has D frames queued)
            sl["ev"].synchronize()
        if int(sl["word"][0]) == 0:
            return
        k["placed"].append(f)
        self._key_frame(f, dst, snap=sl)

    def keyframe_stats(self):
        """what the key-frame decision said over the frames it ran on (cs_keyframe_ready_dev's d_stats; a synchronous read)"""
        if not getattr(self, "kf", None):
            return None
        a = self.kf["stats"].cpu().tolist()
        im = None if "im_total" not in self.kf else dict(zip(("candidates_tried", "points_added", "points_dropped_map_full"), self.kf["im_total"].cpu().tolist()))
        return dict(frames=self.kf["frames"], intracam_new_map_points=im, frames_with_a_camera_ready=a[0], frames_with_decrease_ie_key_frames_added=a[1],
                    cameras_saying_decrease=a[2], cameras_saying_view_angle=a[3], cameras_saying_translation=a[4],
                    min_cam_translation=self.kf["min_translation"], key_frames_placed_by_the_decision=list(self.kf["placed"]),
                    windows_not_applied_history_too_short=self.kf["not_applied"], last_key_frame_per_camera=self.kf["frame"].cpu().tolist(),
                    decision_lag_frames=self.cfg.keyframe_lag, host_waits_that_blocked=self.kf.get("lag_blocked"),
                    mapped_static_at_the_last_key_frame=self.kf["mapped"].cpu().tolist())

    def _handback(self, b, frame, which="all"):
        from coslam_amd.handback import handback_dev

        cfg = self.cfg
        a = dict(N=cfg.n_feat, W=cfg.W, H=cfg.H, nColBlk=cfg.n_col_blk, nRowBlk=cfg.n_row_blk, ptsStride=cfg.pts_stride, device=self.device,
                 frame=frame)
        if which in ("all", "own"):
            handback_dev(self.pose_s.cuda_stream, self.hb_own[b], **a)
        if which in ("all", "other") and self.hb_other is not None:
            handback_dev(self.pose_s.cuda_stream, self.hb_other, **a)

    def _apply_due(self, i, src):
        """RobustBundleRTS::output() of the window whose lag ends at this frame, before anything of frame i touches the map"""
        due = self.apply_at.pop(i, None)
        if due is None:
            return
        k, owner, first_key, seq = due
        frames = None
        if isinstance(first_key, (list, tuple)):   # (key frames the decision placed: a list instead of first + j * key_every)
            frames, first_key = list(first_key), first_key[0]
            if (i - 1) - first_key + 1 > self.pose_upd.storeLen:   # (the frames the history KEEPS: hist_store, not the walks' depth)
                # the camera graphs would start behind the pose history's oldest frame: the record is consumed, nothing is written back
                self.my_seq.pop(k, None)
                self.kf["not_applied"] += 1
                return
        if owner == self.rank:
            # the pose stream waits ON THE DEVICE for this rank's worker to publish the record: the host goes on enqueueing frames
            self.my_seq.pop(k)
            rec = self.out.wait_dev(seq, self.pose_s.cuda_stream) if self.cfg.device_wait else self.out.wait(seq)
        else:
            rec = self.recv_rec[k & 1].data_ptr()
        if self.world > 1:
            self.xchg.broadcast(rec, self.out.record_bytes, owner, self.device, self.pose_s)
        if frames is not None:
            self.out.apply_frames_dev(rec, self.pose_s.cuda_stream, self.pose_upd, self.win, self.pu_args, self.d_pf.data_ptr(), self.n_map,
                                      self.d_map.data_ptr(), self.d_cov.data_ptr(), self.d_mapflags.data_ptr(), self.sig_pix, frames,
                                      self.d_R[src].data_ptr(), self.d_t[src].data_ptr(), self.d_apply_counts.data_ptr(), seq=seq)
        else:
            self.out.apply_dev(rec, self.pose_s.cuda_stream, self.pose_upd, self.win, self.pu_args, self.d_pf.data_ptr(), self.n_map,
                               self.d_map.data_ptr(), self.d_cov.data_ptr(), self.d_mapflags.data_ptr(), self.sig_pix, first_key,
                               self.cfg.key_every, self.d_R[src].data_ptr(), self.d_t[src].data_ptr(), self.d_apply_counts.data_ptr(), seq=seq)
        self.applied += 1
        self.last_apply = dict(window=k, solved_by_rank=owner, first_key_frame=first_key, applied_at_frame=i)

    def stage(self, i):
        f = self.vid(i)
        self.stage_slot[i] = self.grp.stage_h([self.h_frames[f][c].data_ptr() for c in range(self.nc)])

    def step(self, i, key_frame, upload=False):
        from coslam_amd.pose import intraCamEstimate_batch_dev
        from coslam_amd.register import register_search_passes_dev

        torch, cfg = self.torch, self.cfg
        klt_s, pose_s, c0, nc, NA_ = self.klt_s, self.pose_s, self.c0, self.nc, self.cfg.n_cams
        self._frame_in_step = i
        f, fn = self.vid(i), self.vid(i + 1)
        b = i & 1
        if i >= 2:
            klt_s.wait_event(self.dest_free[b])      # the consumer of this dest buffer two frames ago is done
            if cfg.klt_after_intracam:
                klt_s.wait_event(self.intracam_done[(i - 1) & 1])   # ... and the previous frame's pose solve has the chip to itself
        if upload:
            self.stage(i + 2)
            cur, nxt = self.grp.staged(self.stage_slot.pop(i)), self.grp.staged(self.stage_slot[i + 1])
        else:
            cur, nxt = self.img_ptrs[f], self.img_ptrs[fn]
        if cfg.prefetch:   # this frame's detector tail also builds the next frame's pyramids + cornerness maps
            self.grp.prefetch_dev(nxt)
        self.grp.redetect_dev(cur, self.dest_ptrs[b], self.cnt_ptrs)
        self.grp.advanceFrame()
        self.klt_done[b].record(klt_s)
        pose_s.wait_event(self.klt_done[b])          # pose(f) consumes what the tracker produced for frame f
        src, dst = (i + 1) & 1, i & 1
        ps = pose_s.cuda_stream
        self._mark(i, "tracker done")
        if self.out is not None:
            if self._timing is not None:
                import time as _t

                t0 = _t.perf_counter()
                self._apply_due(i, src)
                self._timing["apply"] = self._timing.get("apply", 0.0) + _t.perf_counter() - t0
            else:
                self._apply_due(i, src)
        self._mark(i, "BA applied")
        self._handback(b, i, "own")
        self._mark(i, "hand-back")
        intraCamEstimate_batch_dev(ps, nc, cfg.pts_stride, self.d_K.data_ptr(), self.d_R[src].data_ptr() + 72 * c0,
                                   self.d_t[src].data_ptr() + 24 * c0, self.d_npts.data_ptr() + 4 * c0, 0,
                                   self.d_Ms.data_ptr() + 24 * cfg.pts_stride * c0, self.d_ms.data_ptr() + 16 * cfg.pts_stride * c0, 10.0,
                                   self.d_R[dst].data_ptr() + 72 * c0, self.d_t[dst].data_ptr() + 24 * c0, self.d_opt.data_ptr() + 96 * c0,
                                   self.d_ok.data_ptr() + 4 * c0, device=self.device)
        if cfg.klt_after_intracam:
            self.intracam_done[b].record(pose_s)
        self._mark(i, "intracam")
        if self.world > 1:
            # the merge step: every camera's {dest[], R, t} to every rank, then the other ranks' cameras through the same hand-back
            with torch.cuda.stream(pose_s):
                self.xchg.pack_group(self.d_dests[b], self.d_R[dst][c0:c0 + nc], self.d_t[dst][c0:c0 + nc], pose_s)
                self.xchg.all_gather(pose_s)
                self.xchg.unpack_poses(self.d_R[dst], self.d_t[dst], pose_s, skip_own=True)
            self._handback(b, i, "other")
        if self.pose_upd is not None:
            # parallelPoseUpdate(false): gate 2.0, sigma = PIXEL_ERR_VAR; detectDynamicFeaturePoints(20, 5, 3, MAX_EPI_ERR)
            if cfg.with_classify:   # CoSLAM::poseUpdate as a whole: two launches (the gate also builds the classification's worklist)
                self.pose_upd.pose_update_classify_frame_dev(ps, self.pu_args, self.d_pf.data_ptr(), self.n_map, self.d_R[dst].data_ptr(),
                                                             self.d_t[dst].data_ptr(), self.d_map.data_ptr(), self.d_cov.data_ptr(),
                                                             self.d_mapflags.data_ptr(), 0, self.sig_pix, i, self.d_newpt.data_ptr(),
                                                             self.d_sfn.data_ptr(), self.d_firstfrm.data_ptr(), self.sig(12.0), 20, 5, 3,
                                                             MAX_EPI_ERR, d_counts=self.d_cls_counts.data_ptr())
            else:
                self.pose_upd.pose_update_frame_dev(ps, self.pu_args, self.d_pf.data_ptr(), self.n_map, self.d_R[dst].data_ptr(),
                                                    self.d_t[dst].data_ptr(), self.d_map.data_ptr(), self.d_cov.data_ptr(),
                                                    self.d_mapflags.data_ptr(), 0, self.sig_pix, i, 20, 5, 3, MAX_EPI_ERR)
        self._mark(i, "pose update + classify")
        # the reference's order of a frame (src/gui/CoSLAMThread.cpp:104-118): poseUpdate (with mapPointsClassify) -> activeMapPointsRegister ->
        # genNewMapPoints -> currentMapPointsRegister: the new map points take their features BEFORE the current points' registration
        # looks at them (a feature that carries a point ends a registration walk)
        if getattr(self, "kf", None):
            # genNewMapPoints' first half (:1294-1346): is a camera ready for a key frame, and addKeyFrame's bookkeeping when one's mapped
            # points have decreased
            from coslam_amd.keyframe import keyframe_ready_dev

            k = self.kf
            keyframe_ready_dev(ps, k["cams"][dst], cfg.n_feat, self.n_map, self.d_map.data_ptr(), self.d_mapflags.data_ptr(), self.d_firstfrm.data_ptr(),
                               i, k["min_translation"], k["ready"].data_ptr(), k["cnt"].data_ptr(), k["cen"].data_ptr(), ratio=cfg.keyframe_ratio,
                               addKeyFrame=True, d_stats=k["stats"].data_ptr(), device=self.device)
            k["frames"] += 1
            if cfg.keyframe_drives and cfg.keyframe_lag > 0:
                # no wait: this frame's `decrease`, records and poses go into slot i % (D + 1) of a ring (the word into pinned host memory, an
                # event behind it); what the host acts on below is the decision of frame i - D
                key_frame = False   # (recorded at the end of the frame, where the push of a key frame sits: behind the registration's attachments)
            elif cfg.keyframe_drives:
                # `decrease` (ready[nCams + 1]) back to the host: the frame is a key frame for ALL cameras when one camera's mapped points
                # have decreased (:1331-1346) -- the push and the request below follow the device's answer, not the caller's cadence
                pose_s.synchronize()
                key_frame = bool(int(k["ready"][NA_ + 1].item()))
                if key_frame:
                    k["placed"].append(i)
            if cfg.intracam_mapping and self.pose_upd is not None:
                # ... and for the cameras that are ready (view angle / translation; with `decrease` the cameras that said so too: :1310-1346)
                # SingleSLAM::newMapPoints: the unmapped features on tracks of nMinFeatTrkLen = 20 frames, each from its own track
                if "im_scr" not in k:
                    torch = self.torch
                    k["im_scr"] = torch.zeros(self.pose_upd.newpts_intracam_scratch_bytes(), dtype=torch.uint8, device=self.dev)
                    k["im_cnt"] = torch.zeros(3, dtype=torch.int32, device=self.dev)
                    k["im_total"] = torch.zeros(3, dtype=torch.int64, device=self.dev)
                    torch.cuda.synchronize()
                self.pose_upd.newpts_intracam_dev(ps, self.pu_args, self.d_map.data_ptr(), self.d_cov.data_ptr(), self.d_mapflags.data_ptr(),
                                                  self.d_newpt.data_ptr(), self.d_firstfrm.data_ptr(), self.d_pf.data_ptr(), self.n_map,
                                                  self.d_mapcount.data_ptr(), k["im_scr"].data_ptr(), self.sig_pix, d_ready=k["ready"].data_ptr(),
                                                  readyMin=1, minTrackLen=20, maxWalk=1024, maxEpiErr=2.0, d_counts=k["im_cnt"].data_ptr())
                with self.torch.cuda.stream(self.pose_s):
                    k["im_total"] += k["im_cnt"]
        if self.ncc is not None and i % cfg.ncc_every == 0:
            self._ncc_leg(i, f, dst)
            self._mark(i, "ncc leg")
        if cfg.with_register:
            from coslam_amd.register import register_list_current_dev

            # curMapPts of this frame as a list (mapStateUpdate, SL_CoSLAM.cpp:1176-1194); the rows of every other point lose their candidates
            register_list_current_dev(ps, NA_, self.n_map, self.d_mapcount.data_ptr(), self.d_pf.data_ptr(), self.d_mapflags.data_ptr(),
                                      self.d_curlist.data_ptr(), self.d_curcount.data_ptr(), self.reg_out["slot"].data_ptr(), device=self.device,
                                      listCap=cfg.p_reg, d_overflow=self.d_curoverflow.data_ptr())
            register_search_passes_dev(ps, self.reg_args[dst], cfg.n_feat, cfg.W, cfg.H, self.reg_passes, device=self.device, cam0=c0,
                                       nCamsRun=nc)
            self._mark(i, "list + search")
            if self.pose_upd is not None and cfg.with_mergability:
                self._mergability(ps)
                self._mark(i, "mergability")
        self._dst_now, self._frame_now = dst, i
        if cfg.with_register and cfg.with_decide and self.pose_upd is not None and cfg.with_mergability:
            self._decide(ps)
        elif self.pose_upd is not None:
            self._advance_refs(ps)   # (no decisions in this configuration: the references still follow the tracks, every frame)
        # the tracker of frame i + 2 (it writes this dest buffer) is released HERE, at the end of the frame's pose work, although the
        # buffer's last reader was the hand-back: released earlier the tracker runs two frames ahead and under more of the pose stream's
        # kernels -- measured 1978-1986 (behind the hand-back) / 1928-1934 (behind the gate) / 1894-1903 (behind the classification)
        # against 2173-2193 frames/s here, 2119-2123 behind the key-frame requests (profiles/r04_ab_runs.txt)
        self._mark(i, "decide + refine")
        self.dest_free[b].record(pose_s)
        if cfg.keyframe_drives and cfg.keyframe_lag > 0 and getattr(self, "kf", None):
            self._kf_lag_record(i, dst)
            self._kf_lag_act(i, dst)
        if key_frame:
            if self._timing is not None:
                import time as _t

                t0 = _t.perf_counter()
                self._key_frame(i, dst)
                self._timing["key_frame"] = self._timing.get("key_frame", 0.0) + _t.perf_counter() - t0
            else:
                self._key_frame(i, dst)
            self._mark(i, "key-frame push + requests")

    def _mergability(self, ps):
        """staticCheckMergability of every candidate of the current points' pass over its WHOLE track (SL_CoSLAM.cpp:714-729, :768), as a
        running verdict: the newest `hist` frames walked as they stand, the older ones' verdict cached per (point, camera) and extended
        by one term per frame (cs_register_mergability_running_dev) -- own cameras' columns"""
        cfg = self.cfg
        self.pose_upd.register_mergability_running_dev(ps, self.pu_args, self.n_map, self.d_map.data_ptr(), self.d_cov.data_ptr(),
                                                       self.reg_out["slot"].data_ptr(), self.sig_pix, self.d_merge_cache.data_ptr(),
                                                       self.d_mergeable.data_ptr(), tolPix=cfg.merge_tol_pix, d_counts=self.d_merge_counts.data_ptr(),
                                                       cam0=self.c0, nCamsRun=self.nc, d_list=self.d_curlist.data_ptr(), nList=cfg.p_reg,
                                                       d_flags=self.reg_out["flags"].data_ptr())
        if cfg.merge_tol_pix > 0 and cfg.verdict_check_every > 0 and self._frame_now_for_check() % cfg.verdict_check_every == 0:
            # diagnostic: how often does a CACHED tail change a verdict?  Every n-th frame the same candidates are judged once more
            # with tolPix = 0 -- every tail whose point has moved at all walked again with today's point -- on a copy of the cache, into a
            # scratch table; the two tables are compared where a candidate was judged.  Two launches + a 6 MB copy on those frames.
            torch = self.torch
            with torch.cuda.stream(self.pose_s):
                if not hasattr(self, "_vc"):
                    self._vc = dict(cache=torch.empty_like(self.d_merge_cache), out=torch.empty_like(self.d_mergeable),
                                    n=torch.zeros(3, dtype=torch.int64, device=self.dev))
                V = self._vc
                V["cache"].copy_(self.d_merge_cache, non_blocking=True)
                V["out"].copy_(self.d_mergeable, non_blocking=True)
            self.pose_upd.register_mergability_running_dev(ps, self.pu_args, self.n_map, self.d_map.data_ptr(), self.d_cov.data_ptr(),
                                                           self.reg_out["slot"].data_ptr(), self.sig_pix, V["cache"].data_ptr(), V["out"].data_ptr(),
                                                           tolPix=0.0, d_counts=None, cam0=self.c0, nCamsRun=self.nc, d_list=self.d_curlist.data_ptr(),
                                                           nList=cfg.p_reg, d_flags=self.reg_out["flags"].data_ptr())
            with torch.cuda.stream(self.pose_s):
                lc = slice(self.c0, self.c0 + self.nc)
                judged = (self.reg_out["slot"][:, lc] >= 0) & ((self.d_mergeable[:, lc] == 0) | (self.d_mergeable[:, lc] == 1))
                V["n"] += torch.stack([judged.sum(), (judged & (self.d_mergeable[:, lc] != V["out"][:, lc])).sum(),
                                       torch.ones((), dtype=torch.int64, device=self.dev)])

    def _frame_now_for_check(self):
        return getattr(self, "_frame_in_step", 0)

    def verdict_check(self):
        """{candidates judged, verdicts the cached tails changed, frames checked} of the running mergability verdict's diagnostic"""
        if not hasattr(self, "_vc"):
            return None
        a = self._vc["n"].cpu().tolist()
        return {"candidates_judged": a[0], "verdicts_changed_by_the_cache": a[1], "frames_checked": a[2]}

    def _decide(self, ps):
        """currentMapPointsRegister's decisions -- curStaticPointsRegInGroup (reference src/app/SL_CoSLAM.cpp:854-898, 731-830, bMerge ==
        false) and behind it curDynamicPointsRegInGroup (:904-1020) on the certainly dynamic points, one call -- over the search
        tables of ALL cameras, then refineMapPoint of the points that gained a feature (:889-893, :666-713).  N > 1: the own cameras'
        columns of the tables travel first (one small all-gather), every rank then takes the same decisions on its replica."""
        from coslam_amd.register import register_decide_scratch_bytes, register_decide_static_dev

        cfg, NA = self.cfg, self.cfg.n_cams
        if not hasattr(self, "_dec"):
            torch = self.torch
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.dev)   # noqa: E731
            self.n_merge_frames = 0
            self._dec = dict(att=z((self.n_map, NA), torch.uint8), reg=z(self.n_map, torch.uint8), cnt=z(4, torch.int32), ref_cnt=z(1, torch.int32),
                             mcnt=z(4, torch.int32),
                             scr=z(register_decide_scratch_bytes(NA, cfg.n_feat, self.n_map), torch.uint8), s2m=None,
                             mscr=z(self.pose_upd.decide_merge_scratch_bytes(self.n_map, cfg.p_reg), torch.uint8))
            torch.cuda.synchronize()   # (the zero fills ran on torch's stream: done before the pose stream touches the buffers)
        D = self._dec
        if self.sequential_registration:
            from coslam_amd.register import register_cur_static_sequential_dev

            if self.world > 1:
                raise RuntimeError("sequential_registration: one rank only (the per-loop tables are not exchanged)")
            if not hasattr(self, "_pass_current"):
                T_ = type(self.reg_passes[0])
                self._pass_current = (T_ * 1)(self.reg_passes[0])
            o = self.reg_out
            D["s2m"] = register_cur_static_sequential_dev(ps, self.pose_upd, self.pu_args, self.reg_args[self._dst_now], cfg.n_feat, cfg.W, cfg.H,
                                                          self._pass_current, self.n_map, o["slot"].data_ptr(), o["flags"].data_ptr(),
                                                          self.d_mergeable.data_ptr(), self.d_mapflags.data_ptr(), self.d_pf.data_ptr(),
                                                          D["s2m"] if D["s2m"] is not None else [self.d_slot2map[g].data_ptr() for g in range(NA)],
                                                          D["att"].data_ptr(), D["reg"].data_ptr(), D["scr"].data_ptr(), self.d_map.data_ptr(),
                                                          self.d_cov.data_ptr(), self.sig_pix, d_counts=D["cnt"].data_ptr(), device=self.device,
                                                          with_dynamic=True, merge=(cfg.merge_every > 0 and self._frame_now % cfg.merge_every == 0),   # CoSLAMThread.cpp:117-118
                                                          d_merge_scratch=D["mscr"].data_ptr(), mergability=self._mergability, n_sweeps=0)
            self._advance_refs(ps)   # (the sequential mode refines inside its camera loops over this frame's features; the references follow)
            return
        if self.world > 1:
            self._gather_candidates()
        kinds = 3
        if cfg.merge_every > 0 and self._frame_now % cfg.merge_every == 0:
            # a bMerge frame: the static points' walks one after the other with checkUnify at a conflict, the dynamic points' behind them
            o = self.reg_out
            self.pose_upd.register_decide_merge_dev(ps, self.pu_args, self.n_map, 0, o["slot"].data_ptr(), o["flags"].data_ptr(),
                                                    self.d_mergeable.data_ptr(), self.d_mapflags.data_ptr(), self.d_pf.data_ptr(),
                                                    self.d_map.data_ptr(), self.d_cov.data_ptr(), self.sig_pix, D["att"].data_ptr(),
                                                    D["reg"].data_ptr(), D["mscr"].data_ptr(), D["mcnt"].data_ptr(),
                                                    d_list=self.d_curlist.data_ptr(), nList=cfg.p_reg)
            self._refine(ps, D["reg"].data_ptr())
            self.n_merge_frames += 1
            kinds = 2
        if cfg.fused_registration and kinds == 3 and self.d_fref is not None:
            self._decide_fused(ps, D)
            return
        D["s2m"] = register_decide_static_dev(ps, NA, cfg.n_feat, self.n_map, 0, self.reg_out["slot"].data_ptr(), self.reg_out["flags"].data_ptr(),
                                              self.d_mergeable.data_ptr(), self.d_mapflags.data_ptr(), self.d_pf.data_ptr(),
                                              D["s2m"] if D["s2m"] is not None else [self.d_slot2map[g].data_ptr() for g in range(NA)],
                                              D["att"].data_ptr(), D["reg"].data_ptr(), D["scr"].data_ptr(), D["cnt"].data_ptr(), device=self.device,
                                              kinds=kinds, n_sweeps=0)   # (0: ONE launch that sweeps until the owners have settled)   # curStaticPointsRegInGroup and curDynamicPointsRegInGroup (currentMapPointsRegister, :834-853)
        self._refine(ps, D["reg"].data_ptr())
        if cfg.revisit_rounds > 0 and kinds == 3:
            self._revisit_rounds(ps, D)

    def _decide_fused(self, ps, D):
        """the single pass, its refine and the second visits' rounds with the launches fused (tools/cxx/frame_loop.cpp's sequence): 2 + 4 per
        round instead of 3 + 6; the same map, tables and poses"""
        from coslam_amd.register import register_decide_kinds_rounds_dev, register_revisit_decide_next_dev, register_search_passes_dev

        cfg, NA, R = self.cfg, self.cfg.n_cams, self.cfg.revisit_rounds
        o = self.reg_out
        s2m = D["s2m"] if D["s2m"] is not None else [self.d_slot2map[g].data_ptr() for g in range(NA)]
        D["s2m"] = register_decide_kinds_rounds_dev(ps, NA, cfg.n_feat, self.n_map, 0, o["slot"].data_ptr(), o["flags"].data_ptr(), self.d_mergeable.data_ptr(),
                                                    self.d_mapflags.data_ptr(), self.d_pf.data_ptr(), s2m, D["att"].data_ptr(), D["reg"].data_ptr(),
                                                    D["scr"].data_ptr(), self.d_rvlists.data_ptr() if R > 0 else 0, self.RV_CAP, R, self.d_rvcounts.data_ptr(),
                                                    self.d_rv_visit.data_ptr(), self.d_rv_next.data_ptr(), d_counts=D["cnt"].data_ptr(), device=self.device)
        adv = lambda lst, n, all_, sel, clr: self.pose_upd.feat_ref_advance_refine_dev(   # noqa: E731
            ps, self.pu_args, self.n_map, self.d_pf.data_ptr(), self._frame_now, self.d_fref.data_ptr(), self.d_rstat.data_ptr(), lst, n, all_, sel, clr,
            self.d_map.data_ptr(), self.d_cov.data_ptr(), self.sig_pix, d_counts=self.d_fref_counts.data_ptr())
        adv(self.d_curlist.data_ptr(), cfg.p_reg, True, D["reg"].data_ptr(), False)
        for r in range(R):
            lst = self.d_rvlists[r].data_ptr()
            register_search_passes_dev(ps, self.reg_args[self._dst_now], cfg.n_feat, cfg.W, cfg.H, self.rv_passes[r], device=self.device)
            self.pose_upd.register_mergability_running_dev(ps, self.pu_args, self.n_map, self.d_map.data_ptr(), self.d_cov.data_ptr(), o["slot"].data_ptr(),
                                                           self.sig_pix, self.d_merge_cache.data_ptr(), self.d_mergeable.data_ptr(), tolPix=0.0, d_counts=0,
                                                           cam0=0, nCamsRun=NA, d_list=lst, nList=self.RV_CAP, d_flags=o["flags"].data_ptr())
            more = r + 1 < R
            register_revisit_decide_next_dev(ps, NA, cfg.n_feat, self.n_map, self.RV_CAP, 0, 3, lst, self.d_rv_next.data_ptr(), self.d_rv_visit.data_ptr(),
                                             o["slot"].data_ptr(), o["flags"].data_ptr(), self.d_mergeable.data_ptr(), self.d_mapflags.data_ptr(),
                                             self.d_pf.data_ptr(), D["s2m"], D["att"].data_ptr(), self.d_rv_reg[0].data_ptr(), D["scr"].data_ptr(),
                                             self.d_curlist.data_ptr(), self.d_curcount.data_ptr(), cfg.p_reg, self.d_rv_counts.data_ptr(), device=self.device,
                                             d_listCount=self.d_rvcounts[r:].data_ptr(), d_nextList=self.d_rvlists[r + 1].data_ptr() if more else 0,
                                             d_nextCount=self.d_rvcounts[r + 1:].data_ptr() if more else 0, d_overflow=self.d_rvcounts[R:].data_ptr())
            adv(lst, self.RV_CAP, False, self.d_rv_reg[0].data_ptr(), True)

    def _revisit_rounds(self, ps, D):
        """The reference's SECOND VISITS (src/app/SL_CoSLAM.cpp:864-869, :889-893) behind the single pass and its refine: the points that
        registered are visited again in their next camera's loop, from their refined positions -- list, search, whole-track mergability
        (no cached tail: tolPix 0), the walks, refine; the points that registered again go round once more.  Every rank plays the rounds
        for ALL cameras on its replica (the lists are short): no collective."""
        from coslam_amd.register import register_revisit_decide_dev, register_revisit_list_dev, register_search_passes_dev

        cfg, NA = self.cfg, self.cfg.n_cams
        reg_in, keep = D["reg"], True
        for r in range(cfg.revisit_rounds):
            reg_out = self.d_rv_reg[r & 1]
            register_revisit_list_dev(ps, NA, self.n_map, self.RV_CAP, r == 0, self.d_pf.data_ptr(), D["att"].data_ptr(), reg_in.data_ptr(), keep,
                                      self.d_rv_visit.data_ptr(), self.d_rv_next.data_ptr(), self.d_rvlist.data_ptr(), self.d_rv_listcounts.data_ptr(),
                                      device=self.device, d_regOutClear=reg_out.data_ptr())
            register_search_passes_dev(ps, self.reg_args[self._dst_now], cfg.n_feat, cfg.W, cfg.H, self.rv_pass, device=self.device)
            self.pose_upd.register_mergability_running_dev(ps, self.pu_args, self.n_map, self.d_map.data_ptr(), self.d_cov.data_ptr(),
                                                           self.reg_out["slot"].data_ptr(), self.sig_pix, self.d_merge_cache.data_ptr(),
                                                           self.d_mergeable.data_ptr(), tolPix=0.0, d_counts=0, cam0=0, nCamsRun=NA,
                                                           d_list=self.d_rvlist.data_ptr(), nList=self.RV_CAP, d_flags=self.reg_out["flags"].data_ptr())
            D["s2m"] = register_revisit_decide_dev(ps, NA, cfg.n_feat, self.n_map, self.RV_CAP, 0, 3, self.d_rvlist.data_ptr(), self.d_rv_next.data_ptr(),
                                                   self.d_rv_visit.data_ptr(), self.reg_out["slot"].data_ptr(), self.reg_out["flags"].data_ptr(),
                                                   self.d_mergeable.data_ptr(), self.d_mapflags.data_ptr(), self.d_pf.data_ptr(), D["s2m"],
                                                   D["att"].data_ptr(), reg_out.data_ptr(), D["scr"].data_ptr(), self.d_curlist.data_ptr(),
                                                   self.d_curcount.data_ptr(), cfg.p_reg, self.d_rv_counts.data_ptr(), device=self.device,
                                                   d_listCount=self.d_rv_listcounts.data_ptr())
            self._refine(ps, reg_out.data_ptr(), self.d_rvlist.data_ptr(), self.RV_CAP)   # (the round changed the listed points only)
            reg_in, keep = reg_out, False

    def _advance_refs(self, ps, d_list=None, n_list=0):
        """MapPoint::pFeatures of this frame: cs_feat_ref_advance_dev behind whatever changed pointFeat (hand-back, classification, the
        registration's decisions) -- tracked on / first feature / re-linked behind an older one / stale / detached.  Idempotent within a frame;
        d_list: a further call of the frame over just the rows a registration round has changed."""
        if self.d_fref is not None:
            self.pose_upd.feat_ref_advance_dev(ps, self.pu_args, self.n_map, self.d_pf.data_ptr(), self._frame_now, self.d_fref.data_ptr(),
                                               d_refStatic=self.d_rstat.data_ptr(), d_counts=self.d_fref_counts.data_ptr(), d_list=d_list, nList=n_list)

    def _refine(self, ps, d_select, d_list=None, n_list=0):
        """CoSLAM::refineMapPoint of the points that gained a feature (:889-893, :666-713) -- over the feature references when they are kept
        (stale features of other cameras are views, a re-registered point's second view comes from its OLD chain), else over this frame's"""
        if self.d_fref is not None:
            self._advance_refs(ps, d_list, n_list)
            self.pose_upd.refine_map_points_ref_dev(ps, self.pu_args, self.d_fref.data_ptr(), self.n_map, self.d_map.data_ptr(), self.d_cov.data_ptr(),
                                                    self.sig_pix, d_select=d_select)
        else:
            self.pose_upd.refine_map_points_dev(ps, self.pu_args, self.d_pf.data_ptr(), self.n_map, self.d_map.data_ptr(), self.d_cov.data_ptr(),
                                                self.sig_pix, d_select=d_select)   # (no count asked for: that would be one more launch, and it is counts[1])

    def _gather_candidates(self):
        """the own cameras' columns of the current-static pass's candidate tables to every rank (18 KB per camera: latency-bound, ONE
        collective), so that every rank takes the same registration decisions on its replica"""
        import ctypes as C_

        import coslam_amd
        from coslam_amd._lib import check

        torch, cfg, nc, NA, P = self.torch, self.cfg, self.nc, self.cfg.n_cams, self.cfg.p_reg   # P: rows of a record = the list's cap
        L, vp, ps = coslam_amd.lib(), C_.c_void_p, self.pose_s.cuda_stream
        if not hasattr(self, "_cand"):
            self._cand = (torch.zeros(3 * nc * P, dtype=torch.int32, device=self.dev), torch.zeros(3 * nc * P * self.world, dtype=torch.int32, device=self.dev))
            torch.cuda.synchronize()   # (zero-filled on torch's stream, used on the pose stream)
        send, recv = self._cand
        o = self.reg_out
        check(L.cs_register_candidates_pack_list_dev(self.device, vp(ps), P, NA, self.c0, nc, vp(self.d_curlist.data_ptr()), vp(o["slot"].data_ptr()),
                                                     vp(o["flags"].data_ptr()), vp(self.d_mergeable.data_ptr()), vp(send.data_ptr())),
              "cs_register_candidates_pack_list_dev")
        if self.native is not None:
            L.cs_comm_allgather_dev.argtypes = [vp, vp, vp, vp, C_.c_size_t]
            check(L.cs_comm_allgather_dev(self.native.exchange_comm, vp(ps), vp(send.data_ptr()), vp(recv.data_ptr()), send.numel() * 4),
                  "cs_comm_allgather_dev")
        else:
            import torch.distributed as dist

            with torch.cuda.stream(self.pose_s):
                dist.all_gather_into_tensor(recv, send)
        check(L.cs_register_candidates_unpack_list_dev(self.device, vp(ps), P, NA, nc, self.rank, vp(self.d_curlist.data_ptr()), vp(recv.data_ptr()),
                                                       vp(o["slot"].data_ptr()), vp(o["flags"].data_ptr()), vp(self.d_mergeable.data_ptr())),
              "cs_register_candidates_unpack_list_dev")

    def _key_frame(self, i, dst, snap=None):
        cfg, ps, NA = self.cfg, self.pose_s.cuda_stream, self.cfg.n_cams
        k_ic = self.n_key
        self.n_key += 1
        if cfg.with_intercam and (k_ic + self.world // 2) % self.world == self.rank:
            # InterCamPoseEstimator::addMapPoints + apply (reference src/app/SL_InterCamPoseEstimator.cpp:18-95): every camera's CURRENT
            # pose (own: just solved; others: this frame's all-gather), the static features chosen per block with their map points
            # held fixed, the dynamic points free; sigma 6, 3 x 40
            with self._sec("kf_intercam"):
                if self.icam is not None:
                    w = self.n_my_ic % self.n_ic_workers
                    self.icams[w].solve_async(self.ic_wss[w], ps, self.ic_cams, cfg.W, cfg.H, cfg.n_col_blk, cfg.n_row_blk, self.d_R[dst].data_ptr(),
                                          self.d_t[dst].data_ptr(), self.d_map.data_ptr(), self.d_mapflags.data_ptr(), self.d_newpt.data_ptr(),
                                          self.d_pf.data_ptr(), 6.0, 3, 40)
                else:
                    with self.torch.cuda.stream(self.pose_s):
                        self.d_iR.copy_(self.d_R[dst].view(-1), non_blocking=True)
                        self.d_iT.copy_(self.d_t[dst].view(-1), non_blocking=True)
                    self.ic_ws.solve_async(ps, self.d_iR.data_ptr(), self.d_iT.data_ptr(), self.d_iM.data_ptr(), 0, self.ic["n_static"], 6.0, 3, 40)
            self.n_my_ic += 1
        if self.win is None:
            return
        # this key frame into the ring (every camera's records and poses: identical on every rank), then requestForBA(5, 2, 2, 30):
        # the numCams * 2 oldest key cameras held, 2 points held, maxIter 2, inner 10 -- solved by ONE rank
        with self._sec("kf_push"):
            if snap is not None:   # (a lagged decision: the key frame's own records and poses, kept in the ring)
                self.win.push_dev(ps, snap["hb"], self.d_K1.data_ptr(), 1, snap["R"].data_ptr(), snap["t"].data_ptr(), i)
            else:
                self.win.push_dev(ps, self.hb_all, self.d_K1.data_ptr(), 1, self.d_R[dst].data_ptr(), self.d_t[dst].data_ptr(), i)
        self.n_pushed += 1
        self.pushed_frames = (self.pushed_frames + [i])[-cfg.n_key_frames:]   # the ring's key frames, oldest first
        if self.n_pushed < cfg.n_key_frames:
            return
        if self.skip_busy:
            # the reference's own policy (CoSLAM::requestForBA, src/app/SL_CoSLAM.cpp:1750-1755): a request that finds the previous bundle
            # adjustment still running is dropped.  "Running" is asked when the DEVICE has reached this key frame (the host enqueues
            # frames ahead of it): one host wait per key frame.  Timing-dependent, so only on one rank (bench.py's secondary figure)
            self.pose_s.synchronize()
            if self.ba_ws.pending() > 0:
                self.n_skipped += 1
                return
        k = self.n_windows
        self.n_windows += 1
        owner = k % self.world
        if owner == self.rank:
            with self._sec("kf_request"):
                self.win.solve_flags_async(self.ba_ws, ps, self.d_map.data_ptr(), self.d_mapflags.data_ptr(), 2 * NA, 2, 6.0, 2, 10)
            self.my_seq[k] = self.n_my_solves
            self.n_my_solves += 1
        # the record's sequence number ON ITS OWNER: windows go round the ranks, so it is the owner's (k // world)-th solve (one rank:
        # its own count, which the reference's request policy -- skip_busy -- may leave behind k)
        seq = self.my_seq[k] if owner == self.rank else k // self.world
        first = i - (cfg.n_key_frames - 1) * cfg.key_every
        if cfg.keyframe_drives:
            first = list(self.pushed_frames)   # the window's key frames where the decision put them
        self.apply_at[i + self.lag * cfg.key_every] = (k, owner, first, seq)

    def measure_collectives(self, n=40):
        """GPU-clock latency of each of the frame loop's collectives as THIS loop issues them (the buffers of the last frame, the pose
        stream; every rank calls it at the same point): microseconds per call, averaged over n back-to-back calls.  N = 1: {}."""
        if self.world == 1:
            return {}
        torch, ps = self.torch, self.pose_s
        out = {}

        def timed(name, fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            torch.cuda.synchronize()
            e0.record(ps)
            for _ in range(n):
                fn()
            e1.record(ps)
            e1.synchronize()
            out[name] = e0.elapsed_time(e1) * 1e3 / n

        def gather_features():
            with torch.cuda.stream(ps):
                self.xchg.pack_group(self.d_dests[0], self.d_R[0][self.c0:self.c0 + self.nc], self.d_t[0][self.c0:self.c0 + self.nc], ps)
                self.xchg.all_gather(ps)

        timed("all_gather_features_and_poses_per_frame", gather_features)
        if self.cfg.with_register and self.cfg.with_decide and self.pose_upd is not None and hasattr(self, "_cand"):
            timed("all_gather_registration_candidates_per_frame", self._gather_candidates)
        if self.ncc is not None:
            timed("all_gather_ncc_records_every_4th_frame", self._gather_ncc_records)
        if self.out is not None:
            rec = self.recv_rec[0].data_ptr()
            timed("broadcast_ba_result_per_key_frame", lambda: self.xchg.broadcast(rec, self.out.record_bytes, 0, self.device, ps))
        out["bytes"] = {"features_and_poses_per_rank": int(self.nc * (self.cfg.n_feat * 20 + 96)),
                        "registration_candidates_per_rank": int(3 * self.nc * self.cfg.p_reg * 4),
                        "ncc_records_per_rank": int(self.nc * self.ncc["rec_bytes"]) if self.ncc is not None else 0,
                        "ba_result": int(self.out.record_bytes) if self.out is not None else 0}
        return out

    def drain(self):
        """the worker threads' queues are part of the work: every requested solve completes.  A window / a rig that holds no usable point
        (every map point of its key frames false, say) is a solve with nothing to do, not a failure of the loop: it packed an empty
        record (ok = 0, applies nothing) and is counted here."""
        import coslam_amd

        for w in self.ic_wss + [self.ba_ws]:
            try:
                w.wait()
            except coslam_amd.CoslamHipError as ex:
                if "no map point has two feature points" in str(ex) or "no static feature point carries a map point" in str(ex):
                    self.n_empty_solves = getattr(self, "n_empty_solves", 0) + 1
                else:
                    raise
        self.torch.cuda.synchronize()

    def digest_parts(self):
        """per-array short digests (diagnostic: finding where two runs part ways)"""
        import hashlib

        self.drain()
        names = ("map", "cov", "mapflags", "slot2map", "trackspan", "xy", "state", "isstatic", "R0", "R1", "t0", "t1", "pf", "newpt", "sfn")
        arrs = (self.d_map, self.d_cov, self.d_mapflags, self.d_slot2map, self.d_trackspan, self.d_xy, self.d_state, self.d_isstatic,
                self.d_R[0], self.d_R[1], self.d_t[0], self.d_t[1], self.d_pf, self.d_newpt, self.d_sfn)
        return {n: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:10] for n, t in zip(names, arrs)}

    def digest(self):
        """sha256 over the state every rank must agree on (the map, its flags and covariances, every camera's records and poses)"""
        import hashlib

        self.torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in (self.d_map, self.d_cov, self.d_mapflags, self.d_slot2map, self.d_trackspan, self.d_xy, self.d_state, self.d_isstatic,
                  self.d_R[0], self.d_R[1], self.d_t[0], self.d_t[1], self.d_pf):
            h.update(t.cpu().numpy().tobytes())
        return h.hexdigest()
