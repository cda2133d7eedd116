#!/usr/bin/env python
"""bench.py -- frames/sec of the track + pose + local-BA loop on MI355X (BASELINE.json metric).

Workload (BASELINE.json `metric`): 8 synchronised cameras 640x480, 2000 KLT feature slots each.  One "step" = ONE FRAME of
the whole rig, i.e. every camera's image consumed and poses updated:
    all cameras   redetect (pyramid -> KLT track with gain -> corner detect -> top-K -> slot fill) + advanceFrame
                  [CoSLAM::featureTracking, reference src/app/SL_CoSLAM.cpp:299-305 -> GPUKLT::next]
    all cameras   hand-back on the device (undistort, track bookkeeping, one static mapped track per 40x40 block, Ms/ms
                  packing) [GPUKLT::addToFeaturePoints, SingleSLAM::chooseStaticFeatPts / poseUpdate3D] feeding
                  intraCamEstimate of every camera, started from the previous frame's result [SL_CoSLAM.cpp:366-417]
    all cameras   map-point registration, search step, twice per frame as the reference does [CoSLAMThread.cpp:108-118]:
                  activeMapPointsRegister (1536 active points: covariance sigma 2.5 x, SL_CoSLAM.cpp:1118-1145) and
                  currentMapPointsRegister (1536 current static points, skipping cameras where the point already has a
                  feature of this frame, :731-757): projection with the poses just solved + Mahalanobis-nearest feature
                  over the hand-back's records of every camera
    key frames    (every KEY_EVERY-th frame)
                  joint local BA: last 5 key frames of all 8 cameras = 40 cameras, the 16 oldest fixed, maxIter 2 / inner
                  10 [requestForBA(5, 2, 2, 30) -> RobustBundleRTS, SL_CoSLAM.cpp:1345,1731-1784], on its own stream like
                  the reference's BA worker thread;
                  inter-camera pose solve: the 8 current cameras free, 8 x 192 static points fixed, 60 dynamic points free,
                  sigma 6, 3 x 40 [InterCamPoseEstimator, SL_InterCamPoseEstimator.cpp:18-95].
    N > 1         the cameras are sharded over the ranks (8 / N each); per frame one all-gather of {features, pose} of
                  every camera; the joint BA is sliced by points over the ranks with one all-reduce of S || rhs per LM step.
Inputs (images, map points, BA problems) are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

# The camera-group kernels take their per-camera pointer tables BY VALUE (1-2 KB of kernel arguments per launch, ~10 KB per
# frame).  The HIP runtime hands kernel arguments out of a 1 MB ring and stalls the launching thread for ~12 ms every time the
# ring wraps (every ~290 frames here: one 12 ms step in a 0.45 ms-per-frame loop, measured: host_enqueue_ms_max_step).  A
# larger ring makes the wrap rare; it has to be set before the runtime initialises.  
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(64 << 20))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CAMS = 8
W, H, LEVELS, FW, FH = 640, 480, 4, 50, 40
N_FEAT = FW * FH
N_FRAMES = 120           # the video: a closed camera path of this many frames (coslam_amd.synth.Scene loop_period)
PTS_STRIDE = 192          # 12 x 16 blocks (reference src/app/SL_SingleSLAM.h:36-37)
N_COL_BLK, N_ROW_BLK = 16, 12
KEY_EVERY = 5
P_REG = 4096             # cap on the frame's CURRENT map points (the registration's list; ~2300 of them in the steady state)
CPU_P_REG = 1536         # map points per registration pass of the CPU baseline's restatement (rounds 2-4's pass size)
PIXEL_ERR_VAR = 10.0      # Const::PIXEL_ERR_VAR, reference src/app/SL_GlobParam.cpp:37
MAX_EPI_ERR = 6.0         # Const::MAX_EPI_ERR, :36
HBM_PEAK_GBS = 8000.0
SETUP_SECONDS = float(os.environ.get("BENCH_SETUP_SECONDS", "0.5"))   # untimed set-up run before the warm-up (see main)
SEED = 0xC051A + 2


def klt_config():
    import coslam_amd

    # 4 levels, levelSkip 1, 7x7 window, 10 iterations/level, with gain
    return coslam_amd.KLT_SequenceTrackerConfig(nIterations=10, nLevels=LEVELS, levelSkip=1, windowWidth=7,
                                                trackWithGain=1, minCornerness=3000.0, convergenceThreshold=1.0,
                                                SSD_Threshold=20000.0, minDistance=4)


def frame_order(n):
    # the video is a closed path: frame n is frame 0 again
    return list(range(n))


def build_scene():
    from coslam_amd.synth import Scene

    if os.environ.get("BENCH_VIDEO") == "pingpong":   # (diagnostic: the straight 24-frame path of rounds 1-3, played back and forth)
        return Scene(N_CAMS, W, H, 7000, seed=SEED, sigma=1.0)
    return Scene(N_CAMS, W, H, 7000, seed=SEED, sigma=1.0, loop_period=N_FRAMES)


def build_ic_problem(sc):
    from coslam_amd.synth import make_intercam_problem

    return make_intercam_problem(sc, seed=SEED + 11)


def build_joint_problem(sc):
    """a pre-baked joint problem of the headline's size: the CPU baseline's stand-in for the window parse (the GPU loop parses its
    own from the tracked records)"""
    from coslam_amd.synth import make_joint_ba_problem

    return make_joint_ba_problem(sc, seed=SEED + 7)


def build_ba_problems(sc):
    """(joint, inter-camera): the two key-frame problems at the headline's size (tests/test_pose_ba_gpu.py solves both against the oracle)"""
    return build_joint_problem(sc), build_ic_problem(sc)


def csr(pr):
    from coslam_amd.synth import csr_of_problem

    return csr_of_problem(pr)


def associate(sc, cam, frame, dest):
    """slot -> scene point: the map point a freshly detected feature belongs to (nearest projected point within 1 px).
    Stands in for CoSLAM's map initialisation (out of scope); computed once, before the clock starts."""
    from scipy.spatial import cKDTree

    uv, vis = sc.project(cam, frame)
    idx = np.nonzero(vis)[0]
    s2m = np.full(len(dest), -1, dtype=np.int32)
    live = np.nonzero(dest["status"] >= 0)[0]
    if len(live) == 0 or len(idx) == 0:
        return s2m
    p = dest["pos"][live].astype(np.float64) * [W, H]
    d, j = cKDTree(uv[idx]).query(p, distance_upper_bound=1.0)
    ok = np.isfinite(d)
    s2m[live[ok]] = idx[j[ok]]
    return s2m


def build_pose_graphs(sc, joint, cams, n_kf=5, kf_step=5):
    """The camera graphs RobustBundleRTS::constructCameraGraphs builds for the joint BA's window (reference
    src/app/SL_CoSLAMRobustBA.cpp:182-229): per camera the chain of frames from the first key frame of the window to the last,
    key frames fixed.  Poses before the adjustment: the BA's initial estimate at the key frames, the tracked poses in
    between.  Returns (graphs, nodeR, nodeT, camNode): camNode[j] = node of BA camera j (j = kf * numCams + c) or -1."""
    n = (n_kf - 1) * kf_step + 1
    graphs, R, T = [], [], []
    camNode = np.full(len(joint["Rs0"]), -1, dtype=np.int32)
    for gi, c in enumerate(cams):
        fixed = np.zeros(n, dtype=np.uint8)
        fixed[::kf_step] = 1
        for f in range(n):
            if f % kf_step == 0:
                j = (f // kf_step) * sc.C + c
                camNode[j] = gi * n + f
                R.append(joint["Rs0"][j].reshape(9)), T.append(joint["ts0"][j])
            else:
                Rf, tf = sc.pose(c, f)
                R.append(Rf.reshape(9)), T.append(tf)
        graphs.append((fixed, np.arange(n - 1, dtype=np.int32), np.arange(1, n, dtype=np.int32)))
    return graphs, np.array(R), np.array(T), camNode


def reg_covariances(n=None):
    """MapPoint::cov of the first n map points (default: the 2 x P_REG registered ones): synthetic SPD 3 x 3, a few cm"""
    rng = np.random.default_rng(SEED + 23)
    A = rng.normal(size=(2 * CPU_P_REG if n is None else max(n, 2 * CPU_P_REG), 3, 3)) * 0.02
    return A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3)


def export_workload(path, sc, frames, joint, ic, cams_per_launch):
    """The headline workload as one binary file for tools/cxx/frame_loop.cpp (the same loop driven from C++ through the C-ABI):
    magic, int32 header[16], frame order, K, the KLT configuration, the frames, the map, the projections of the visible points in
    the first frame (for the slot -> map association), the initial poses, the registration covariances, both BA problems, the
    pose graphs of the joint BA's window."""
    import struct

    order = frame_order(N_FRAMES)
    cfg = klt_config()
    pg_graphs, pg_R, pg_T, pg_cam = build_pose_graphs(sc, joint, range(N_CAMS))
    with open(path, "wb") as f:
        f.write(b"CSWL1\0\0\0")
        hd = [N_CAMS, W, H, LEVELS, FW, FH, N_FRAMES, len(order), len(sc.points), P_REG, PTS_STRIDE, N_COL_BLK, N_ROW_BLK, KEY_EVERY,
              cams_per_launch, 0]
        f.write(np.asarray(hd, np.int32).tobytes())
        f.write(np.asarray(order, np.int32).tobytes())
        f.write(np.ascontiguousarray(sc.K, np.float64).tobytes())
        f.write(np.asarray([cfg.nIterations, cfg.nLevels, cfg.levelSkip, cfg.windowWidth, cfg.trackWithGain, cfg.minDistance], np.int32).tobytes())
        f.write(np.asarray([cfg.trackBorderMargin, cfg.convergenceThreshold, cfg.SSD_Threshold, cfg.minCornerness, cfg.detectBorderMargin],
                           np.float32).tobytes())
        for c in range(N_CAMS):
            f.write(np.ascontiguousarray(frames[c], np.uint8).tobytes())
        f.write(np.ascontiguousarray(sc.points, np.float64).tobytes())
        for c in range(N_CAMS):
            uv, vis = sc.project(c, order[0])
            idx = np.nonzero(vis)[0].astype(np.int32)
            f.write(struct.pack("i", len(idx)))
            f.write(idx.tobytes())
            f.write(np.ascontiguousarray(uv[idx], np.float64).tobytes())
        f.write(np.stack([sc.pose(c, order[0])[0].ravel() for c in range(N_CAMS)]).astype(np.float64).tobytes())
        f.write(np.stack([sc.pose(c, order[0])[1] for c in range(N_CAMS)]).astype(np.float64).tobytes())
        f.write(np.ascontiguousarray(reg_covariances(len(sc.points)), np.float64).tobytes())   # MapPoint::cov of every map point
        for pr, ncon, npcon, mi, inner in ((joint, joint["n_cams_con"], joint["n_pts_con"], 2, 10), (ic, 0, ic["n_static"], 3, 40)):
            ptr, cam, xy = csr(pr)
            f.write(np.asarray([len(pr["Rs0"]), len(pr["pts0"]), len(cam), ncon, npcon, mi, inner], np.int32).tobytes())
            f.write(struct.pack("d", 6.0))
            for a in (pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"]):
                f.write(np.ascontiguousarray(a, np.float64).tobytes())
            f.write(np.ascontiguousarray(ptr, np.int32).tobytes())
            f.write(np.ascontiguousarray(cam, np.int32).tobytes())
            f.write(np.ascontiguousarray(xy, np.float64).tobytes())
        npc = len(pg_graphs[0][0])
        f.write(np.asarray([len(pg_graphs), npc], np.int32).tobytes())
        f.write(np.concatenate([g[0] for g in pg_graphs]).astype(np.uint8).tobytes())
        f.write(np.ascontiguousarray(pg_R, np.float64).tobytes())
        f.write(np.ascontiguousarray(pg_T, np.float64).tobytes())
        f.write(np.ascontiguousarray(pg_cam, np.int32).tobytes())
        # fundamental matrices of the consecutive camera pairs at every frame (the NCC matching leg): [frame][pair][9]
        Kinv = np.linalg.inv(sc.K)
        Fs = np.zeros((N_FRAMES, N_CAMS - 1, 9))
        for fr in range(N_FRAMES):
            for c in range(N_CAMS - 1):
                (R1, t1), (R2, t2) = sc.pose(c, fr), sc.pose(c + 1, fr)
                R = R1 @ R2.T
                t = t1 - R @ t2
                E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
                Fs[fr, c] = (Kinv.T @ E @ Kinv).reshape(9)
        f.write(Fs.tobytes())


def cpu_baseline(sc, frames, joint, ic, n_threads, budget_s, with_register=True, with_posegraph=True, with_ncc=True, reading="variance"):
    """The oracle (C restatement of the reference's path: kind "port") on `n_threads` host cores: the cameras of a frame
    in parallel (the ctypes calls release the GIL), the key-frame solves on the calling thread."""
    import oracle

    # what the covariance helpers get where the reference passes Const::PIXEL_ERR_VAR (a variance: sqrt(10) px; "std": the constant as it is)
    SIG = float(np.sqrt(PIXEL_ERR_VAR)) if reading == "variance" else PIXEL_ERR_VAR
    SIG_CLS = float(np.sqrt(12.0)) if reading == "variance" else 12.0
    order = frame_order(N_FRAMES)
    cfg = klt_config()
    trk, s2m, tl, xy, Rc, tc, st = [], [], [], [], [], [], []
    for c in range(N_CAMS):
        o = oracle.SequenceTracker(cfg)
        o.allocate(W, H, LEVELS, FW, FH)
        _, d = o.detect(frames[c][order[0]])
        o.advanceFrame()
        trk.append(o)
        s2m.append(associate(sc, c, order[0], d))
        tl.append(np.full(2 * N_FEAT, -1, dtype=np.int32))
        xy.append(np.zeros(2 * N_FEAT))
        st.append(oracle.handback(d, W, H, sc.K, np.zeros(7), sc.points, s2m[c], tl[c], xy[c], 0)["state"])
        s2m[c][:] = associate(sc, c, order[0], d)   # (the first hand-back resets new tracks: restore the map)
        R, t = sc.pose(c, order[0])
        Rc.append(R.copy())
        tc.append(t.copy())
    jptr, jcam, jxy = csr(joint)
    iptr, icam, ixy = csr(ic)
    kud = np.zeros(7)
    pg_graphs, pg_R, pg_T, pg_cam = build_pose_graphs(sc, joint, range(N_CAMS))
    npc = len(pg_graphs[0][0])
    ends = np.concatenate([np.arange(npc - 1) + g * npc for g in range(len(pg_graphs))])
    pg_eR, pg_eT = oracle.posegraph_edges(pg_R, pg_T, ends, ends + 1)

    def cam_step(c, f, frame_no):
        _, d = trk[c].redetect(frames[c][f])
        trk[c].advanceFrame()
        hb = oracle.handback(d, W, H, sc.K, kud, map_pts, s2m[c], tl[c], xy[c], frame_no, isStatic=is_static[c])
        st[c] = hb["state"]
        if hb["npts"] >= 6:
            ok, R, t, _ = oracle.intracam_estimate(sc.K, Rc[c], tc[c], hb["npts"], None, hb["Ms"], hb["ms"], 10.0)
            if ok:
                Rc[c], tc[c] = R, t

    cov = reg_covariances(len(sc.points))
    no_feat = np.full((CPU_P_REG, 1), -1, dtype=np.int32)
    # poseUpdate3D's gate + seqTriangulate and the dynamic-point test: camera after camera on the calling thread (they share the map)
    map_pts, map_cov = sc.points.copy(), np.ascontiguousarray(cov[:len(sc.points)].reshape(-1, 9))
    map_flags = np.zeros(len(sc.points), dtype=np.uint8)
    reproj = [np.zeros(N_FEAT) for _ in range(N_CAMS)]
    is_static = [np.ones(N_FEAT, dtype=np.uint8) for _ in range(N_CAMS)]
    hist = [dict(R=[], t=[], xy=[]) for _ in range(N_CAMS)]
    iK = np.linalg.inv(sc.K)
    cls_new, cls_sfn = np.zeros(len(sc.points), dtype=np.uint8), np.zeros(len(sc.points), dtype=np.int32)
    cls_first = np.zeros(len(sc.points), dtype=np.int32)

    def pose_update_all(frame_no):
        for c in range(N_CAMS):
            oracle.pose_update_gate([Kc] * N_CAMS, np.stack([r.reshape(9) for r in Rc]), np.stack(tc), xy, st, s2m, map_pts, map_cov,
                                    map_flags, 0, SIG, reproj, cams=[c])
            h = hist[c]
            h["R"].insert(0, Rc[c].reshape(9).copy()), h["t"].insert(0, tc[c].copy()), h["xy"].insert(0, xy[c].copy())
            del h["R"][64:], h["t"][64:], h["xy"][64:]
            oracle.detect_dynamic(iK, np.stack(h["R"]), np.stack(h["t"]), np.stack(h["xy"]), st[c], s2m[c], tl[c], map_flags, 20, 5, 3,
                                  MAX_EPI_ERR, is_static[c])
        # CoSLAM::mapPointsClassify(12.0) behind the cameras' pose updates
        pf_all = np.ascontiguousarray(np.stack([oracle.point_features(st[c], s2m[c], len(sc.points)) for c in range(N_CAMS)], 1))
        fs_all = np.ascontiguousarray(np.stack(is_static))
        oracle.map_points_classify([Kc] * N_CAMS, [iK] * N_CAMS, np.stack([np.stack(h["R"]) for h in hist]),
                                   np.stack([np.stack(h["t"]) for h in hist]), np.stack([np.stack(h["xy"]) for h in hist]), np.stack(tl),
                                   fs_all, pf_all, frame_no, map_pts, map_cov, map_flags, cls_new, cls_sfn, cls_first, SIG_CLS)
        for c in range(N_CAMS):
            is_static[c][:] = fs_all[c]

    Kc = sc.K

    # currentMapPointsRegister as the GPU loop runs it: ONE search pass over the frame's CURRENT points (the map points with a feature of
    # this frame in some camera, wherever they sit in the map), tables indexed by the map index; no active pass (the reference's attach
    # loop behind it cannot be reached: tests/cxx/ref_active_test.cpp)
    nMapAll = len(sc.points)
    reg_slot, reg_flags = np.full((nMapAll, N_CAMS), -1, np.int32), np.zeros((nMapAll, N_CAMS), np.int32)
    reg_merge, reg_pf = np.zeros((nMapAll, N_CAMS), np.uint8), np.full((nMapAll, N_CAMS), -1, np.int32)
    cur_list = [np.zeros(0, np.int64)]

    def list_current():
        for c in range(N_CAMS):
            reg_pf[:, c] = oracle.point_features(st[c], s2m[c], nMapAll)
        cur_list[0] = np.nonzero((reg_pf >= 0).any(axis=1) & ((map_flags & 2) == 0))[0]
        reg_slot[:] = -1

    def reg_step(c):
        one = lambda a: [a]  # noqa: E731
        idx = cur_list[0]
        if len(idx) == 0:
            return
        pf = np.ascontiguousarray(reg_pf[idx, c:c + 1])
        rs = oracle.register_search(W, H, Kc, Rc[c], tc[c], one(xy[c]), one(st[c]), one(s2m[c]), one(None), map_pts[idx], cov[idx], pf, SIG,
                                    3 * PIXEL_ERR_VAR, SIG)
        reg_slot[idx, c], reg_flags[idx, c] = rs["slot"][:, 0], rs["flags"][:, 0]
        h = hist[c]
        reg_merge[idx, c] = 0
        if h["R"]:   # staticCheckMergability of the candidates over the frames the history holds (64: the restatement keeps no store; a longer
            # track comes out "unjudged" -- the GPU's running verdict judges it)
            reg_merge[idx, c] = oracle.register_mergability_cam(Kc, np.stack(h["R"]), np.stack(h["t"]), np.stack(h["xy"]), tl[c], map_pts[idx],
                                                                cov[idx], rs["slot"][:, 0], SIG)

    def decide_all():
        # currentMapPointsRegister's decisions over the cameras' columns (org_register_decide: static points, then dynamic ones), then
        # refineMapPoint of the points that gained a feature -- on the calling thread, behind the cameras' searches
        s2m_all = np.ascontiguousarray(np.stack(s2m)).astype(np.int32)
        pf_all = np.ascontiguousarray(reg_pf)
        _, reg = oracle.register_decide_static_c(reg_slot, reg_flags, reg_merge, map_flags, pf_all, s2m_all, kinds=3)
        if reg.any() and hist[0]["R"]:
            for c in range(N_CAMS):
                s2m[c][:] = s2m_all[c]
            pf_map = np.ascontiguousarray(np.stack([oracle.point_features(st[c], s2m[c], nMapAll) for c in range(N_CAMS)], 1))
            oracle.refine_map_points([Kc] * N_CAMS, [iK] * N_CAMS, np.stack([np.stack(h["R"]) for h in hist]), np.stack([np.stack(h["t"]) for h in hist]),
                                     np.stack([np.stack(h["xy"]) for h in hist]), np.stack(tl), pf_map, map_pts, map_cov.reshape(-1, 9), SIG,
                                     select=np.ascontiguousarray(reg, dtype=np.uint8))

    # genNewMapPoints' NCC stage, every 4th frame like the GPU loop: getNCCBlocks of a camera's candidate features (with its thread),
    # getEpiNccMat of the consecutive camera pairs behind the cameras, then the match / reconstruct / decidePointType tail (all C restatements)
    ncc_rec = [None] * N_CAMS
    ncc_pairs = [np.zeros((0, 4))] * (N_CAMS - 1)
    Kinv_ = np.linalg.inv(sc.K)

    def ncc_cam(c, f):
        f1, f2, m = tl[c][:N_FEAT], tl[c][N_FEAT:], s2m[c]
        free_or_false = (m < 0) | ((map_flags[np.clip(m, 0, len(map_flags) - 1)] & 2) != 0)
        idx = np.nonzero(((st[c] == 0) | (st[c] == 1)) & (f1 >= 0) & (f2 - f1 >= 3) & free_or_false)[0]
        x, y = np.ascontiguousarray(xy[c][:N_FEAT][idx]), np.ascontiguousarray(xy[c][N_FEAT:][idx])
        blk, abc = oracle.get_ncc_blocks(frames[c][f], x, y, 0.3)
        ncc_rec[c] = (x, y, blk, abc, np.ones(len(idx), dtype=np.int32), idx)

    def ncc_pair(a, f):
        (R1, t1), (R2, t2) = sc.pose(a, f), sc.pose(a + 1, f)
        R = R1 @ R2.T
        t = t1 - R @ t2
        F = Kinv_.T @ (np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R) @ Kinv_
        (x1, y1, b1, c1, v1, i1), (x2, y2, b2, c2, v2, i2) = ncc_rec[a], ncc_rec[a + 1]
        ncc_pairs[a] = np.zeros((0, 4))
        if len(x1) and len(x2):
            epi, ncc = oracle.ncc_epi_mat(F, x1, y1, b1, c1, v1, x2, y2, b2, c2, v2, 50.0, 0.80)
            ii, jj = np.nonzero(ncc >= 0)          # the pairs that passed both gates: what NewMapPtsNCC::matchBetween's matcher gets
            ncc_pairs[a] = np.stack([i1[ii].astype(np.float64), i2[jj].astype(np.float64), epi[ii, jj], ncc[ii, jj]], 1)

    tail_s = [0.0]

    def ncc_tail(frame_no):
        # NewMapPtsNCC::run's tail + output in C (onc_new_points_from_pairs): seeds, disparity guide, greedy matches, tracks, triangulation,
        # decidePointType -- on a scratch copy of the map with room behind it (the CPU loop's map does not grow: the result is dropped)
        t0 = time.perf_counter()
        n0 = len(map_pts)
        sp = 2048
        M_ = np.concatenate([map_pts, np.zeros((sp, 3))])
        C_ = np.concatenate([map_cov.reshape(-1, 9), np.zeros((sp, 9))])
        fl_ = np.concatenate([map_flags, np.zeros(sp, np.uint8)])
        pf_ = np.ascontiguousarray(np.concatenate([np.stack([oracle.point_features(st[c], s2m[c], n0) for c in range(N_CAMS)], 1).astype(np.int32),
                                                   np.full((sp, N_CAMS), -1, np.int32)]))
        oracle.new_map_points_from_pairs_c(N_FEAT, [ncc_pairs[a] for a in range(N_CAMS - 1)], [sc.K] * N_CAMS, [Kinv_] * N_CAMS, Rc, tc, xy, st,
                                           [q.copy() for q in s2m], is_static, M_, C_, fl_, np.zeros(n0 + sp, np.uint8), np.zeros(n0 + sp, np.int32),
                                           pf_, n0, frame_no, max_disp=80.0, max_rp_err=3.0, sigma=SIG, min_len=2, W=W, H=H)
        tail_s[0] += time.perf_counter() - t0

    def run_cams(cams, f, frame_no):
        for c in cams:
            cam_step(c, f, frame_no)

    def run_reg(cams):
        for c in cams:
            reg_step(c)

    def run_ncc(what, items, f):
        for q in items:
            what(q, f)

    def in_threads(fn, items, *a):
        if n_threads <= 1:
            fn(list(items), *a)
            return
        items = list(items)
        th = [threading.Thread(target=fn, args=(items[q::n_threads],) + a) for q in range(n_threads) if items[q::n_threads]]
        for x in th:
            x.start()
        for x in th:
            x.join()

    kf_ring, joint_sizes = [], [0, 0, 0]
    t_start = time.perf_counter()
    n = 0
    while True:
        f = order[(n + 1) % len(order)]
        in_threads(run_cams, range(N_CAMS), f, n + 1)
        if with_register:   # (behind ALL cameras' hand-backs: the list of current points needs every camera's features)
            list_current()
            in_threads(run_reg, range(N_CAMS))
            decide_all()
        pose_update_all(n + 1)
        if with_ncc and (n + 1) % 4 == 0:
            in_threads(lambda cs, f_: run_ncc(ncc_cam, cs, f_), range(N_CAMS), f)
            in_threads(lambda ps, f_: run_ncc(ncc_pair, ps, f_), range(N_CAMS - 1), f)
            ncc_tail(n + 1)
        if n % KEY_EVERY == 0:
            # RobustBundleRTS::addKeyFrames / addPoints / parseInputs over the last 5 key frames' records (the window the GPU loop parses on
            # the device), then requestForBA(5, 2, 2, 30): 2 * numCams oldest key cameras and 2 points held, maxIter 2, inner 10; static points
            kf_ring.append([dict(xy=xy[c].copy(), state=st[c].copy(), slot2map=s2m[c].copy(), K=sc.K.ravel(), R=Rc[c].ravel().copy(), t=tc[c].copy())
                            for c in range(N_CAMS)])
            del kf_ring[:-5]
            if len(kf_ring) == 5:
                pw = oracle.parse_inputs_window_fast(kf_ring, map_pts, (map_flags & 7) == 0)
                jR, jT = oracle.ba_robust(pw["Ks"].reshape(-1, 3, 3), pw["Rs"].reshape(-1, 3, 3), pw["Ts"], pw["pts"], pw["obs_ptr"], pw["obs_cam"],
                                          pw["obs_xy"], 2 * N_CAMS, 2, 6.0, 2, 10)[:2]
                joint_sizes[:] = [len(pw["Ks"]), len(pw["pts"]), len(pw["obs_cam"])]
            else:   # (the first four key frames: the pre-baked problem of the same size stands in, as in rounds 3-4)
                jR, jT = oracle.ba_robust(joint["Ks"], joint["Rs0"], joint["ts0"], joint["pts0"], jptr, jcam, jxy,
                                          joint["n_cams_con"], joint["n_pts_con"], 6.0, 2, 10)[:2]
            if with_posegraph:   # RobustBundleRTS::output(): the non-key frames follow the adjusted key frames
                nR, nT = pg_R.copy(), pg_T.copy()
                nR[pg_cam[pg_cam >= 0]], nT[pg_cam[pg_cam >= 0]] = jR.reshape(-1, 9)[pg_cam >= 0], jT[pg_cam >= 0]
                npc = len(pg_graphs[0][0])
                for g, (fx, a, b) in enumerate(pg_graphs):
                    ns, es = slice(g * npc, (g + 1) * npc), slice(g * (npc - 1), (g + 1) * (npc - 1))
                    oracle.posegraph_relax(fx, nR[ns], nT[ns], a, b, pg_eR[es], pg_eT[es])
            if hist[0]["R"]:   # RobustBundleRTS::updateNewPosesPoints behind the adjustment (on a copy of the map, like the GPU loop)
                pf_all = np.stack([oracle.point_features(st[c], s2m[c], len(sc.points)) for c in range(N_CAMS)], 1)
                oracle.update_new_poses_points([Kc] * N_CAMS, [iK] * N_CAMS, np.stack([np.stack(h["R"]) for h in hist]),
                                               np.stack([np.stack(h["t"]) for h in hist]), np.stack([np.stack(h["xy"]) for h in hist]),
                                               np.stack(tl), np.stack(is_static), pf_all, map_pts.copy(), map_cov.copy(), map_flags,
                                               SIG)
            oracle.ba_robust(ic["Ks"], ic["Rs0"], ic["ts0"], ic["pts0"], iptr, icam, ixy, 0, ic["n_static"], 6.0, 3, 40)
        n += 1
        if time.perf_counter() - t_start > budget_s or n >= 200:
            break
    dt = time.perf_counter() - t_start
    joint_sizes.append(tail_s[0] / dt)   # (the share of the new-map-point tail in the figure)
    return n / dt, n, dt, joint_sizes


def live_tracker_pmc():
    """FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU of k_track_rows_fused per launch (KB, KB, wave instructions), collected now: one rocprofv3 --pmc
    pass each over tools/pmc_klt.py (the 8-camera KLT stage), summarised by tools/rocpd_summary.py.  {"error": ...} when anything fails."""
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    out = {}
    td = tempfile.mkdtemp(prefix="coslam_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            d = os.path.join(td, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "pmc_klt.py")]
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 start_new_session=True)
            try:
                p.wait(timeout=45)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)   # (the process group this call started: rocprofv3 and its child)
                return {"error": f"{ctr} pass timed out"}
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if p.returncode != 0 or not dbs:
                return {"error": f"{ctr} pass failed (rc {p.returncode})"}
            txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), "counters", dbs[0]], capture_output=True, text=True,
                                 timeout=60).stdout
            val = None
            for ln in txt.splitlines():
                c = [x.strip() for x in ln.split("|")]
                if len(c) > 5 and c[1].startswith("k_track_rows_fused") and c[2] == ctr:
                    val = float(c[4])
            if val is None:
                return {"error": f"{ctr}: no k_track_rows_fused row in the summary"}
            out[ctr] = val
        return out
    except Exception as ex:   # noqa: BLE001
        return {"error": str(ex)[:200]}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def spawn_command(n_gpus, argv, port):
    """the launcher line the driver itself uses for N > 1 (one rank per GPU, rendezvous on 127.0.0.1)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n_gpus, argv):
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = spawn_command(n_gpus, argv, port)
    if os.environ.get("BENCH_SPAWN_DRYRUN"):   # (CPU test hook: what would be launched)
        print(json.dumps({"spawn": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on these hosts
    return subprocess.call(cmd, env=env)


def render_video(cams, n_frames):
    """the rank's cameras' frames, rendered on the host cores in parallel BEFORE the HIP runtime is up (the workers are forked)"""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    jobs = [(c, f) for c in cams for f in range(n_frames)]
    workers = max(1, min(32, os.cpu_count() or 1, len(jobs)))
    if workers == 1:
        imgs = [_render_one(j) for j in jobs]
    else:
        with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("fork")) as ex:
            imgs = list(ex.map(_render_one, jobs, chunksize=max(1, len(jobs) // (4 * workers))))
    return {c: np.stack(imgs[i * n_frames:(i + 1) * n_frames]) for i, c in enumerate(cams)}


_SCENE = None


def _render_one(job):
    global _SCENE
    if _SCENE is None:
        _SCENE = build_scene()
    return _SCENE.render(*job)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ba-lag", type=int, default=int(os.environ.get("BENCH_BA_LAG", "0")),
                    help="key-frame intervals between a window's key frame and the frame its bundle adjustment is written back into the "
                         "map (RobustBundleRTS::output); 0 = min(max(N, 2), 4): window k is solved by rank k mod N; with two intervals the "
                         "solve's latency is off the frame loop's path also at N = 1 (1: the result of window k is in the map before "
                         "window k + 1 is built, the pose stream waits for it -- measured in DESIGN.md 6)")
    ap.add_argument("--key-every", type=int, default=KEY_EVERY, help="diagnostic: 0 disables the key-frame solves (not a valid bench line)")
    ap.add_argument("--only-solve", choices=["both", "joint", "intercam"], default="both",
                    help="diagnostic: run only one of the two key-frame solves (not a valid bench line)")
    ap.add_argument("--klt-cams-per-launch", type=int, default=int(os.environ.get("BENCH_KLT_CAMS_PER_LAUNCH", "0")),
                    help="cameras per persistent tracker launch; 0 = as many as are co-resident (all of the rank's in ONE launch)")
    ap.add_argument("--no-classify", action="store_true", help="diagnostic: skip mapPointsClassify (not a valid bench line)")
    ap.add_argument("--no-pose-update", action="store_true", help="diagnostic: skip the gate / dynamic test / BA write-back (not a valid bench line)")
    ap.add_argument("--no-mergability", action="store_true", help="diagnostic: skip staticCheckMergability (not a valid bench line)")
    ap.add_argument("--no-decide", action="store_true", help="diagnostic: skip the registration decision + refineMapPoint (not a valid bench line)")
    ap.add_argument("--hist", type=int, default=64, help="depth of the bounded walks (dynamic test, classification, re-triangulation) and of the "
                    "mergability walk's exact window; the whole-track verdict behind it is the running one (--hist-store)")
    ap.add_argument("--hist-store", type=int, default=4096, help="frames of pixels + poses kept behind the walks (what a dropped mergability "
                    "cache entry is rebuilt from)")
    ap.add_argument("--pixel-err-reading", choices=["variance", "std"], default=os.environ.get("BENCH_PIXEL_ERR_READING", "variance"),
                    help="Const::PIXEL_ERR_VAR = 10 handed to the covariance helpers as a variance (sqrt(10) px: the default, the reference's own "
                         "SL_Define.h:16) or as a standard deviation (10 px: rounds 1-4)")
    ap.add_argument("--active-search", type=int, default=0, help="diagnostic: 1 = also run the search half of activeMapPointsRegister (rounds 2-4)")
    ap.add_argument("--merge-every", type=int, default=50, help="bMerge frames: every n-th frame the static points' walks may unify two points "
                    "(the reference: 50, CoSLAMThread.cpp:117-118); 0: never (diagnostic)")
    ap.add_argument("--no-ncc", action="store_true", help="diagnostic: skip the inter-camera NCC matching leg (not a valid bench line)")
    ap.add_argument("--no-register", action="store_true", help="diagnostic: skip the map-point registration search (not a valid bench line)")
    ap.add_argument("--no-cxx-loop", action="store_true", help="skip the C++ frame loop (tools/cxx/frame_loop.bin, config.cxx_frame_loop)")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the upload-inclusive repetition of the loop (config.with_upload)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (cfg2, cfg5 BA, cfg5 KLT, reference-default KLT)")
    ap.add_argument("--setup-rounds", type=int, default=int(os.environ.get("BENCH_SETUP_ROUNDS", "-1")),
                    help="untimed set-up rounds of 4 key-frame intervals in front of the warm-up; -1 = N = 1: run for BENCH_SETUP_SECONDS, "
                         "N > 1: 12 (every rank must take the same number of steps)")
    ap.add_argument("--klt-xcd-placement", type=int, default=int(os.environ.get("BENCH_KLT_XCD", "1")),
                    help="1: the persistent tracker numbers its workgroups so that a camera lands on its own XCD; 0: cameras as grid rows")
    ap.add_argument("--live-pmc", type=int, default=int(os.environ.get("BENCH_LIVE_PMC", "1")),
                    help="1 (default): roofline.traffic / valu are collected by THIS run -- three rocprofv3 --pmc passes of the 8-camera KLT stage in "
                         "child processes behind the timed loops, a few seconds each, bounded by timeouts; a pass that fails falls back to the "
                         "committed profiles/r05_tracker_pmc.json and says so.  0: read the committed file")
    ap.add_argument("--keyframe-decision", type=int, default=0,
                    help="1: CoSLAM::IsReadyForKeyFrame + addKeyFrame's bookkeeping per frame on the device (reported in config.key_frame_decision; "
                         "the key frames stay on the fixed cadence)")
    ap.add_argument("--keyframe-drives", type=int, default=0,
                    help="1: the decision PLACES the key frames (LoopConfig.keyframe_drives; Python loop only: the C++ loop's leg is skipped); with "
                         "--keyframe-lag D > 0 the host acts on the decision of frame i - D read from pinned memory (no wait per frame; any N)")
    ap.add_argument("--keyframe-lag", type=int, default=0)
    ap.add_argument("--keyframe-ratio", type=float, default=0.93, help="m_mappedPtsReduceRatio (0.93 in the reference; never fires in this synthetic world)")
    ap.add_argument("--feature-chains", type=int, default=1,
                    help="1: MapPoint::pFeatures kept as feature references (stale features are views, re-linked tracks: SL_CoSLAM.cpp:775-779); "
                         "0: this frame's features on their own tracks (rounds 1-4)")
    ap.add_argument("--klt-cus", type=int, default=int(os.environ.get("BENCH_KLT_CUS", "0")),
                    help="> 0: the tracker stream confined to that many CU-mask bits (multiples of 32: the same CUs of every XCD); experiment")
    ap.add_argument("--pose-cus", type=int, default=int(os.environ.get("BENCH_POSE_CUS", "0")),
                    help="> 0: the pose stream confined to the last that many CU-mask bits; experiment")
    ap.add_argument("--klt-after-intracam", type=int, default=int(os.environ.get("BENCH_KLT_AFTER_INTRACAM", "0")),
                    help="1: the tracker of frame i + 1 starts behind frame i's intraCamEstimate (the pose solve runs on an empty chip)")
    ap.add_argument("--native-comm", type=int, default=1, help="N > 1: collectives issued by libcoslam_hip (RCCL behind the C-ABI) instead of torch.distributed")
    args = ap.parse_args()

    # `python bench.py --gpus N` started bare (no launcher: WORLD_SIZE unset) spawns its own N ranks, one per GPU, through
    # torch.distributed.run on 127.0.0.1 and a free port, and hands their exit code on.  Started BY a launcher the world size
    # must be what --gpus says: the line never reports an n_gpus that was not asked for.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; refusing to report "
                         "an n_gpus that differs from --gpus")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = max(world, 1)
    if N_CAMS % n_gpus != 0:
        raise SystemExit(f"bench.py: {N_CAMS} cameras do not shard over {n_gpus} GPUs (use 1, 2, 4 or 8)")
    cams_here = N_CAMS // n_gpus
    my_cams = list(range(rank * cams_here, (rank + 1) * cams_here))
    # the video of this rank's cameras (a closed camera path of N_FRAMES frames: no reversal, no jump, any length)
    t_r = time.perf_counter()
    if os.environ.get("BENCH_VIDEO") == "pingpong":
        frames = {c: np.concatenate([v, v[-2:0:-1]]) for c, v in render_video(my_cams, 24).items()}
    else:
        frames = render_video(my_cams, N_FRAMES)
    t_render = time.perf_counter() - t_r
    # N > 1: rank 0 also holds the other cameras' frames on the host, for the workload file of the C++ loop's ranks (config.cxx_frame_loop)
    cxx_frames = None
    if n_gpus > 1 and rank == 0 and not args.no_cxx_loop and os.environ.get("BENCH_VIDEO") != "pingpong":
        cxx_frames = dict(frames)
        cxx_frames.update(render_video([c for c in range(N_CAMS) if c not in my_cams], N_FRAMES))

    import torch
    import torch.distributed as dist

    import coslam_amd
    from coslam_amd.ba import BAWorkspace
    from coslam_amd.frameloop import FrameLoop, LoopConfig
    from coslam_amd.handback import handback_cams, handback_dev
    from coslam_amd.pose import intraCamEstimate_batch_dev

    # test hooks for exercising the N > 1 code path on a box with ONE GPU: BENCH_FORCE_DEVICE pins every rank to that
    # device, BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU).  Never set by the driver.
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    dist_backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(dist_backend, rank=rank, world_size=world)

    sc = build_scene()
    ic = build_ic_problem(sc)
    nc = cams_here
    n_map = len(sc.points)
    # ---- everything resident in HBM before the clock starts -------------------------------------
    video = {c: torch.from_numpy(frames[c]).to(dev) for c in my_cams}
    ke = args.key_every
    cfg = LoopConfig(n_cams=N_CAMS, W=W, H=H, levels=LEVELS, fw=FW, fh=FH, pts_stride=PTS_STRIDE, n_col_blk=N_COL_BLK, n_row_blk=N_ROW_BLK,
                     key_every=max(ke, 1), ba_lag=args.ba_lag, p_reg=P_REG, klt_cams_per_launch=max(args.klt_cams_per_launch, 0),
                     prefetch=os.environ.get("BENCH_PREFETCH", "1") != "0", with_pose_update=not args.no_pose_update,
                     with_classify=not args.no_classify, with_register=not args.no_register, with_mergability=not args.no_mergability,
                     with_ncc=not args.no_ncc, with_decide=not args.no_decide, merge_every=args.merge_every, hist=args.hist, hist_store=args.hist_store, with_active_search=bool(args.active_search), pixel_err_reading=args.pixel_err_reading, with_joint=args.only_solve != "intercam", with_intercam=args.only_solve != "joint",
                     native_comm=bool(args.native_comm), klt_cus=args.klt_cus, pose_cus=args.pose_cus, klt_xcd_placement=bool(args.klt_xcd_placement), feature_chains=bool(args.feature_chains), keyframe_decision=bool(args.keyframe_decision),
                     keyframe_drives=bool(args.keyframe_drives), keyframe_lag=args.keyframe_lag, keyframe_ratio=args.keyframe_ratio,
                     klt_after_intracam=bool(args.klt_after_intracam),
                     klt_fused=os.environ.get("BENCH_FORCE_DEVICE") is None or world == 1)   # (ranks sharing ONE GPU: test hook)
    try:
        loop = FrameLoop(cfg, sc, video, None if os.environ.get("BENCH_IC_PREBAKED") is None else ic, klt_config(), reg_covariances(n_map), rank=rank, world=world, device=local_rank,
                         dist_backend=dist_backend, associate=associate)
    except coslam_amd.CoslamHipError as ex:
        raise SystemExit(f"bench.py: the frame loop could not be set up ({ex})")
    trks, grp, klt_s, pose_s, ba_ws, ic_ws = loop.trks, loop.grp, loop.klt_s, loop.pose_s, loop.ba_ws, loop.ic_ws
    step = loop.step

    def barrier():
        loop.drain()      # the worker threads' queues are part of the timed work
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    loop.first_frame()

    # set-up, not warm-up: the window ring filled (every timed solve then has its 5 key frames = 40 cameras) and everything that
    # happens once per process behind us -- the BA workers capture and instantiate their graphs on first use, the runtime loads
    # each kernel's code object at its first launch -- whatever W the caller asks for
    import gc

    gc.collect()
    gc.disable()   # no collector pauses on the launching thread from here to the end of the timed region
    is_key = (lambda j: ke > 0 and j % ke == 0)
    n_done = 0     # frames enqueued so far; the frame sequence continues through set-up, warm-up and the timed region

    def run(n, upload=False):
        nonlocal n_done
        for _ in range(n):
            step(n_done + 1, is_key(n_done), upload)
            n_done += 1
            if os.environ.get("BENCH_DIGEST_EVERY_FRAME"):   # diagnostic: where two runs part ways (drains every frame)
                print(f"[digest rank {rank} frame {n_done}]", json.dumps(loop.digest_parts()), file=sys.stderr, flush=True)

    run(5 * max(ke, 1) + 1)
    barrier()
    # ... and the device out of its idle power state (a fresh process on a fresh box: +13 % on the driver's 20-step command,
    # profiles/r03_bench_lines.txt): the loop keeps running, same cadence, untimed; K and W are untouched.  Rounds of 4 key-frame
    # intervals, ending on a key-frame boundary.
    t_su = time.perf_counter()
    rounds = 0
    fixed_rounds = args.setup_rounds if args.setup_rounds >= 0 else (12 if world > 1 else -1)
    run((-n_done) % max(ke, 1))
    while (rounds < fixed_rounds) if fixed_rounds >= 0 else (time.perf_counter() - t_su < SETUP_SECONDS):
        run(4 * max(ke, 1))
        rounds += 1
        barrier()
    # the bMerge frames (every 50th: a few milliseconds each) fall into the timed region at their true rate whatever K is: the region
    # starts 25 frames behind one, so K steps hold round(K / 50) of them (K = 20: none, K = 50: one, K = 300: six) -- not "one if the
    # set-up loop's clock happened to stop there"
    if cfg.merge_every > 0 and ke > 0:
        run((cfg.merge_every // 2 - args.warmup - n_done) % cfg.merge_every)
    run(args.warmup)
    barrier()
    ba_ws.worker_stats(), [w.worker_stats() for w in loop.ic_wss]   # (reset: the sums below cover the timed region only)
    applied0 = loop.applied
    rv0 = loop.d_rv_counts.cpu().tolist()   # (the second visits' counters at the start of the timed region: the first frame's bootstrap is behind us)
    prof = None
    if os.environ.get("BENCH_PYPROFILE") and rank == 0:   # where the host thread's time goes (diagnostic: slows the loop)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t_begin = time.perf_counter()
    t_step_max, i_step_max, t_prev = 0.0, -1, t_begin
    for i in range(args.steps):
        # key frames: every KEY_EVERY-th frame of the sequence (K / KEY_EVERY solves of each kind in K frames, each on the rank that
        # owns it, all completed before the clock stops)
        run(1)
        t_now = time.perf_counter()
        if t_now - t_prev > t_step_max:
            t_step_max, i_step_max = t_now - t_prev, i
        t_prev = t_now
    t_host = time.perf_counter() - t_begin
    barrier()
    dt = time.perf_counter() - t_begin
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(45)
    wj = ba_ws.worker_stats()
    wis = [w.worker_stats() for w in loop.ic_wss]
    wi = [sum(w[0] for w in wis), sum(w[1] for w in wis), 0, max(w[3] for w in wis)]
    applied_timed = loop.applied - applied0
    dec_counts = None if not hasattr(loop, "_dec") else (loop._dec["cnt"].cpu().tolist(), int(loop._dec["cnt"][1].item()),   # (every registered point is refined: the count of one is the other's)
                                                         int(loop._dec["scr"][-4:].view(torch.int32).item()))
    digest = loop.digest() if os.environ.get("BENCH_STATE_DIGEST") else None
    map_in_use_timed_end = int(loop.d_mapcount.item())   # (as of the end of the timed region: the secondary legs run the loop on)
    if loop._marks is not None:
        print("[pose stream, GPU us per section]", json.dumps(loop.gpu_sections(first_frame=args.warmup), indent=1), file=sys.stderr)
    if loop._timing is not None:
        print("[frameloop host seconds by section]", {k: round(v, 4) for k, v in loop._timing.items()}, file=sys.stderr)
    n_timed_end = n_done
    solve_duty = {"what": "time the key-frame solves THIS RANK ran held their workspaces' streams inside the timed region (GPU clock, from the "
                          "moment the frame they wait for was done), against the region's length",
                  "joint_ba": {"solves": wj[0], "ms_total": wj[1], "ms_max": wj[3], "ms_parse_total": wj[4], "share_of_timed_region": wj[1] / (dt * 1e3)},
                  "inter_camera": {"solves": wi[0], "ms_total": wi[1], "ms_max": wi[3], "share_of_timed_region": wi[1] / (dt * 1e3),
                                   "workspaces_the_key_frames_rotate_over": loop.n_ic_workers}}
    with_upload = None
    if not args.no_upload_leg:
        # the same loop once more, the images coming from pinned host memory every frame (same key-frame cadence, same drain):
        # one pinned ring entry per frame, the cameras' images back to back, as capture threads writing into cs_pinned_alloc'd
        # memory would leave them -> ONE host-to-device copy per frame
        loop.h_frames = torch.from_numpy(np.stack([frames[c] for c in my_cams], axis=1).copy()).pin_memory()   # [frame][camera][H][W]
        loop.stage(n_done + 1)
        loop.stage(n_done + 2)
        run(args.warmup, upload=True)
        barrier()
        tu = time.perf_counter()
        run(args.steps, upload=True)
        barrier()
        dtu = time.perf_counter() - tu
        if world > 1:
            tmu = torch.tensor([dtu], dtype=torch.float64, device=dev)
            dist.all_reduce(tmu, op=dist.ReduceOp.MAX)
            dtu = float(tmu.item())
        with_upload = {"frames_per_s": args.steps / dtu, "ms_per_step": dtu / args.steps * 1e3,
                       "ratio_to_value": (args.steps / dtu) / (args.steps / dt),
                       "what": f"the same loop with every frame's {nc} x {W * H} B images copied from pinned host memory inside the "
                               "loop (cs_klt_group_stage_h: copy stream + ring of 3 device slots, two frames ahead of the tracker)"}
        loop.stage_slot.clear()
    ref_policy = None
    if world == 1 and not args.no_secondary and loop.out is not None:
        # SECONDARY: the reference's own request policy -- CoSLAM::requestForBA refuses a request while the previous bundle adjustment
        # is still running (src/app/SL_CoSLAM.cpp:1750-1755) -- instead of solving every key frame's window: same loop, same frames
        loop.drain()
        loop.skip_busy, w0, s0 = True, loop.n_windows, loop.n_skipped
        run(args.warmup)
        barrier()
        w1, s1 = loop.n_windows, loop.n_skipped
        tp = time.perf_counter()
        run(args.steps)
        barrier()
        dtp = time.perf_counter() - tp
        loop.skip_busy = False
        ref_policy = {"frames_per_s": args.steps / dtp, "ms_per_step": dtp / args.steps * 1e3, "ratio_to_value": (args.steps / dtp) / (args.steps / dt),
                      "windows_solved": loop.n_windows - w1, "requests_dropped_because_the_previous_solve_was_running": loop.n_skipped - s1,
                      "what": "the same loop with the reference's request policy: a key frame's window BA is requested only when no bundle "
                              "adjustment is running (CoSLAM::requestForBA, SL_CoSLAM.cpp:1750-1755); the headline solves EVERY window"}
    seq_reg = single_pass = None
    if world == 1 and not args.no_secondary and hasattr(loop, "_dec"):
        # SECONDARY: the registration of the current static points step for step as the reference runs it (camera loop after camera
        # loop, search + mergability + walks + refineMapPoint per loop: bit-identical to the reference's own run on its golden scenes,
        # tests/test_register_decide_gpu.py) instead of the headline's single pass: same loop, same frames, fewer steps
        loop.drain()
        loop.sequential_registration = True
        n_seq = max(args.steps // 3, 10)
        run(min(args.warmup, 10))
        barrier()
        tq = time.perf_counter()
        run(n_seq)
        barrier()
        dtq = time.perf_counter() - tq
        loop.sequential_registration = False
        seq_reg = {"frames_per_s": n_seq / dtq, "ms_per_step": dtq / n_seq * 1e3, "steps": n_seq, "ratio_to_value": (n_seq / dtq) / (args.steps / dt),
                   "loops_whose_sweeps_did_not_settle": int(loop._dec["scr"][-4:].view(torch.int32).item()),
                   "what": "the same loop with CoSLAM::currentMapPointsRegister reproduced step for step (8 camera loops of the static points, "
                           "then 8 of the dynamic ones; per loop a search, a mergability pass, the walks of that camera's points and a refine: "
                           "16 x the launches) -- the parity mode the headline's form is measured against: tools/r06_exact_vs_single.py runs both from "
                           "the same state, frame by frame; with the second visits' two rounds 450 of 450 compared frames are byte-identical "
                           "(profiles/r06_exact_vs_single.txt)"}
        # SECONDARY: the single pass ALONE (rounds 3-5's headline form): what the second visits cost
        rr = loop.cfg.revisit_rounds
        loop.cfg.revisit_rounds = 0
        n_sp = max(args.steps // 2, 20)
        run((-n_done) % max(ke, 1))
        barrier()
        tq = time.perf_counter()
        run(n_sp)
        barrier()
        dtq = time.perf_counter() - tq
        loop.cfg.revisit_rounds = rr
        single_pass = {"frames_per_s": n_sp / dtq, "ms_per_step": dtq / n_sp * 1e3, "steps": n_sp, "ratio_to_value": (n_sp / dtq) / (args.steps / dt),
                       "what": "the same loop without the second visits' rounds (LoopConfig.revisit_rounds = 0): the registration of rounds 3-5, which "
                               "parts from the reference's order within a few registering frames (1 / 8 / 18 frames from three start frames)"}
    kf_leg = None
    if world == 1 and not args.no_secondary and not args.keyframe_decision:
        # SECONDARY: the reference's key-frame DECISION per frame beside the same loop (CoSLAM::IsReadyForKeyFrame + addKeyFrame's key-pose
        # bookkeeping on the device, pinned: tests/golden/keyframe_golden.npz); the key frames themselves stay on the fixed cadence
        loop.drain()
        loop.enable_keyframe_decision(loop._frame_now, loop._dst_now)
        n_kf = max(args.steps // 2, 10)
        barrier()
        tq = time.perf_counter()
        run(n_kf)
        barrier()
        dtq = time.perf_counter() - tq
        kf_leg = dict(loop.keyframe_stats(), frames_per_s=n_kf / dtq, ratio_to_value=(n_kf / dtq) / (args.steps / dt),
                      fixed_cadence_key_frames_in_the_same_frames=n_kf // max(cfg.key_every, 1),
                      what="the same loop with cs_keyframe_ready_dev per frame: is a camera ready for a key frame (mapped points decreased below "
                           "0.93 of the last key pose's / view angle > 5 degrees / translation), and the key-pose state moved on as addKeyFrame does "
                           "when one has decreased (SL_CoSLAM.cpp:1269-1309); reported, not yet driving the window BA (its apply assumes equally "
                           "spaced key frames)")
        loop.kf = None
    gc.enable()
    coll_us = None
    if world > 1:
        try:
            loop.drain()
            coll_us = loop.measure_collectives()
        except Exception as ex:   # (a diagnostic: never the reason a bench line is lost)
            coll_us = {"error": str(ex)[:300]}
    replicas = None
    if world > 1:
        # ONE map held N times: every rank hashes the state all ranks must agree on after the last frame
        import hashlib

        dg = loop.digest()
        mine = torch.tensor(list(hashlib.sha256(dg.encode()).digest()[:8]), dtype=torch.int64)
        mine = mine.to(dev) if dist_backend == "nccl" else mine
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        replicas = {"ranks": world, "identical_map_records_and_poses_on_every_rank": bool(all(torch.equal(v, allv[0]) for v in allv)),
                    "what": "sha256 over map points, covariances, flags, every camera's hand-back records, track spans, feature types and "
                            "poses, compared across the ranks after the last frame"}
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    grp.synchronize()
    last = n_done & 1
    d_R, d_t = loop.d_R, loop.d_t
    n_live = [int((d.cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum()) for d in loop.d_dests[last]]
    pose_ok = loop.d_ok.cpu().numpy()[my_cams].tolist()
    from coslam_amd.pose import IntraCamPoseOption
    _opts = [IntraCamPoseOption.from_buffer_copy(loop.d_opt[c].cpu().numpy().tobytes()) for c in my_cams]
    pose_iters = [[o.nIterRW, o.verboseRW] for o in _opts]     # re-weighting rounds, LM steps over all rounds (last frame)
    pose_npts = loop.d_npts.cpu().numpy()[my_cams].tolist()
    # how far the device-resident poses are from the synthetic ground truth of the last frame (data-coupled pose leg)
    f_last = loop.vid(n_done)
    tl_ = d_t[last].cpu().numpy()
    pose_err = max(float(np.abs(tl_[c] - sc.pose(c, f_last)[1]).max()) for c in my_cams)
    n_pts0 = loop.n_pts0
    # gauge-free: every rank holds every camera's pose (its own: solved; the others': the frame's all-gather)
    from coslam_amd.synth import rig_error_vs_truth, umeyama
    rig_err = rig_error_vs_truth(sc, f_last, d_R[last].cpu().numpy(), tl_)
    # the map points this frame's static features USE (of the initial points: their true positions are known)
    st_, s2m_, iss_, fl_ = loop.d_state.cpu().numpy(), loop.d_slot2map.cpu().numpy(), loop.d_isstatic.cpu().numpy(), loop.d_mapflags.cpu().numpy()
    used_ = np.unique(s2m_[(st_ >= 0) & (s2m_ >= 0) & (iss_ != 0)])
    used_ = used_[(fl_[used_] & 3) == 0]
    u0_ = used_[used_ < n_pts0]
    map_err = None
    if len(u0_) >= 4:
        Mu_ = loop.d_map.cpu().numpy()[u0_]
        e_ = np.linalg.norm(Mu_ - sc.points[u0_], axis=1)
        s_, Ra_, ta_ = umeyama(Mu_, sc.points[u0_], True)
        ea_ = np.linalg.norm(s_ * Mu_ @ Ra_.T + ta_ - sc.points[u0_], axis=1)
        map_err = {"points_in_use_initial": int(len(u0_)), "points_in_use_new": int(len(used_) - len(u0_)), "raw_median": float(np.median(e_)),
                   "raw_p90": float(np.percentile(e_, 90)), "after_own_sim3_median": float(np.median(ea_)), "after_own_sim3_p90": float(np.percentile(ea_, 90)),
                   "what": "distance of the map points this frame's static features use from their true positions (the 7000 initial points: truth "
                           "known), raw and after the similarity that fits them best; most of it is DEPTH noise of points one camera sees over a "
                           "short baseline (1.5 cm / frame at 6-14 m): updateStaticPointPosition re-triangulates them from two views behind every BA"}
    win_info = st_j = None
    if loop.win is not None and loop.n_my_solves > 0:
        wC, wP, wO, _, wkf = loop.win.last_problem()
        ba_ws.set_sizes(wC, wP, wO)
        win_info = {"cameras": wC, "points": wP, "measurements": wO, "key_frames": wkf,
                    "what": "parsed on the device from the last 5 key frames' hand-back records and solved poses of ALL cameras "
                            "(cs_ba_window_*), the last window this rank solved"}
        st_j = ba_ws.download()[4]
    st_i = ic_info = None
    if loop.n_my_ic > 0:
        if loop.icam is not None:
            iC, iP, iO, iS, _ = loop.icam.last_problem()
            ic_ws.set_sizes(iC, iP, iO)
            ic_info = {"cameras": iC, "static_points_fixed": iS, "dynamic_points": iP - iS, "measurements": iO,
                       "what": "InterCamPoseEstimator::addMapPoints built on the device from the key frame's records of all cameras "
                               "(cs_ba_solve_intercam_async)"}
        st_i = ic_ws.download()[4]
    apply_info = None
    if loop.out is not None:
        cnt = loop.d_apply_counts.cpu().tolist()
        apply_info = {"what": "RobustBundleRTS::output() of every window solve, `lag` key-frame intervals after its key frame, on the LIVE map "
                              "of every rank (cs_ba_output_apply_dev: key poses into the pose history / the window ring, points into the map, "
                              "outlier points false, relaxation of the non-key frames up to the newest, updateNewPosesPoints)",
                      "lag_key_frame_intervals": loop.lag, "windows_applied_in_timed_region": applied_timed, "windows_applied": loop.applied,
                      "last": loop.last_apply, "apply_wait_errors": loop.out.wait_errors(), "static_points_retriangulated_last": cnt[0], "dynamic_points_retriangulated_last": cnt[1],
                      "points_set_false_last": cnt[2]}

    # ---- roofline of the dominant kernel: the persistent gain tracker of all cameras of this rank (one launch per frame).
    # Timed with HIP events on the stream it is launched on, over a replay of the same frames after the timed region.
    prefetch = cfg.prefetch
    img_ptrs, dest_ptrs, cnt_ptrs = loop.img_ptrs, loop.dest_ptrs, loop.cnt_ptrs
    roof = None

    def replay(n):
        nonlocal n_done
        for _ in range(n):
            f, fn = loop.vid(n_done + 1), loop.vid(n_done + 2)
            if prefetch:
                grp.prefetch_dev(img_ptrs[fn])
            grp.redetect_dev(img_ptrs[f], dest_ptrs[0], cnt_ptrs)
            grp.advanceFrame()
            n_done += 1

    in_loop_us = None
    if rank == 0 and world == 1:
        # the same kernel INSIDE the frame loop (every stream busy): 100 more frames of the whole loop with the tracker launch bracketed
        trks[0].set_profiling(True)
        run(20)
        trks[0].get_profile()
        run(100)
        barrier()
        p_in = trks[0].get_profile()
        trks[0].set_profiling(False)
        in_loop_us = p_in["tracker_us_total"] / max(p_in["frames"], 1) / max(p_in["launches_per_frame"], 1)
    if rank == 0:
        trks[0].set_profiling(True)
        replay(100)   # (also for a short --steps run: the first frames after the switch to profiling are slower)
        prof = trks[0].get_profile()
        trks[0].set_profiling(False)
        hw = 7 // 2
        # per feature per visited level two (2hw+2)^2 footprints of 6-byte texels, + 2 x 12 B feature I/O
        per_feature = LEVELS * 2 * (2 * hw + 2) ** 2 * 6 + 2 * 12
        alg_bytes = per_feature * N_FEAT * nc
        launches = max(prof["launches_per_frame"], 1)
        avg_us = prof["tracker_us_total"] / max(prof["frames"], 1) / launches
        ach = (alg_bytes / launches) / (avg_us * 1e-6) / 1e9
        traffic, traffic_src, valu = None, None, None
        pmc_file = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r05_tracker_pmc.json", "r04_tracker_pmc.json", "r03_tracker_pmc.json"))
                         if os.path.exists(q)), None)
        pj = json.load(open(pmc_file)) if (nc == N_CAMS and pmc_file) else None
        if pj is not None:   # HBM bytes per frame's worth of launches from the committed --pmc passes (8 cameras per launch there)
            traffic = pj["traffic_bytes_per_launch"] / launches
            traffic_src = os.path.relpath(pmc_file, ROOT) + " (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE: bytes of all 8 cameras / launches per frame)"
            if "valu_wave_insts_per_launch" in pj:
                # the second roofline: the kernel is bound by VALU issue, not by HBM.  A wave64 VALU instruction holds its SIMD's
                # issue port for 4 cycles; 1024 SIMDs.
                insts = pj["valu_wave_insts_per_launch"] / launches
                floor_us = insts * 4.0 / (1024 * pj.get("sclk_ghz", 2.4) * 1e3)
                valu = {"insts": insts, "floor_us": floor_us, "frac": floor_us / avg_us, "unit": "wave64 VALU instructions per launch",
                        "source": os.path.relpath(pmc_file, ROOT) + " (SQ_INSTS_VALU, separate --pmc pass)"}
        if args.live_pmc and pj is not None:
            # the same counters collected NOW (opt-in: three rocprofv3 --pmc passes of the 8-camera KLT stage in child processes, ~10 s each, the
            # way MI355X_MICROARCH.md prescribes: separate passes, --kernel-trace only beside them); a pass that fails leaves the committed figure
            live = live_tracker_pmc()
            if "error" not in live:
                traffic = (2.0 * live["FETCH_SIZE"] * 1024 + live["WRITE_SIZE"] * 1024) / launches
                traffic_src = "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of tools/pmc_klt.py (x 2 on the fetch counter: gfx950 tallies 128-byte requests at 64 B)"
                if valu is not None and "SQ_INSTS_VALU" in live:
                    insts = live["SQ_INSTS_VALU"] / launches
                    floor_us = insts * 4.0 / (1024 * pj.get("sclk_ghz", 2.4) * 1e3)
                    valu = {"insts": insts, "floor_us": floor_us, "frac": floor_us / avg_us, "unit": "wave64 VALU instructions per launch",
                            "source": "measured in this run (SQ_INSTS_VALU, its own --pmc pass)"}
            else:
                traffic_src += "; --live-pmc failed: " + live["error"]
        fused_kernel = prof["launches_per_frame"] <= N_CAMS
        roof = {"bound": "hbm", "kernel": "k_track_rows_fused" if fused_kernel else "k_track_rows_pass",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes / launches,
                "avg_launch_us": avg_us, "avg_launch_us_in_loop": in_loop_us,
                "frac_in_loop": None if not in_loop_us else (alg_bytes / launches) / (in_loop_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "launches_per_frame": launches, "frames_timed": prof["frames"],
                "cameras_per_launch": nc / launches, "valu": valu}

    # ---- secondary key: cfg2 (BASELINE.json configs[1]) = ONE camera on the GPU, KLT + hand-back + pose per frame, no
    # key-frame solves; same kernels through the single-handle entry points.  Not the headline; kept for continuity.
    cfg2 = None
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        for t in trks:
            t.set_cu_count(256)   # (the headline loop may have budgeted the tracker for fewer cameras per launch)
        n2 = min(args.steps, 200)
        k0 = trks[0]
        hb1 = handback_cams([dict(dest=loop.d_dests[0][0].data_ptr(), K=loop.d_K1.data_ptr(), kud=loop.d_kud.data_ptr(), mapPts=loop.d_map.data_ptr(),
                                  slot2map=loop.d_slot2map[0].data_ptr(), trackSpan=loop.d_trackspan[0].data_ptr(), xy=loop.d_xy[0].data_ptr(),
                                  state=loop.d_state[0].data_ptr(), Ms=loop.d_Ms[0].data_ptr(), ms=loop.d_ms[0].data_ptr(),
                                  sel=loop.d_sel[0].data_ptr(), npts=loop.d_npts[0:1].data_ptr(), opt=loop.d_opt[0].data_ptr())])
        pose1 = torch.cuda.Event()

        def step1():
            nonlocal n_done
            f, fn = loop.vid(n_done + 1), loop.vid(n_done + 2)
            klt_s.wait_event(pose1)
            if prefetch:
                k0.prefetch_dev(img_ptrs[fn][0])
            k0.redetect_dev(img_ptrs[f][0], dest_ptrs[0][0], cnt_ptrs[0])
            k0.advanceFrame()
            loop.klt_done[0].record(klt_s)
            pose_s.wait_event(loop.klt_done[0])
            handback_dev(pose_s.cuda_stream, hb1, N_FEAT, W, H, N_COL_BLK, N_ROW_BLK, PTS_STRIDE, device=local_rank, frame=n_done + 1)
            src, dst = n_done & 1, (n_done + 1) & 1
            intraCamEstimate_batch_dev(pose_s.cuda_stream, 1, PTS_STRIDE, loop.d_K.data_ptr(), d_R[src].data_ptr(), d_t[src].data_ptr(),
                                       loop.d_npts.data_ptr(), 0, loop.d_Ms.data_ptr(), loop.d_ms.data_ptr(), 10.0, d_R[dst].data_ptr(),
                                       d_t[dst].data_ptr(), loop.d_opt.data_ptr(), loop.d_ok.data_ptr(), device=local_rank)
            pose1.record(pose_s)
            n_done += 1

        pose1.record(pose_s)
        for _ in range(20):
            step1()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(n2):
            step1()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        cfg2 = {"workload": "cfg2: 1 camera 640x480 x 2000 slots, KLT (redetect, prefetch) + hand-back + intraCamEstimate "
                            "per frame, no key-frame solves", "camera_frames_per_s": n2 / dt2, "frames": n2}

    # ---- secondary key: the sliding-window BA of cfg5 (BASELINE.json configs[4]: 4 cameras x 30 key frames = 120 poses of which 8
    # fixed, 5000 points, every point in every key frame: 600 k measurements, reduced system of order 672), one full robust solve
    cfg5 = None
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        from coslam_amd.synth import csr_of_problem, make_ba_problem

        pr5 = make_ba_problem(n_cams=120, n_pts=5000, W=1920, H=1080, noise=0.3, outlier_frac=0.01, outlier_mag=40.0, n_cams_con=8,
                              n_pts_con=2, seed=55)
        p5, c5, x5 = csr_of_problem(pr5)
        ws5 = BAWorkspace(local_rank)
        ws5.upload(pr5["Ks"], pr5["Rs0"], pr5["ts0"], pr5["pts0"], p5, c5, x5)
        d5 = [torch.from_numpy(pr5[k].reshape(-1).copy()).to(dev) for k in ("Rs0", "ts0", "pts0")]
        s5 = torch.cuda.current_stream().cuda_stream
        for rep in range(3):   # the first solve allocates the workspace of the large-problem kernels and captures the graph
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            ws5.solve_dev(s5, d5[0].data_ptr(), d5[1].data_ptr(), d5[2].data_ptr(), 8, 2, 6.0, 2, 5)
            torch.cuda.synchronize()
            dt5 = time.perf_counter() - t5
        st5 = ws5.download()[4]
        cfg5 = {"workload": "cfg5 BA: C=120 (8 fixed) x 5000 pts x 600 k meas, order 672, maxIter 2 / inner 5", "ms_per_solve": dt5 * 1e3,
                "lm_steps": st5.nIterTotal, "us_per_lm_step": dt5 * 1e6 / max(st5.nIterTotal, 1), "cost0": st5.cost0, "cost": st5.cost,
                "outliers": st5.nOutliers}
        ws5.close()

    # ---- secondary key: NewMapPtsNCC's tail on a LOADED run: the reference's own scenes (tests/golden/newpts_golden.npz, made by the
    # reference's featTracksFromMatches + reconstructTracks + decidePointType), the one with the most tracks, through
    # cs_newpts_from_pairs_dev.  (The frame loop's runs above find few pairs on the synthetic scene; this one has every track to
    # reconstruct.)
    newpts_loaded = None
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        try:
            newpts_loaded = _newpts_loaded_leg(dev)
        except Exception as e:   # (the fixture is data under tests/golden; a missing file must not cost the line)
            newpts_loaded = {"error": repr(e)}

    # ---- secondary keys: KLT stages of other configurations as one camera group on one GPU: cfg5 (BASELINE.json configs[4]: 4 cameras
    # 1920 x 1080 x 5000 slots) and the headline's 8 cameras with the REFERENCE-DEFAULT parameter set (nLevels 6,
    # levelSkip 2, 12 iterations, 5 x 5 window -- v3d_gpuklt.h:181-191 -- with gain as CoSLAM runs it)
    def klt_stage(Wk, Hk, Lk, FWk, FHk, Ck, conf, frames_k, n_time=40):
        t5s = []
        for _ in range(Ck):
            t = coslam_amd.KLT_SequenceTracker(conf, device=local_rank)
            t.allocate(Wk, Hk, Lk, FWk, FHk)
            t.set_concurrent_handles(Ck)   # (the headline's trackers are still alive but idle: only this group's launches overlap)
            t5s.append(t)
        g5 = coslam_amd.KLT_TrackerGroup(t5s)
        g5.set_stream(klt_s.cuda_stream)
        dd5 = [torch.zeros(FWk * FHk * 5, dtype=torch.int32, device=dev) for _ in range(Ck)]
        cc5 = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(Ck)]
        dp5, cp5 = [d.data_ptr() for d in dd5], [c.data_ptr() for c in cc5]
        nf = int(frames_k[0].shape[0])
        g5.detect_dev([f[0].data_ptr() for f in frames_k], dp5, cp5)
        g5.advanceFrame()

        def frame5(i):
            a, b2 = (i + 1) % nf, (i + 2) % nf
            g5.prefetch_dev([f[b2].data_ptr() for f in frames_k])
            g5.redetect_dev([f[a].data_ptr() for f in frames_k], dp5, cp5)
            g5.advanceFrame()

        for i in range(8):
            frame5(i)
        g5.synchronize()
        t5s[0].set_profiling(True)
        tk = time.perf_counter()
        for i in range(n_time):
            frame5(8 + i)
        g5.synchronize()
        dtk = time.perf_counter() - tk
        p5 = t5s[0].get_profile()
        t5s[0].set_profiling(False)
        live5 = [int((d.cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum()) for d in dd5]
        hwk = conf.windowWidth // 2
        visited = len(range(Lk - 1, -1, -(conf.levelSkip if conf.levelSkip > 0 else Lk - 1)))
        per_feat = visited * 2 * (2 * hwk + 2) ** 2 * 6 + 2 * 12
        trk_us = p5["tracker_us_total"] / max(p5["frames"], 1)
        res = {"frames_per_s": n_time / dtk, "camera_frames_per_s": Ck * n_time / dtk, "us_per_frame": dtk / n_time * 1e6,
               "tracker_stage_us": trk_us, "tracker_launches_per_frame": p5["launches_per_frame"],
               "tracker_algorithmic_GBps": per_feat * FWk * FHk * Ck / (trk_us * 1e-6) / 1e9, "live_features": live5,
               "hbm_frac_per_launch": per_feat * FWk * FHk * Ck / (trk_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        g5.close()
        for t in t5s:
            t.close()
        return res

    cfg5_klt = ref_default = None
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        from coslam_amd.synth import Scene as _Scene

        sc5 = _Scene(4, 1920, 1080, 12000, seed=0xC051A + 5, loop_period=6)
        fr5 = [torch.from_numpy(np.stack([sc5.render(c, f) for f in range(6)])).to(dev) for c in range(4)]
        conf5 = coslam_amd.KLT_SequenceTrackerConfig(nIterations=10, nLevels=4, levelSkip=1, windowWidth=7, trackWithGain=1, minCornerness=3000.0,
                                                     convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=8)
        cfg5_klt = dict(klt_stage(1920, 1080, 4, 100, 50, 4, conf5, fr5),
                        workload="cfg5 KLT: 4 cameras 1920x1080 x 5000 slots (100x50) as one camera group on one GPU, 4 levels, 7x7, 10 it/level "
                                 "with gain, redetect + prefetch per frame",
                        pmc="profiles/r06_cfg5_klt_pmc.json (2 + 2 cameras per launch, a camera on four XCDs: traffic 2.56 x the algorithmic bytes -- 128-byte fetches of sparse 72-byte patch rows --, VALU issue 0.52 of the launch; round 5, 3 + 1 unplaced: 2.85 x, 0.47)")
        del fr5
        confd = coslam_amd.KLT_SequenceTrackerConfig(trackWithGain=1, minCornerness=3000.0, SSD_Threshold=20000.0, minDistance=4)
        ref_default = dict(klt_stage(W, H, 6, FW, FH, N_CAMS, confd, [video[c] for c in range(N_CAMS)]),
                           workload="the headline's 8 cameras 640x480 x 2000 slots with the reference's default KLT parameters "
                                    "(v3d_gpuklt.h:181-191: 6 levels, levelSkip 2, 12 iterations, 5x5 window; with gain), KLT stage only: "
                                    "redetect + prefetch per frame")

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cores = os.cpu_count() or 1
        nt = min(cores, N_CAMS)
        joint = build_joint_problem(sc)
        v1, n1, dt1, js1 = cpu_baseline(sc, frames, joint, ic, 1, 12.0, not args.no_register, True, not args.no_ncc, args.pixel_err_reading)
        vN, nN, dtN, jsN = cpu_baseline(sc, frames, joint, ic, nt, 12.0, not args.no_register, True, not args.no_ncc, args.pixel_err_reading) if nt > 1 else (v1, n1, dt1, js1)
        cpu = {"value": vN, "unit": "frames/s", "cores": nt, "kind": "port",
               "sample": f"{nN} frames of the same 8-camera workload on {nt} threads (cameras in parallel) in {dtN:.1f} s; "
                         f"{n1} frames on 1 thread in {dt1:.1f} s (oracle/: C restatement, gcc -O2); host has {cores} cores.  Legs: KLT, "
                         "hand-back, intra-camera pose, register search + mergability, pose update gate + dynamic test + classify, "
                         "NCC blocks + epipolar/NCC matrix every 4th frame, per key frame the joint BA of the window PARSED from the last 5 key "
                         "frames' records like the GPU's (numpy parse; the pre-baked problem only until 5 key frames exist) + pose graph + the "
                         "update behind it + inter-camera BA; the registration decision (C) + refineMapPoint; the new-map-point match / reconstruct / "
                         "decidePointType tail in C on a scratch copy of the map (the CPU loop's map does not grow)",
               "joint_problem_last_parsed": dict(zip(("cameras", "points", "measurements"), jsN[:3])),
               "new_map_point_tail_share_of_the_figure": jsN[3],
               "value_1_thread": v1, "host_cores": cores}

    # ---- the same loop driven from C++ through the C-ABI only (north_star: "Host stays C++"): tools/cxx/frame_loop.cpp, its
    # own process, the workload handed over as a file; same steps / warm-up / key-frame cadence / drain
    cxx = None
    cxx_args = [str(args.steps), str(args.warmup), str(args.klt_cams_per_launch), str(loop.lag), str(n_timed_end - args.steps + 1)]   # (the same
    # stretch of the sequence as the timed region above)
    cxx_env = dict(os.environ, COSLAM_PIXEL_ERR_STD="1" if args.pixel_err_reading == "std" else "0", HSA_KERNARG_POOL_SIZE=str(64 << 20))
    cxx_exe = os.path.join(ROOT, "tools", "cxx", "frame_loop.bin")
    if rank == 0 and n_gpus == 1 and not args.no_cxx_loop and ke == KEY_EVERY and not args.keyframe_drives:
        import subprocess
        import tempfile

        if os.path.exists(cxx_exe):
            torch.cuda.synchronize()
            with tempfile.TemporaryDirectory() as td:
                wl = os.path.join(td, "workload.bin")
                export_workload(wl, sc, frames, build_joint_problem(sc), ic, args.klt_cams_per_launch)
                pr = subprocess.run([cxx_exe, wl] + cxx_args, capture_output=True, text=True, timeout=600, env=cxx_env)
            if pr.returncode == 0 and pr.stdout.strip().startswith("{"):
                cxx = json.loads(pr.stdout.strip().splitlines()[-1])
                cxx["what"] = ("tools/cxx/frame_loop.cpp: the headline loop from C++ through include/coslam_hip.h only (no Python, no "
                               "torch), own process, images resident in HBM")
            else:
                cxx = {"error": (pr.stderr or pr.stdout)[-400:]}
        else:
            cxx = {"error": "tools/cxx/frame_loop.bin not built (python -c 'import __graft_entry__ as g; g.build()')"}
    elif n_gpus > 1 and not args.no_cxx_loop and ke == KEY_EVERY and not args.keyframe_drives:
        # N > 1, host stays C++: every rank starts ITS rank of tools/cxx/frame_loop.bin (one process per GPU; the ranks find each other
        # through cs_comm_unique_id left in a file by rank 0, ncclCommInitRank inside the library) on the workload file rank 0 wrote.  A
        # diagnostic leg: bounded by a timeout, never the reason a bench line is lost.
        import shutil
        import subprocess

        td = os.path.join("/tmp", f"coslam_cxx_{os.environ.get('MASTER_PORT', '0')}_{n_gpus}")
        cdev = dev if dist_backend == "nccl" else torch.device("cpu")   # (where the backend takes its tensors)
        ok_here = torch.tensor([1 if os.path.exists(cxx_exe) else 0], dtype=torch.int32, device=cdev)
        try:
            if rank == 0:
                shutil.rmtree(td, ignore_errors=True)
                os.makedirs(td)
                if cxx_frames is not None:
                    export_workload(os.path.join(td, "workload.bin"), sc, cxx_frames, build_joint_problem(sc), ic, args.klt_cams_per_launch)
                else:
                    ok_here[0] = 0
        except Exception:   # noqa: BLE001
            ok_here[0] = 0
        dist.all_reduce(ok_here, op=dist.ReduceOp.MIN)   # (also the barrier behind the export)
        if int(ok_here.item()) == 1:
            torch.cuda.synchronize()
            env = dict(cxx_env, COSLAM_COMM_ID_FILE=os.path.join(td, "id"))
            env.pop("COSLAM_FORCE_DEVICE", None)
            if os.environ.get("BENCH_FORCE_DEVICE") is not None:
                # test hook (ranks sharing ONE GPU, tests/test_bench_contract_gpu.py): RCCL refuses two ranks on a device -- the library's
                # test transport and the launch-per-pass tracker instead
                env.update(COSLAM_FORCE_DEVICE=os.environ["BENCH_FORCE_DEVICE"], COSLAM_COMM=f"host:/coslam_bench_{os.environ.get('MASTER_PORT', '0')}",
                           COSLAM_KLT_FUSED="0")
            try:
                pr = subprocess.run([cxx_exe, os.path.join(td, "workload.bin")] + cxx_args, capture_output=True, text=True, timeout=240, env=env)
                mine = json.loads(pr.stdout.strip().splitlines()[-1]) if pr.returncode == 0 and pr.stdout.strip().startswith("{") else \
                    {"error": (pr.stderr or pr.stdout)[-400:]}
            except Exception as ex:   # noqa: BLE001  (timeout: a rank that never arrived)
                mine = {"error": str(ex)[:300]}
            # the ranks' digests must agree; the rate is the slowest rank's
            try:
                dgv, msv = (int(mine["digest"], 16) >> 1, int(1e6 * float(mine["ms_per_step"]))) if "digest" in mine else (-1, -1)
            except Exception:   # noqa: BLE001
                dgv, msv = -1, -1
            dg = torch.tensor([dgv, msv], dtype=torch.int64, device=cdev)
            alld = [torch.zeros_like(dg) for _ in range(world)]
            dist.all_gather(alld, dg)
            if rank == 0:
                cxx = dict(mine)
                if "error" not in mine and all(int(v[0]) >= 0 for v in alld):
                    ms = max(int(v[1]) for v in alld) / 1e6
                    cxx.update(ms_per_step=ms, frames_per_s=1e3 / ms, ranks=world,
                               identical_digest_on_every_rank=bool(all(int(v[0]) == int(alld[0][0]) for v in alld)),
                               what="tools/cxx/frame_loop.cpp at N > 1: one C++ process per GPU, cameras sharded like the Python loop's, the "
                                    "collectives through the library's own cs_comm_* / cs_exchange_* (RCCL); the slowest rank's rate")
                else:
                    cxx["ranks_that_failed"] = [q for q, v in enumerate(alld) if int(v[0]) < 0]
                shutil.rmtree(td, ignore_errors=True)
        elif rank == 0:
            cxx = {"error": "the C++ loop's ranks were not started (binary or workload file missing on a rank)"}

    if rank == 0:
        reg_out, d_mapflags = loop.reg_out, loop.d_mapflags
        lc = slice(my_cams[0], my_cams[-1] + 1)
        # north_star: "host stays C++".  `value` is the C++ frame loop's rate (tools/cxx/frame_loop.cpp: the same W warm-up + K timed steps over the
        # same stretch of the sequence, barrier + device synchronisation on both sides, at N > 1 the slowest rank) whenever that process ran to
        # its end (and, N > 1, every rank ended in the same state); the Python loop timed above -- the same library calls from ctypes -- is
        # config.python_frame_loop, and the fallback when the binary is missing or failed.
        py_loop = {"frames_per_s": args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                   "what": "coslam_amd/frameloop.py: the same loop driven from Python (ctypes), timed in this process per the bench contract"}
        cxx_ok = (isinstance(cxx, dict) and "error" not in cxx and cxx.get("frames_per_s") and cxx.get("steps") == args.steps and
                  cxx.get("pose_ok", True) and (n_gpus == 1 or cxx.get("identical_digest_on_every_rank")))
        value, ms_step = (float(cxx["frames_per_s"]), 1e3 / float(cxx["frames_per_s"])) if cxx_ok else (py_loop["frames_per_s"], py_loop["ms_per_step"])
        out = {
            "metric": "frames/sec for track+local-BA loop, 8 cams 640x480 x 2000 feats (one frame = all 8 cameras)",
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32 (KLT, f16 pyramid storage) + f64 (pose, BA)",
            "data": "synthetic",
            "config": {"workload": "8 cams 640x480 x 2000 KLT slots (50x40), 4-level pyramid, 7x7 window, 10 it/level with "
                                   "gain, redetect every frame; on-device hand-back + intraCamEstimate of all 8 cameras "
                                   "every frame (fed by the tracker's output)"
                                   + ("" if loop.pose_upd is None else ", then poseUpdate3D's Mahalanobis gate + seqTriangulate refinement of the "
                                      "map points, the dynamic-point test (64-frame history) and mapPointsClassify") + "; currentMapPointsRegister "
                                   "every frame over the frame's CURRENT map points (a list built on the device: every point with a feature of this "
                                   "frame, new ones included; search x 8 cams x 2000 slots, staticCheckMergability over WHOLE tracks as a running "
                                   "verdict, the decision settled in one launch, refineMapPoint, then the reference's SECOND VISITS of the points that registered -- two rounds of list / search / mergability / walks / refine over just those points" + (" over feature references -- MapPoint::pFeatures as the reference holds them: stale features are views, a re-registered point's old chain is linked behind the new feature" if cfg.feature_chains else "") + "), every 50th frame with bMerge (checkUnify); "
                                   "activeMapPointsRegister's search is not run (its attach loop is unreachable in the reference: "
                                   "tests/cxx/ref_active_test.cpp); every 4th frame the NCC matching of the consecutive camera pairs (F from the "
                                   f"poses just solved) -> new map points; every {KEY_EVERY}th frame: joint local BA C=40 (16 fixed), "
                                   "parsed on the device from the last 5 key frames' tracked features and poses"
                                   + (f" (last: {win_info['points']} pts x {win_info['measurements']} meas)" if win_info else "")
                                   + f", maxIter 2 / inner 10, its result written back into the LIVE map, pose history and window {loop.lag} "
                                   "key-frame interval(s) later (key poses, points, outlier points false, pose-graph relaxation of the non-key "
                                   "frames, updateNewPosesPoints), and inter-camera solve C=8 free, built on the device from the "
                                   "frame's records (the block-voted static features' map points fixed, <= 61 dynamic points), sigma 6, 3 x 40; N>1: cameras sharded 8/N per "
                                   "GPU, one all-gather of features+pose per frame, every rank replays the other cameras' hand-back and the map "
                                   "update (ONE map held N times, bit-identical), window k solved by rank k mod N and its packed result broadcast; "
                                   "KLT minDistance 4 (the reference's default is 8, SL_GlobParam.cpp:29: 4 keeps all 2000 slots alive on this "
                                   "scene); Const::PIXEL_ERR_VAR = 10 handed to the covariance helpers as a " + args.pixel_err_reading +
                                   " (DESIGN.md 5.1)",
                       "cameras": N_CAMS, "cameras_per_gpu": nc, "camera_frames_per_s": N_CAMS * value,
                       "video": {"frames": N_FRAMES, "what": "closed camera path (coslam_amd.synth.Scene loop_period): never reverses, never jumps",
                                 "host_render_s": t_render},
                       "frames_enqueued_until_end_of_timed_region": n_timed_end,
                       "live_features_last_frame": n_live, "pose_ok": pose_ok, "pose_correspondences": pose_npts, "pose_rounds_and_lm_steps": pose_iters,
                       "pose_translation_error_vs_truth": pose_err, "rig_error_vs_truth": rig_err, "map_error_vs_truth": map_err,
                       "joint_ba_from_window": loop.win is not None,
                       "joint_ba_problem": win_info,
                       "joint_ba_last": None if st_j is None else {"lm_steps": st_j.nIterTotal, "outliers": st_j.nOutliers,
                                                                  "cost0": st_j.cost0, "cost": st_j.cost},
                       "intercam_last": None if st_i is None else {"lm_steps": st_i.nIterTotal, "outliers": st_i.nOutliers, "cost0": st_i.cost0,
                                                                  "cost": st_i.cost},
                       "intercam_problem": ic_info, "ba_output": apply_info, "state_digest": digest, "replicas": replicas,
                       "frame_front_prefetch": bool(prefetch), "secondary_cfg2": cfg2, "secondary_cfg5_ba": cfg5, "secondary_cfg5_klt": cfg5_klt,
                       "secondary_reference_default_klt": ref_default, "secondary_newpts_loaded": newpts_loaded,
                       "register_candidates_last_frame": None if args.no_register else
                       {"current_points_listed": int(loop.d_curcount.item()), "list_cap": P_REG, "current_points_beyond_the_cap_all_frames": int(loop.d_curoverflow.item()),
                        "candidates": int((reg_out["slot"][:, lc] >= 0).sum().item()),
                        "mergeable_over_the_whole_track": None if loop.pose_upd is None else int(((loop.d_mergeable[:, lc] == 1) & (reg_out["slot"][:, lc] >= 0)).sum().item()),
                        "not_mergeable": None if loop.pose_upd is None else int(((loop.d_mergeable[:, lc] == 0) & (reg_out["slot"][:, lc] >= 0)).sum().item()),
                        "unjudged_track_older_than_the_store": None if loop.pose_upd is None else int(((loop.d_mergeable[:, lc] == 2) & (reg_out["slot"][:, lc] >= 0)).sum().item()),
                        "running_verdict": None if loop.pose_upd is None else dict(zip(
                            ("cache_hits", "full_tail_walks", "verdicts_unjudged", "tail_terms_evaluated"), loop.d_merge_counts.cpu().tolist()),
                            frames=n_timed_end, window_frames=cfg.hist, store_frames=cfg.hist_store, tol_pix=cfg.merge_tol_pix,
                            cache_check=loop.verdict_check(),
                            what="staticCheckMergability over WHOLE tracks (reference SL_CoSLAM.cpp:714-729): the newest 64 frames of a candidate's "
                                 "track walked every frame as they stand, the verdict over the older ones cached per (map point, camera) and extended "
                                 "by one term per frame (cs_register_mergability_running_dev); counts summed over the whole run"),
                        "active_search": "off: the reference's activeMapPointsRegister cannot attach (numVisCam == 0 on actMapPts, SL_CoSLAM.cpp:1114; "
                                         "tests/cxx/ref_active_test.cpp)" if not cfg.with_active_search else "on (diagnostic)"},
                       "key_frame_decision": loop.keyframe_stats() or "not run in this line (--keyframe-decision 1; secondary_keyframe_decision has a run)",
                       "feature_references": None if getattr(loop, "d_fref", None) is None else dict(zip(
                           ("tracked_on", "first_features", "re_linked_behind_an_older_feature", "links_dropped_pool_full", "detached"),
                           loop.d_fref_counts.cpu().tolist()),
                           stale_now=int(((loop.d_fref[:, :, 0] >= 0) & (loop.d_fref[:, :, 1] < loop._frame_now)).sum().item()),
                           linked_segments_per_camera=loop.pose_upd.segment_counts()[0].tolist(),
                           what="MapPoint::pFeatures as the reference holds them (cs_feat_ref): a camera that lost a point keeps its last feature "
                                "as a view of refineMapPoint / updateNewPosesPoints, a point registered to a new track where it held an older "
                                "feature gets the old chain linked behind it (SL_CoSLAM.cpp:775-779); counts summed over the whole run"),
                       "register_decision": None if dec_counts is None else dict(zip(
                           ("features_attached_last_frame", "points_regged_last_frame", "sweeps_last_frame", "converged"), dec_counts[0]),
                           points_refined_last_frame=dec_counts[1],
                           frames_whose_sweeps_did_not_settle=dec_counts[2],
                           of_which_barrier_timeouts=int(loop._dec["scr"][-8:-4].view(torch.int32).item()),
                           second_visits=dict(zip(("features_attached", "registrations", "conflicts_counted", "rounds_unsettled"), loop.d_rv_counts.cpu().tolist()),
                                              since_the_timed_region_began=dict(zip(("features_attached", "registrations", "conflicts_counted", "rounds_unsettled"),
                                                                                    [a - b for a, b in zip(loop.d_rv_counts.cpu().tolist(), rv0)])),
                                              rounds_per_frame=loop.cfg.revisit_rounds, points_beyond_the_list=int(loop.d_rv_listcounts[1].item()) + int(loop.d_rvcounts[loop.cfg.revisit_rounds].item()),
                                              what="the reference visits a point that registered AGAIN, refined, in its next camera's loop "
                                                   "(SL_CoSLAM.cpp:864-869, :889-893): played behind the single pass in rounds over just those points "
                                                   "(cs_register_revisit_*); counts over the whole run of this process.  tools/r06_exact_vs_single.py / "
                                                   "profiles/r06_exact_vs_single.txt: with two rounds 450 of 450 compared frames end in the map of the "
                                                   "reference's step-for-step order, byte for byte; conflicts_counted = visits whose outcome the reference's "
                                                   "order would have changed and the rounds could not take back (0 outside the first frame's bootstrap)"),
                           merge=None if not hasattr(loop, "_dec") else dict(
                               every_frames=loop.cfg.merge_every, bmerge_frames=loop.n_merge_frames,
                               what="every 50th frame the static points' walks run with bMerge (CoSLAMThread.cpp:117-118): one after the other, "
                                    "checkUnify at a feature of another static point, the two unified on a yes (cs_register_decide_merge_dev)",
                               **dict(zip(("features_attached_last_bmerge_frame", "points_registered_last_bmerge_frame",
                                           "points_unified_away_last_bmerge_frame", "check_unify_calls_last_bmerge_frame"),
                                          loop._dec["mcnt"].cpu().tolist()))),
                           what="currentMapPointsRegister's decisions (bMerge false) -- curStaticPointsRegInGroup and, behind it, "
                                "curDynamicPointsRegInGroup on the certainly dynamic points -- over the search + mergability tables of all cameras "
                                "(cs_register_decide_kinds_dev, kinds 3: the sequential first-claimant rule resolved exactly), then refineMapPoint "
                                "of the points that gained a feature (cs_refine_map_points_dev)"),
                       "pose_update": None if loop.pose_upd is None else {
                           "what": "poseUpdate3D's gate + seqTriangulate over all static mapped features and detectDynamicFeaturePoints over "
                                   "all unmapped / dynamic tracks, every frame, one launch for ALL cameras (cs_pose_update_frame_dev)",
                           "history_frames": loop.pose_upd.frames, "map_points_uncertain": int((d_mapflags & 4).ne(0).sum().item()),
                           "map_points_classify": None if args.no_classify else {
                               "what": "CoSLAM::mapPointsClassify(12.0) every frame behind the gate (cs_map_points_classify_dev)",
                               "examined_last_frame": int(loop.d_cls_counts[0].item()), "became_false_last_frame": int(loop.d_cls_counts[1].item()),
                               "map_points_false": int((d_mapflags & 2).ne(0).sum().item()),
                               "map_points_dynamic": int(((d_mapflags & 3) == 1).sum().item())},
                           "map_points_refined": int((loop.d_map[:n_pts0] - torch.from_numpy(sc.points).to(dev)).abs().amax(dim=1).gt(0).sum().item()),
                           "features_dynamic_last_frame": [int(v) for v in ((loop.d_isstatic == 0) & (loop.d_state >= 0)).sum(dim=1).cpu().tolist()],
                           "static_mapped_features_last_frame": [int(v) for v in ((loop.d_state >= 0) & (loop.d_slot2map >= 0)).sum(dim=1).cpu().tolist()]},
                       "key_frame_solves_duty": solve_duty,
                       "host_enqueue_ms_per_step": t_host / args.steps * 1e3, "host_enqueue_ms_max_step": t_step_max * 1e3, "host_enqueue_max_at_step": i_step_max,
                       "ncc_matching": None if loop.ncc is None else {
                           "every_frames": cfg.ncc_every, "camera_pairs_per_run": N_CAMS - 1, "runs": loop.ncc["runs"],
                           "what": "NewMapPtsNCC every 4th frame: getNCCBlocks of the rank's cameras (N > 1: one all-gather of the blocks), "
                                   "the epipolar / NCC test of all consecutive camera pairs as candidate lists, then on the device: seeds + "
                                   "disparity guide + greedy matches, featTracksFromMatches, reconstructTracks, output -- new map points "
                                   "appended to the live map (cs_newpts_from_pairs_dev)",
                           "candidate_pairs_last_run": [int(v) for v in loop.ncc["pair_count"].cpu().tolist()],
                           "candidate_features_last_run": [int(loop.ncc["valid"][g].view(torch.int32).sum().item()) for g in range(N_CAMS)],
                           **dict(zip(("new_map_points_last_run", "tracks_last_run", "tracks_of_two_or_more_views_last_run", "flags_last_run"),
                                      loop.ncc["np_cnt"].cpu().tolist()[:4])),
                           "matches_per_pair_last_run": loop.ncc["np_cnt"].cpu().tolist()[4:4 + N_CAMS - 1],
                           "map_points_in_use": map_in_use_timed_end, "map_points_at_start": n_pts0, "map_capacity": loop.n_map},
                       "with_upload": with_upload, "secondary_reference_ba_request_policy": ref_policy,
                       "secondary_sequential_registration": seq_reg, "secondary_single_pass_registration": single_pass, "secondary_keyframe_decision": kf_leg, "cxx_frame_loop": cxx,
                       "value_source": "cxx_frame_loop" if cxx_ok else "python_frame_loop (the C++ loop did not run: see cxx_frame_loop)",
                       "python_frame_loop": py_loop,
                       "secondary_legs_note": "the secondary_* legs and with_upload are run by the Python loop: their ratio_to_value compares with python_frame_loop",
                       "collectives": None if world == 1 else {
                           "issued_by": ("libcoslam_hip RCCL (C-ABI)" if loop.native else "torch.distributed " + dist_backend +
                                         (f" (FALLBACK: libcoslam_hip's communicator could not be created: {loop.native_fallback})"
                                          if getattr(loop, "native_fallback", None) else "")),
                           "us_per_call_measured_after_the_run": coll_us},
                       "streams": "tracker group | hand-back + pose + map update + registration (event-ordered behind the tracker of the same "
                                  "frame) | inter-camera solve and joint local BA each on its workspace's worker thread + stream "
                                  "(cs_ba_solve_async, like the reference's BA worker thread)"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()



def _newpts_loaded_leg(dev, reps=20):
    """cs_newpts_from_pairs_dev on the golden scene with the most tracks: per-launch time by HIP events on the launch's stream, the
    new points checked against the reference's (count, positions bit for bit) after every launch."""
    import numpy as np
    import torch

    from coslam_amd.ncc import NCC_PAIR_DTYPE
    from coslam_amd.newpts import NewPtsJob, newpts_from_pairs_dev, newpts_scratch_bytes
    from tests.newpts_golden_util import GOLDEN, exact_inverse_of, scene

    g = np.load(GOLDEN)
    sc = max(range(int(g["n_scenes"])), key=lambda i: len(g[f"s{i}_track_len"]))
    S = scene(g, sc)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    nc, N, NS, cap, n_old = S["nc"], S["N"], S["NS"], S["cap"], S["n_old"]
    dK, diK = d(S["K"].reshape(9)), d(exact_inverse_of(S["K"]).reshape(9))
    dxy, dst, ds2m, dstat = d(np.stack(S["xy"])), d(np.stack(S["state"])), d(np.stack(S["s2m"])), d(np.stack(S["is_static"]))
    drep = torch.zeros((nc, NS), dtype=torch.float64, device=dev)
    CAPP = 1024
    dpairs = torch.zeros((nc - 1, CAPP * NCC_PAIR_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    dcnt = torch.zeros(nc - 1, dtype=torch.int32, device=dev)
    for a in range(nc - 1):
        arr = np.zeros(len(S["pairs"][a]), dtype=NCC_PAIR_DTYPE)
        for k, q in enumerate(S["pairs"][a]):
            arr[k] = q
        dpairs[a, :arr.nbytes] = torch.from_numpy(arr.view(np.uint8)).to(dev)
        dcnt[a] = len(arr)
    cams = [dict(K=dK.data_ptr(), iK=diK.data_ptr(), xy=dxy[c].data_ptr(), state=dst[c].data_ptr(), slot2map=ds2m[c].data_ptr(),
                 isStatic=dstat[c].data_ptr(), reprojErr=drep[c].data_ptr()) for c in range(nc)]
    job = NewPtsJob(cams, [dpairs[a].data_ptr() for a in range(nc - 1)], [dcnt[a:a + 1].data_ptr() for a in range(nc - 1)])
    dM, dC = torch.zeros((cap, 3), dtype=torch.float64, device=dev), torch.zeros((cap, 9), dtype=torch.float64, device=dev)
    dF0, dPf0, ds2m0 = d(S["flags"]), d(S["pf"]), ds2m.clone()
    dF, dPf = dF0.clone(), dPf0.clone()
    dNew, dFirst = torch.zeros(cap, dtype=torch.uint8, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
    dCount = torch.tensor([n_old], dtype=torch.int32, device=dev)
    dScr = torch.zeros(newpts_scratch_bytes(nc, NS), dtype=torch.uint8, device=dev)
    dOut = torch.zeros(4 + nc, dtype=torch.int32, device=dev)
    dR, dT = d(S["R"]), d(S["t"])
    st = torch.cuda.current_stream()
    w = S["want"]
    n_new = len(w["M"])
    us, ok = [], True
    for rep in range(reps + 3):
        dF.copy_(dF0), dPf.copy_(dPf0), ds2m.copy_(ds2m0), dCount.fill_(n_old), dNew.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        newpts_from_pairs_dev(st.cuda_stream, job, NS, CAPP, dR.data_ptr(), dT.data_ptr(), dM.data_ptr(), dC.data_ptr(), dF.data_ptr(),
                              dNew.data_ptr(), dFirst.data_ptr(), dPf.data_ptr(), cap, dCount.data_ptr(), S["frame"], dScr.data_ptr(),
                              dOut.data_ptr(), maxDisp=80.0, W=S["W"], H=S["H"])
        e1.record(st)
        torch.cuda.synchronize()
        if rep >= 3:
            us.append(e0.elapsed_time(e1) * 1e3)
        ok = ok and int(dOut[0].item()) == n_new and bool(np.array_equal(dM.cpu().numpy()[n_old:n_old + n_new], w["M"]))
    us.sort()
    return {"workload": "tests/golden/newpts_golden.npz scene %d: %d cameras, %d candidate features per camera, %d matches over the "
                        "consecutive pairs, %d tracks (%d of two or more views), %d new map points" %
                        (sc, nc, N, sum(len(q) for q in S["pairs"]), len(w["track_len"]), int((w["track_len"] >= 2).sum()), n_new),
            "what": "cs_newpts_from_pairs_dev (greedy matches, featTracksFromMatches, reconstructTracks, decidePointType, output), "
                    "timed per call with HIP events on its stream; the new points compared with the reference's after every call",
            "us_per_call_median": us[len(us) // 2], "us_per_call_min": us[0], "us_per_call_max": us[-1], "calls": len(us),
            "new_points_equal_the_reference": ok}

if __name__ == "__main__":
    main()
