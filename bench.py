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
N_FRAMES = 24
PTS_STRIDE = 192          # 12 x 16 blocks (reference src/app/SL_SingleSLAM.h:36-37)
N_COL_BLK, N_ROW_BLK = 16, 12
KEY_EVERY = 5
P_REG = 1536             # map points per registration pass (the size of the inter-camera solve's static set)
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
    # ping-pong so that consecutive frames always differ by one camera step
    fwd = list(range(n))
    return fwd + fwd[-2:0:-1]


def build_scene():
    from coslam_amd.synth import Scene

    return Scene(N_CAMS, W, H, 7000, seed=SEED, sigma=1.0)


def build_ba_problems(sc):
    from coslam_amd.synth import make_intercam_problem, make_joint_ba_problem

    return make_joint_ba_problem(sc, seed=SEED + 7), make_intercam_problem(sc, seed=SEED + 11)


def csr(pr):
    P = len(pr["pts0"])
    obs_pt = np.asarray(pr["obs_pt"])
    order = np.argsort(obs_pt, kind="stable")
    ptr = np.zeros(P + 1, dtype=np.int32)
    np.add.at(ptr, obs_pt + 1, 1)
    return np.cumsum(ptr).astype(np.int32), pr["obs_cam"][order], pr["obs_xy"][order]


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
    A = rng.normal(size=(2 * P_REG if n is None else max(n, 2 * P_REG), 3, 3)) * 0.02
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


def cpu_baseline(sc, frames, joint, ic, n_threads, budget_s, with_register=True, with_posegraph=True):
    """The oracle (C restatement of the reference's path: kind "port") on `n_threads` host cores: the cameras of a frame
    in parallel (the ctypes calls release the GIL), the key-frame solves on the calling thread."""
    import oracle

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
    no_feat = np.full((P_REG, 1), -1, dtype=np.int32)
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
                                    map_flags, 0, PIXEL_ERR_VAR, reproj, cams=[c])
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
                                   fs_all, pf_all, frame_no, map_pts, map_cov, map_flags, cls_new, cls_sfn, cls_first, 12.0)
        for c in range(N_CAMS):
            is_static[c][:] = fs_all[c]

    Kc = sc.K

    def reg_step(c):
        # activeMapPointsRegister + currentMapPointsRegister, search step, this camera's column of the tables
        one = lambda a: [a]  # noqa: E731
        oracle.register_search(W, H, Kc, Rc[c], tc[c], one(xy[c]), one(st[c]), one(s2m[c]), one(None), map_pts[P_REG:2 * P_REG],
                               cov[P_REG:2 * P_REG], no_feat, 2.5 * PIXEL_ERR_VAR, 3 * PIXEL_ERR_VAR, PIXEL_ERR_VAR)
        pf = oracle.point_features(st[c], s2m[c], P_REG).reshape(P_REG, 1)
        rs = oracle.register_search(W, H, Kc, Rc[c], tc[c], one(xy[c]), one(st[c]), one(s2m[c]), one(None), map_pts[:P_REG],
                                    cov[:P_REG], pf, PIXEL_ERR_VAR, 3 * PIXEL_ERR_VAR, PIXEL_ERR_VAR)
        h = hist[c]
        if h["R"]:   # staticCheckMergability of the candidates over their whole tracks (the history as of the previous frame's pose update)
            oracle.register_mergability_cam(Kc, np.stack(h["R"]), np.stack(h["t"]), np.stack(h["xy"]), tl[c], map_pts[:P_REG], cov[:P_REG],
                                            rs["slot"][:, 0], PIXEL_ERR_VAR)

    def run_cams(cams, f, frame_no):
        for c in cams:
            cam_step(c, f, frame_no)
            if with_register:
                reg_step(c)

    t_start = time.perf_counter()
    n = 0
    while True:
        f = order[(n + 1) % len(order)]
        if n_threads <= 1:
            run_cams(range(N_CAMS), f, n + 1)
        else:
            parts = [list(range(N_CAMS))[q::n_threads] for q in range(n_threads)]
            th = [threading.Thread(target=run_cams, args=(p, f, n + 1)) for p in parts if p]
            for x in th:
                x.start()
            for x in th:
                x.join()
        pose_update_all(n + 1)
        if n % KEY_EVERY == 0:
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
                                               PIXEL_ERR_VAR)
            oracle.ba_robust(ic["Ks"], ic["Rs0"], ic["ts0"], ic["pts0"], iptr, icam, ixy, 0, ic["n_static"], 6.0, 3, 40)
        n += 1
        if time.perf_counter() - t_start > budget_s or n >= 200:
            break
    dt = time.perf_counter() - t_start
    return n / dt, n, dt


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="diagnostic: every leg on ONE stream (no overlap)")
    ap.add_argument("--no-classify", action="store_true",
                    help="diagnostic: skip mapPointsClassify behind the pose update (not a valid bench line)")
    ap.add_argument("--no-update-points", action="store_true",
                    help="diagnostic: skip updateNewPosesPoints behind the finished joint BA (not a valid bench line)")
    ap.add_argument("--key-every", type=int, default=KEY_EVERY, help="diagnostic: 0 disables the key-frame solves (not a valid bench line)")
    ap.add_argument("--only-solve", choices=["both", "joint", "intercam"], default="both",
                    help="diagnostic: run only one of the two key-frame solves (not a valid bench line)")
    ap.add_argument("--no-pose", action="store_true", help="diagnostic: skip hand-back + pose (not a valid bench line)")
    ap.add_argument("--klt-cus", type=int, default=int(os.environ.get("BENCH_KLT_CUS", "0")),
                    help="tracker stream confined to the first N compute units (0 = whole chip): leaves CUs the persistent tracker "
                         "never occupies, where the BA's 1024-thread solver workgroup can start while the tracker runs")
    ap.add_argument("--klt-cams-per-launch", type=int, default=int(os.environ.get("BENCH_KLT_CAMS_PER_LAUNCH", "-1")),
                    help="cameras per persistent tracker launch.  0 (default, also -1) = as many as are co-resident: all 8 in ONE launch at "
                         "two waves per SIMD, the tracker's own best configuration (145 us per frame, HBM frac 0.043).  4 = two launches of 4 "
                         "cameras back to back at ONE wave per SIMD (2 x 102 us): every SIMD keeps 352 free VGPRs and every CU 96 KB of LDS for "
                         "the key-frame solves' kernels, whose latency chain -- not the tracker -- bounds the loop: +2.5-3 % frames/s, tracker "
                         "HBM frac 0.030 (profiles/r03_tracker_split.txt).  3: 2374, 2: 1978 frames/s")
    ap.add_argument("--reg-stream", type=int, default=int(os.environ.get("BENCH_REG_STREAM", "0")),
                    help="1: the two registration passes of frame f on their own stream behind pose(f) -- they are consumers of the "
                         "frame's poses and features, nothing of frame f+1's tracking or pose depends on them, so they overlap the next "
                         "frame's tracker (hand-back(f+1) waits for them: it rewrites the records they read); 0: on the pose stream")
    ap.add_argument("--ba-persist", default=os.environ.get("BENCH_BA_PERSIST", ""),
                    help="GJ:GI -- the LM loops of the joint BA / the inter-camera solve as ONE cooperative launch of at most GJ / GI "
                         "workgroups each (cs_ba_set_persistent: a compute unit per workgroup, kept for the whole run); the tracker is "
                         "budgeted for the other 256 - GJ - GI compute units")
    ap.add_argument("--ba-cus", default=os.environ.get("BENCH_BA_CUS", ""), help="FIRST:COUNT -- the joint BA's stream confined to these CU-mask bits")
    ap.add_argument("--ic-cus", default=os.environ.get("BENCH_IC_CUS", ""), help="FIRST:COUNT -- the inter-camera solve's stream confined to these CU-mask bits")
    ap.add_argument("--pose-cus", default=os.environ.get("BENCH_POSE_CUS", ""), help="FIRST:COUNT -- the pose stream (hand-back, pose, registration) confined to these CU-mask bits")
    ap.add_argument("--ba-prebaked", action="store_true",
                    help="the joint local BA re-solves ONE pre-baked synthetic problem at every key frame (rounds 1-2) instead of the problem "
                         "parsed on the device from the last 5 key frames' tracked features and poses (N = 1 default: cs_ba_window_*)")
    ap.add_argument("--no-pose-update", action="store_true",
                    help="skip poseUpdate3D's gate + seqTriangulate loop and the dynamic-point test behind the pose solve")
    ap.add_argument("--ncc-dense", action="store_true",
                    help="the NCC matching leg writes getEpiNccMat's two dense 2000 x 2000 matrices per camera pair (64 MB) instead of the "
                         "list of the pairs that pass (cs_ncc_epi_pairs_dev)")
    ap.add_argument("--ncc-stream", type=int, default=int(os.environ.get("BENCH_NCC_STREAM", "0")),
                    help="1: the NCC matching leg on its own stream against a snapshot of the frame's records (A/B)")
    ap.add_argument("--no-mergability", action="store_true",
                    help="skip staticCheckMergability over the candidates' whole tracks behind the current-static registration pass")
    ap.add_argument("--no-ncc", action="store_true", help="diagnostic: skip the inter-camera NCC matching leg (not a valid bench line)")
    ap.add_argument("--no-cxx-loop", action="store_true", help="skip the C++ frame loop (tools/cxx/frame_loop.bin, config.cxx_frame_loop)")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the upload-inclusive repetition of the loop (config.with_upload)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary cfg5 BA leg (3 s of problem generation)")
    ap.add_argument("--no-posegraph", action="store_true", help="diagnostic: skip the pose-graph relaxation behind the joint BA (not a valid bench line)")
    ap.add_argument("--no-register", action="store_true", help="diagnostic: skip the map-point registration search (not a valid bench line)")
    ap.add_argument("--ba-sliced", choices=["auto", "0", "1"], default="auto",
                    help="N > 1: joint BA sliced by points with an all-reduce per LM step (1), solved redundantly by every rank "
                         "without any collective (0), or chosen by size (auto: sliced from 200 k measurements)")
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

    import torch
    import torch.distributed as dist

    import coslam_amd
    from coslam_amd.ba import BAWorkspace
    from coslam_amd.handback import handback_cams, handback_dev
    from coslam_amd.pose import intraCamEstimate_batch_dev
    from coslam_amd.register import register_cams, register_search_dev

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks for exercising the N > 1 code path on a box with ONE GPU: BENCH_FORCE_DEVICE pins every rank to that
    # device, BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU).  Never set by the driver.
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    dist_backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    n_gpus = max(world, 1)
    if N_CAMS % n_gpus != 0:
        raise SystemExit(f"bench.py: {N_CAMS} cameras do not shard over {n_gpus} GPUs (use 1, 2, 4 or 8)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(dist_backend, rank=rank, world_size=world)

    cams_here = N_CAMS // n_gpus
    my_cams = list(range(rank * cams_here, (rank + 1) * cams_here))
    sc = build_scene()
    order = frame_order(N_FRAMES)
    frames = {c: np.stack([sc.render(c, f) for f in range(N_FRAMES)]) for c in (range(N_CAMS) if (rank == 0 and n_gpus == 1) else my_cams)}
    joint, ic = build_ba_problems(sc)

    # ---- everything resident in HBM before the clock starts -------------------------------------
    d_frames = [torch.from_numpy(frames[c]).to(dev) for c in my_cams]
    nc = cams_here
    d_K = torch.from_numpy(np.tile(sc.K.ravel(), nc)).to(dev)
    d_K1 = torch.from_numpy(sc.K.ravel().copy()).to(dev)
    d_kud = torch.zeros(7, dtype=torch.float64, device=dev)
    d_map = torch.from_numpy(sc.points.copy()).to(dev)
    d_slot2map = torch.full((nc, N_FEAT), -1, dtype=torch.int32, device=dev)
    d_trackspan = torch.full((nc, 2 * N_FEAT), -1, dtype=torch.int32, device=dev)
    d_xy = torch.zeros((nc, 2 * N_FEAT), dtype=torch.float64, device=dev)
    d_state = torch.zeros((nc, N_FEAT), dtype=torch.int32, device=dev)
    d_Ms = torch.zeros((nc, PTS_STRIDE, 3), dtype=torch.float64, device=dev)
    d_ms = torch.zeros((nc, PTS_STRIDE, 2), dtype=torch.float64, device=dev)
    d_sel = torch.zeros((nc, PTS_STRIDE), dtype=torch.int32, device=dev)
    d_npts = torch.zeros(nc, dtype=torch.int32, device=dev)
    d_opt = torch.zeros((nc, 96), dtype=torch.uint8, device=dev)
    d_ok = torch.zeros(nc, dtype=torch.int32, device=dev)
    # map-point registration: P x nCams tables (this rank's cameras are its columns), covariances of the 2 x P_REG points
    n_map = len(sc.points)
    d_cov = torch.from_numpy(reg_covariances(n_map).reshape(-1).copy()).to(dev)  # MapPoint::cov of every map point
    d_pf = torch.full((n_map, nc), -1, dtype=torch.int32, device=dev)           # MapPoint::pFeatures of this frame: written by the hand-back
    d_pf_none = torch.full((P_REG, nc), -1, dtype=torch.int32, device=dev)      # active points: no feature of this frame
    reg_out = [dict(slot=torch.zeros((P_REG, nc), dtype=torch.int32, device=dev), m=torch.zeros((P_REG, nc, 2), dtype=torch.float64, device=dev),
                    var=torch.zeros((P_REG, nc, 4), dtype=torch.float64, device=dev), dist=torch.zeros((P_REG, nc), dtype=torch.float64, device=dev),
                    flags=torch.zeros((P_REG, nc), dtype=torch.int32, device=dev)) for _ in range(2)]
    R0 = np.stack([sc.pose(c, order[0])[0].ravel() for c in my_cams])
    t0 = np.stack([sc.pose(c, order[0])[1] for c in my_cams])
    d_R = [torch.from_numpy(R0.copy()).to(dev), torch.from_numpy(R0.copy()).to(dev)]   # pose ping-pong: frame f reads [f&1^1]
    d_t = [torch.from_numpy(t0.copy()).to(dev), torch.from_numpy(t0.copy()).to(dev)]
    d_dests = [[torch.zeros(N_FEAT * 5, dtype=torch.int32, device=dev) for _ in range(nc)] for _ in range(2)]
    d_counts = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(nc)]

    # Streams and threads mirror the reference's threads and data dependences:
    #   klt     the tracker of all cameras (one camera group): frame f+1 only needs the tracker state of frame f;
    #   pose    hand-back + intraCamEstimate of frame f (event-ordered behind the tracker of frame f), then the all-gather
    #           of features||pose at the merge step (N > 1);
    #   key-frame solves (inter-camera pose, joint local BA): each on its workspace's own worker thread + stream
    #           (cs_ba_solve_async), started behind that frame's pose -- the reference's BA worker thread
    #           (src/app/SL_CoSLAM.cpp:1702-1784); the thread enqueues LM steps in chunks and stops at convergence.
    if args.klt_cus > 0:
        coslam_amd.lib().cs_stream_create_cu_range.restype = C.c_void_p
        ptr = coslam_amd.lib().cs_stream_create_cu_range(local_rank, 0, args.klt_cus)
        if not ptr:
            raise SystemExit("bench.py: cs_stream_create_cu_range failed")
        klt_s = torch.cuda.ExternalStream(ptr, device=dev)
    else:
        klt_s = torch.cuda.Stream(device=dev)
    def masked_stream(spec):
        first, count = (int(v) for v in spec.split(":"))
        coslam_amd.lib().cs_stream_create_cu_range.restype = C.c_void_p
        p = coslam_amd.lib().cs_stream_create_cu_range(local_rank, first, count)
        if not p:
            raise SystemExit("bench.py: cs_stream_create_cu_range failed: " + coslam_amd.lib().cs_last_error().decode())
        return p

    pose_s = klt_s if args.serial else (torch.cuda.ExternalStream(masked_stream(args.pose_cus), device=dev) if args.pose_cus
                                        else torch.cuda.Stream(device=dev))
    ba_s = klt_s if args.serial else torch.cuda.Stream(device=dev)   # N > 1: the sliced joint BA and its collectives
    reg_s = torch.cuda.Stream(device=dev) if (args.reg_stream and not args.serial) else pose_s

    trks = []
    for _ in my_cams:
        t = coslam_amd.KLT_SequenceTracker(klt_config(), device=local_rank)
        t.allocate(W, H, LEVELS, FW, FH)
        trks.append(t)
    grp = coslam_amd.KLT_TrackerGroup(trks)
    grp.set_stream(klt_s.cuda_stream)
    if args.klt_cus > 0:
        for t in trks:
            t.set_cu_count(args.klt_cus)
    persist_j = persist_i = 0
    if args.ba_persist:
        persist_j, persist_i = (int(v) for v in args.ba_persist.split(":"))
        if not args.klt_cus and args.klt_cams_per_launch <= 0:
            for t in trks:
                t.set_cu_count(256 - persist_j - persist_i)
    if args.klt_cams_per_launch < 0:
        args.klt_cams_per_launch = 0
    if args.klt_cams_per_launch > 0 and not args.klt_cus:
        # the co-residency budget of the persistent tracker is what decides how many cameras share a launch: hand it the
        # budget of cams_per_launch cameras (250 waves each, 8 resident waves per CU) -- no CU mask, the launches still spread
        # over the whole chip
        for t in trks:
            t.set_cu_count(min(256, (250 * args.klt_cams_per_launch + 60) // 8 + 5))
    prefetch = os.environ.get("BENCH_PREFETCH", "1") != "0"

    # Joint local BA, data-coupled (N = 1): a ring of the last 5 key frames on the device -- every camera's hand-back records
    # and solved pose at the key frame -- from which RobustBundleRTS::addKeyFrames / addPoints / parseInputs' flat problem is
    # built on the device at every key frame (cameras = 5 key frames x 8, the 16 oldest held; points = mapped map points with
    # more than one feature point in the window) and solved from the tracked poses.  N > 1 keeps the pre-baked problem: the
    # all-gather carries the trackers' dest[] records, not the hand-back's.
    use_window = (world == 1) and not args.ba_prebaked and not args.serial and not args.no_pose
    jptr, jcam, jxy = csr(joint)
    ba_ws = BAWorkspace(local_rank)
    if args.ba_cus:
        ba_ws.set_stream(masked_stream(args.ba_cus))
    ba_win = None
    if use_window:
        from coslam_amd.ba import BAWindow

        ba_win = BAWindow(nc, 5, N_FEAT, len(sc.points), device=local_rank)
        ba_win.reserve(ba_ws)
    else:
        ba_ws.upload(joint["Ks"], joint["Rs0"], joint["ts0"], joint["pts0"], jptr, jcam, jxy)
    if persist_j:
        ba_ws.set_persistent(persist_j)
    d_jR = torch.from_numpy(joint["Rs0"].reshape(-1).copy()).to(dev)
    d_jT = torch.from_numpy(joint["ts0"].reshape(-1).copy()).to(dev)
    d_jM = torch.from_numpy(joint["pts0"].reshape(-1).copy()).to(dev)
    # RobustBundleRTS::output() behind every joint BA: adjusted key poses into the fixed nodes of this rank's camera graphs,
    # relaxation of the non-key frames -- installed as the workspace's follow-up (N = 1: the worker thread enqueues it behind
    # the solve's last kernel) or enqueued behind the sliced solve (N > 1)
    # N > 1: the joint BA of the headline (20 k measurements, LM step ~100 us of dependent launches) is latency-bound -- slicing
    # it by points shortens no link of that chain and adds two all-reduces per LM step, so every rank solves it redundantly from
    # the all-gathered measurements (bit-identical results, no collective).  Throughput-bound problems (cfg5: 600 k measurements)
    # are sliced.  --ba-sliced 1 forces the sliced solve (SURVEY 8e collective 2).
    ba_sliced = world > 1 and (args.ba_sliced == "1" or (args.ba_sliced == "auto" and len(joint["obs_cam"]) >= 200000))
    pg = None
    if not args.no_posegraph:
        from coslam_amd.posegraph import PoseGraphs, after_ba_function, after_ba_record, posegraph_set_poses_dev

        pg_graphs, pg_R, pg_T, pg_cam = build_pose_graphs(sc, joint, my_cams)
        pg = PoseGraphs(pg_graphs, device=local_rank)
        d_pgR, d_pgT = torch.from_numpy(pg_R).to(dev), torch.from_numpy(pg_T).to(dev)
        d_pgER = torch.zeros(pg.n_edges, 9, dtype=torch.float64, device=dev)
        d_pgET = torch.zeros(pg.n_edges, 3, dtype=torch.float64, device=dev)
        d_pgNR, d_pgNT = torch.zeros_like(d_pgR), torch.zeros_like(d_pgT)
        d_pgCam = torch.from_numpy(pg_cam).to(dev)
        s0 = torch.cuda.current_stream().cuda_stream
        pg.edges_dev(s0, d_pgR.data_ptr(), d_pgT.data_ptr(), d_pgER.data_ptr(), d_pgET.data_ptr())   # constructCameraGraphs
        torch.cuda.synchronize()
        bR, bT, _ = ba_ws.result_buffers()
        pg_rec = after_ba_record(pg, len(pg_cam), d_pgCam.data_ptr(), bR, bT, d_pgR.data_ptr(), d_pgT.data_ptr(), d_pgER.data_ptr(),
                                 d_pgET.data_ptr(), d_pgNR.data_ptr(), d_pgNT.data_ptr(), device=local_rank)
        if world == 1 or not ba_sliced:
            ba_ws.set_followup(after_ba_function(), C.addressof(pg_rec))
    iptr, icam, ixy = csr(ic)
    ic_ws = BAWorkspace(local_rank)
    if args.ic_cus:
        ic_ws.set_stream(masked_stream(args.ic_cus))
    ic_ws.upload(ic["Ks"], ic["Rs0"], ic["ts0"], ic["pts0"], iptr, icam, ixy)
    if persist_i:
        ic_ws.set_persistent(persist_i)
    d_iR = torch.from_numpy(ic["Rs0"].reshape(-1).copy()).to(dev)
    d_iT = torch.from_numpy(ic["ts0"].reshape(-1).copy()).to(dev)
    d_iM = torch.from_numpy(ic["pts0"].reshape(-1).copy()).to(dev)

    # merge step at N > 1: one all-gather of every camera's {features, R, t}; the joint BA sliced by points
    xchg = None
    native = None
    if world > 1:
        from coslam_amd import multicam

        if args.native_comm and dist_backend == "nccl":
            try:
                native = multicam.NativeComm(world, rank, local_rank)
            except Exception as ex:  # noqa: BLE001
                # no silent change of what is measured: the library's RCCL path is the product; torch.distributed collectives
                # are available, but only when asked for
                raise SystemExit(f"bench.py: libcoslam_hip's RCCL communicator could not be created ({ex}); "
                                 "run with --native-comm 0 to measure with torch.distributed collectives instead")
        xchg = multicam.CameraExchange(N_FEAT * nc, dev, native=native, cams_per_rank=nc)

    ic_start_snap = torch.zeros(12 * N_CAMS, dtype=torch.float64, device=dev)
    klt_done = [torch.cuda.Event(), torch.cuda.Event()]
    dest_free = [torch.cuda.Event(), torch.cuda.Event()]
    pose_done = torch.cuda.Event()
    pose_ready, reg_done = torch.cuda.Event(), torch.cuda.Event()

    # poseUpdate3D's second half + detectDynamicFeaturePoints (reference src/app/SL_SingleSLAM.cpp:672-708, 784-824), every frame
    # behind the pose solve, one launch for all cameras: the Mahalanobis gate and seqTriangulate refine the map points and their
    # covariances IN PLACE (what the next frame's hand-back, the registration and the BA window then read), the dynamic test
    # walks a ring of the last 64 frames' pixels and poses and sets the feature types the next hand-back's block vote takes.
    # N > 1: this rank's cameras only (the map is a per-rank replica there).
    pose_upd = None
    if not args.no_pose and not args.no_pose_update:
        from coslam_amd.poseupdate import TrackHistory, poseupdate_cams

        PU_HIST = 64
        d_isstatic = torch.ones((nc, N_FEAT), dtype=torch.uint8, device=dev)
        d_reproj = torch.zeros((nc, N_FEAT), dtype=torch.float64, device=dev)
        d_mapflags = torch.zeros(n_map, dtype=torch.uint8, device=dev)
        d_mergeable = torch.zeros((P_REG, nc), dtype=torch.uint8, device=dev)
        # CoSLAM::mapPointsClassify (reference src/app/SL_CoSLAM.cpp:381-385, 418-520): the points the gate made uncertain and the
        # dynamic ones decided again every frame, one launch behind the pose update (MapPoint::bNewPt / staticFrameNum / firstFrame)
        d_newpt = torch.zeros(n_map, dtype=torch.uint8, device=dev)
        d_sfn = torch.zeros(n_map, dtype=torch.int32, device=dev)
        d_firstfrm = torch.zeros(n_map, dtype=torch.int32, device=dev)
        d_cls_counts = torch.zeros(2, dtype=torch.int32, device=dev)
        d_iK1 = torch.from_numpy(np.linalg.inv(sc.K).ravel().copy()).to(dev)
        pose_upd = TrackHistory(nc, N_FEAT, PU_HIST, device=local_rank)
        pu_args = poseupdate_cams([dict(K=d_K1.data_ptr(), iK=d_iK1.data_ptr(), xy=d_xy[i].data_ptr(), state=d_state[i].data_ptr(),
                                        slot2map=d_slot2map[i].data_ptr(), trackSpan=d_trackspan[i].data_ptr(),
                                        reprojErr=d_reproj[i].data_ptr(), isStatic=d_isstatic[i].data_ptr()) for i in range(nc)])

    # RobustBundleRTS::output()'s updateNewPosesPoints (reference src/app/SL_CoSLAMRobustBA.cpp:248-271, 311-315): once the worker's
    # joint BA has finished -- the main loop reads cs_ba_completed between frames (the host runs ahead of the device, so the NEXT
    # solve is always queued already; the reference's BA thread calls output() itself under the lock it shares with tracking) --
    # every map point is triangulated again from its features of this frame and the widest-parallax view of each track (history
    # ring), one launch behind this frame's pose update.  Here it works on a COPY of the map: the bench's video repeats 24 frames
    # and its pose graph is the pre-baked chain, so feeding re-triangulated points (and scattering relaxed poses into the ring,
    # cs_track_history_set_poses_dev) back into the loop would change what the following frames compute from run to run.
    upd_pts = None
    if pose_upd is not None and ba_win is not None and not args.no_update_points:
        upd_pts = dict(requested=0, applied=ba_ws.completed(), runs=0, first_key=0, d_map=torch.zeros_like(d_map), d_cov=torch.zeros_like(d_cov),
                       d_counts=torch.zeros(2, dtype=torch.int32, device=dev))

    def update_points_leg(i):
        done = ba_ws.completed()
        if done == upd_pts["applied"]:
            return
        upd_pts["applied"] = done
        upd_pts["runs"] += 1
        with torch.cuda.stream(pose_s):
            upd_pts["d_map"].copy_(d_map, non_blocking=True)
            upd_pts["d_cov"].copy_(d_cov, non_blocking=True)
        pose_upd.update_new_poses_points_dev(pose_s.cuda_stream, pu_args, d_pf.data_ptr(), n_map, upd_pts["d_map"].data_ptr(),
                                             upd_pts["d_cov"].data_ptr(), d_mapflags.data_ptr(), PIXEL_ERR_VAR,
                                             firstKeyFrame=upd_pts["first_key"], d_counts=upd_pts["d_counts"].data_ptr())

    def hb_cams(b):
        return [dict(dest=d_dests[b][i].data_ptr(), K=d_K1.data_ptr(), kud=d_kud.data_ptr(), mapPts=d_map.data_ptr(),
                     slot2map=d_slot2map[i].data_ptr(), trackSpan=d_trackspan[i].data_ptr(), xy=d_xy[i].data_ptr(),
                     state=d_state[i].data_ptr(), Ms=d_Ms[i].data_ptr(), ms=d_ms[i].data_ptr(), sel=d_sel[i].data_ptr(),
                     npts=d_npts[i:i + 1].data_ptr(), opt=d_opt[i].data_ptr(), pointFeat=d_pf.data_ptr() + 4 * i,
                     pointFeatStride=nc, nPointFeat=n_map, isStatic=(d_isstatic[i].data_ptr() if pose_upd is not None else 0))
                for i in range(nc)]

    hb_args = [handback_cams(hb_cams(0)), handback_cams(hb_cams(1))]   # ctypes arrays, built once
    dest_ptrs = [[d.data_ptr() for d in d_dests[b]] for b in range(2)]
    cnt_ptrs = [c.data_ptr() for c in d_counts]
    img_ptrs = [[d_frames[i][f].data_ptr() for i in range(nc)] for f in range(N_FRAMES)]

    def pose_leg(b, i):
        if reg_s is not pose_s and i >= 2 and not args.no_register:
            pose_s.wait_event(reg_done)     # the registration of the previous frame reads the records this hand-back rewrites
        handback_dev(pose_s.cuda_stream, hb_args[b], N_FEAT, W, H, N_COL_BLK, N_ROW_BLK, PTS_STRIDE, device=local_rank,
                     frame=i)
        src, dst = (i + 1) & 1, i & 1
        intraCamEstimate_batch_dev(pose_s.cuda_stream, nc, PTS_STRIDE, d_K.data_ptr(), d_R[src].data_ptr(),
                                   d_t[src].data_ptr(), d_npts.data_ptr(), 0, d_Ms.data_ptr(), d_ms.data_ptr(), 10.0,
                                   d_R[dst].data_ptr(), d_t[dst].data_ptr(), d_opt.data_ptr(), d_ok.data_ptr(),
                                   device=local_rank)
        if pose_upd is not None:
            # parallelPoseUpdate(false): gate 2.0, sigma = PIXEL_ERR_VAR; detectDynamicFeaturePoints(20, 5, 3, MAX_EPI_ERR)
            pose_upd.pose_update_frame_dev(pose_s.cuda_stream, pu_args, d_pf.data_ptr(), n_map, d_R[dst].data_ptr(),
                                           d_t[dst].data_ptr(), d_map.data_ptr(), d_cov.data_ptr(), d_mapflags.data_ptr(), 0,
                                           PIXEL_ERR_VAR, i, 20, 5, 3, MAX_EPI_ERR)
            if not args.no_classify:
                pose_upd.map_points_classify_dev(pose_s.cuda_stream, pu_args, d_pf.data_ptr(), n_map, i, d_map.data_ptr(), d_cov.data_ptr(),
                                                 d_mapflags.data_ptr(), d_newpt.data_ptr(), d_sfn.data_ptr(), d_firstfrm.data_ptr(), 12.0,
                                                 d_counts=d_cls_counts.data_ptr())
        if not args.no_register:
            if reg_s is not pose_s:
                pose_ready.record(pose_s)
                reg_s.wait_event(pose_ready)
            register_leg(dst)
            if reg_s is not pose_s:
                reg_done.record(reg_s)

    def reg_cams(dst):
        return [dict(K=d_K1.data_ptr(), R=d_R[dst].data_ptr() + 72 * i, t=d_t[dst].data_ptr() + 24 * i, xy=d_xy[i].data_ptr(),
                     state=d_state[i].data_ptr(), slot2map=d_slot2map[i].data_ptr()) for i in range(nc)]

    reg_args = [register_cams(reg_cams(0)), register_cams(reg_cams(1))]

    from coslam_amd.register import register_passes, register_search_passes_dev

    # CoSLAMThread.cpp:108 activeMapPointsRegister, then :117 currentMapPointsRegister (static points), search step: the two
    # passes of a frame in ONE launch (cs_register_search_passes_dev)
    reg_passes = register_passes([dict(P=P_REG, sigmaSearch=sS, maxDist=3 * PIXEL_ERR_VAR, sigmaMerge=PIXEL_ERR_VAR,
                                       M=d_map.data_ptr() + 24 * pts_off, cov=d_cov.data_ptr() + 72 * pts_off, pointFeat=pf.data_ptr(),
                                       slot=reg_out[k]["slot"].data_ptr(), m=reg_out[k]["m"].data_ptr(), var=reg_out[k]["var"].data_ptr(),
                                       dist=reg_out[k]["dist"].data_ptr(), flags=reg_out[k]["flags"].data_ptr())
                                  for k, (pts_off, pf, sS) in enumerate(((P_REG, d_pf_none, 2.5 * PIXEL_ERR_VAR), (0, d_pf, PIXEL_ERR_VAR)))])

    def register_leg(dst):
        register_search_passes_dev(reg_s.cuda_stream, reg_args[dst], N_FEAT, W, H, reg_passes, device=local_rank)
        if pose_upd is not None and not args.no_mergability:
            # staticCheckMergability of every candidate of the current-static pass over its whole track (SL_CoSLAM.cpp:714-729, :768)
            pose_upd.register_mergability_dev(reg_s.cuda_stream, pu_args, P_REG, d_map.data_ptr(), d_cov.data_ptr(),
                                              reg_out[1]["slot"].data_ptr(), PIXEL_ERR_VAR, d_mergeable.data_ptr())

    # upload-inclusive variant (config.with_upload): the frames arrive in PINNED HOST memory (the capture threads' buffers,
    # reference src/app/SL_CoSLAM.cpp:119-133) and every frame's 8 x 300 KB go host -> device inside the loop: frame i + 2 is
    # staged (cs_klt_group_stage_h: copy stream, ring of 3 slots) while frame i is tracked and frame i + 1 is prefetched
    h_frames = None
    stage_slot = {}

    upload_mode = os.environ.get("BENCH_UPLOAD_MODE", "")   # diagnostics: "copyonly" = stage but track the resident images

    def stage(i):
        f = order[i % len(order)]
        stage_slot[i] = grp.stage_h([h_frames[f][c].data_ptr() for c in range(nc)])

    # Inter-camera NCC matching for new map points, every NCC_EVERY-th frame (CoSLAM::genNewMapPoints: "curFrame -
    # m_lastFrmInterMapping > 3", reference src/app/SL_CoSLAM.cpp:1368-1371 -> NewMapPtsNCC::run: matchBetween(i, i + 1) for the
    # consecutive cameras of the group, src/app/SL_NewMapPointsInterCam.cpp:150-158,273-290): per camera getNCCBlocks on the FULL
    # frame (cv::resize by 0.3 + cv::getRectSubPix per feature), per pair the epipolar-error and NCC matrices over all slots
    # (slots that are not unmapped features of this frame are masked out); F from the frame's poses; the greedy matcher that
    # consumes the matrices stays with the caller.
    NCC_EVERY = 4
    NCC_PAIR_CAP = 1 << 16   # passing pairs kept per camera pair and run (the dense matrices hold 4 M entries, a few dozen pass)
    ncc = None
    if not args.no_ncc and nc >= 2:
        from coslam_amd._lib import check
        from coslam_amd.ncc import (NCC_PAIR_DTYPE, ncc_cams, ncc_epi_mat_dev, ncc_epi_pairs_group_dev, ncc_get_blocks_dev,
                                    ncc_get_blocks_group_dev, ncc_pair_jobs, ncc_scaled_dims)

        ws_, hs_ = ncc_scaled_dims(W, H, 0.3)
        Kinv = np.linalg.inv(sc.K)

        def f_matrix(c1, c2, f):
            (R1, t1), (R2, t2) = sc.pose(c1, f), sc.pose(c2, f)
            R = R1 @ R2.T
            t = t1 - R @ t2
            E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
            return Kinv.T @ E @ Kinv

        ncc = dict(small=torch.zeros((nc, ws_ * hs_), dtype=torch.uint8, device=dev), blk=torch.zeros((nc, N_FEAT, 128), dtype=torch.uint8, device=dev),
                   abc=torch.zeros((nc, N_FEAT, 4), dtype=torch.float64, device=dev), valid=torch.zeros((nc, N_FEAT), dtype=torch.int32, device=dev),
                   epi=torch.zeros((N_FEAT, N_FEAT), dtype=torch.float64, device=dev), score=torch.zeros((N_FEAT, N_FEAT), dtype=torch.float64, device=dev),
                   F={(my_cams[i], f): f_matrix(my_cams[i], my_cams[i + 1], f) for i in range(nc - 1) for f in range(N_FRAMES)}, runs=0,
                   pairs=torch.zeros((nc - 1, NCC_PAIR_CAP * NCC_PAIR_DTYPE.itemsize), dtype=torch.uint8, device=dev),
                   pair_count=torch.zeros(nc - 1, dtype=torch.int32, device=dev))

    # --ncc-stream 1: the matching leg on its own stream, working off a snapshot of the frame's records (the mask of unmapped
    # features and the pixels: two small launches / copies on the pose stream), so that the next frames' hand-backs do not wait
    # for its 270 us
    ncc_s = torch.cuda.Stream(device=dev) if (ncc is not None and args.ncc_stream and not args.serial) else None
    ncc_xy = torch.zeros_like(d_xy) if ncc_s is not None else d_xy
    ncc_snap, ncc_free = torch.cuda.Event(), torch.cuda.Event()
    ncc_group = {}   # per frame of the sequence: the ctypes tables of the run's three launches (built once)

    def ncc_leg(f):
        s_ = pose_s.cuda_stream
        # unmapped features of this frame: state 0 / 1 and no map point (hand-back records of all cameras, back to back)
        if ncc_s is not None and ncc["runs"] > 0:
            pose_s.wait_event(ncc_free)      # (the previous run has read its snapshot)
        check(coslam_amd.lib().cs_ncc_unmapped_mask_dev(local_rank, C.c_void_p(s_), nc * N_FEAT, C.c_void_p(d_state.data_ptr()),
                                                        C.c_void_p(d_slot2map.data_ptr()), C.c_void_p(ncc["valid"].data_ptr())),
              "cs_ncc_unmapped_mask_dev")
        if ncc_s is not None:
            with torch.cuda.stream(pose_s):
                ncc_xy.copy_(d_xy, non_blocking=True)
            ncc_snap.record(pose_s)
            ncc_s.wait_event(ncc_snap)
            s_ = ncc_s.cuda_stream
        if not args.ncc_dense:
            # the whole run in three launches: getNCCBlocks of all cameras (resize, cutter), then every camera pair's passing pairs as
            # a list (what getEpiNccMat's dense matrices hold besides -1: kilobytes instead of 64 MB per camera pair)
            key = (f, id(ncc_xy))
            if key not in ncc_group:
                cams_ = ncc_cams([dict(img=img_ptrs[f][i], x=ncc_xy[i].data_ptr(), y=ncc_xy[i].data_ptr() + 8 * N_FEAT,
                                       scaled=ncc["small"][i].data_ptr(), blocks=ncc["blk"][i].data_ptr(), abc=ncc["abc"][i].data_ptr(),
                                       valid=ncc["valid"][i].data_ptr()) for i in range(nc)])
                jobs_ = ncc_pair_jobs([dict(F=ncc["F"][(my_cams[i], f)], camA=i, camB=i + 1, pairs=ncc["pairs"][i].data_ptr(),
                                            count=ncc["pair_count"][i:i + 1].data_ptr()) for i in range(nc - 1)])
                ncc_group[key] = (cams_, jobs_)
            cams_, jobs_ = ncc_group[key]
            ncc_get_blocks_group_dev(s_, cams_, W, H, N_FEAT, 0.3, device=local_rank)
            ncc_epi_pairs_group_dev(s_, cams_, N_FEAT, jobs_, 50.0, 0.80, NCC_PAIR_CAP, device=local_rank)
            if ncc_s is not None:
                ncc_free.record(ncc_s)
            ncc["runs"] += 1
            return
        for i in range(nc):
            ncc_get_blocks_dev(s_, img_ptrs[f][i], W, H, N_FEAT, ncc_xy[i].data_ptr(), ncc_xy[i].data_ptr() + 8 * N_FEAT, 0.3,
                               ncc["small"][i].data_ptr(), ncc["blk"][i].data_ptr(), ncc["abc"][i].data_ptr(), 0, device=local_rank)
        for i in range(nc - 1):
            ncc_epi_mat_dev(s_, ncc["F"][(my_cams[i], f)], N_FEAT, ncc_xy[i].data_ptr(), ncc_xy[i].data_ptr() + 8 * N_FEAT, ncc["blk"][i].data_ptr(),
                            ncc["abc"][i].data_ptr(), ncc["valid"][i].data_ptr(), N_FEAT, ncc_xy[i + 1].data_ptr(),
                            ncc_xy[i + 1].data_ptr() + 8 * N_FEAT, ncc["blk"][i + 1].data_ptr(), ncc["abc"][i + 1].data_ptr(),
                            ncc["valid"][i + 1].data_ptr(), 50.0, 0.80, -1.0, ncc["epi"].data_ptr(), ncc["score"].data_ptr(),
                            device=local_rank)   # maxEpiErr 50, minNcc 0.80: src/app/SL_NewMapPointsInterCam.h:71-72
        if ncc_s is not None:
            ncc_free.record(ncc_s)
        ncc["runs"] += 1

    def step(i, key_frame, upload=False):
        f, fn = order[i % len(order)], order[(i + 1) % len(order)]
        b = i & 1
        if i >= 2:
            klt_s.wait_event(dest_free[b])      # the consumer of this dest buffer two frames ago is done
        if upload and upload_mode == "copyonly":
            stage(i + 2)
            stage_slot.pop(i)
            cur, nxt = img_ptrs[f], img_ptrs[fn]
        elif upload:
            stage(i + 2)
            cur, nxt = grp.staged(stage_slot.pop(i)), grp.staged(stage_slot[i + 1])
        else:
            cur, nxt = img_ptrs[f], img_ptrs[fn]
        if prefetch:   # this frame's detector tail also builds the next frame's pyramids + cornerness maps
            grp.prefetch_dev(nxt)
        grp.redetect_dev(cur, dest_ptrs[b], cnt_ptrs)
        grp.advanceFrame()
        klt_done[b].record(klt_s)
        pose_s.wait_event(klt_done[b])          # pose(f) consumes what the tracker produced for frame f
        if not args.no_pose:
            pose_leg(b, i)
            if upd_pts is not None:
                update_points_leg(i)
        if world > 1:
            with torch.cuda.stream(pose_s):
                xchg.pack_group(d_dests[b], d_R[i & 1], d_t[i & 1], pose_s)
                xchg.all_gather(pose_s)
        dest_free[b].record(pose_s)
        if key_frame and world > 1:
            # InterCamPoseEstimator::addMapPoints (reference src/app/SL_InterCamPoseEstimator.cpp:24-37) starts the solve from
            # every camera's CURRENT pose: at N > 1 those are the poses this frame's all-gather just delivered (records of all 8
            # cameras: local ones included), not anything rank-local
            with torch.cuda.stream(pose_s):
                for g in range(N_CAMS):
                    _, Rg, tg = xchg.unpack(g, device=local_rank)
                    d_iR[9 * g: 9 * g + 9].copy_(Rg, non_blocking=True)
                    d_iT[3 * g: 3 * g + 3].copy_(tg, non_blocking=True)
                    ic_start_snap[12 * g: 12 * g + 9].copy_(Rg, non_blocking=True)
                    ic_start_snap[12 * g + 9: 12 * g + 12].copy_(tg, non_blocking=True)
        if key_frame:
            if args.only_solve == "joint":
                pass
            elif args.serial:
                ic_ws.solve_dev(klt_s.cuda_stream, d_iR.data_ptr(), d_iT.data_ptr(), d_iM.data_ptr(), 0, ic["n_static"], 6.0, 3, 40)
            else:
                ic_ws.solve_async(pose_s.cuda_stream, d_iR.data_ptr(), d_iT.data_ptr(), d_iM.data_ptr(), 0, ic["n_static"], 6.0, 3, 40)
            if args.only_solve == "intercam":
                pass
            elif world > 1 and ba_sliced:
                pose_done.record(pose_s)
                ba_s.wait_event(pose_done)
                multicam.bundle_adjust_sliced(ba_ws, ba_s, d_jR.data_ptr(), d_jT.data_ptr(), d_jM.data_ptr(),
                                              joint["n_cams_con"], joint["n_pts_con"], 6.0, 2, 10, local_rank, native=native)
                if pg is not None:
                    posegraph_set_poses_dev(ba_s.cuda_stream, len(pg_cam), d_pgCam.data_ptr(), bR, bT, d_pgR.data_ptr(),
                                            d_pgT.data_ptr(), device=local_rank)
                    pg.relax_dev(ba_s.cuda_stream, d_pgR.data_ptr(), d_pgT.data_ptr(), d_pgER.data_ptr(), d_pgET.data_ptr(),
                                 d_pgNR.data_ptr(), d_pgNT.data_ptr())
            elif args.serial:
                ba_ws.solve_dev(klt_s.cuda_stream, d_jR.data_ptr(), d_jT.data_ptr(), d_jM.data_ptr(), joint["n_cams_con"],
                                joint["n_pts_con"], 6.0, 2, 10)
            elif ba_win is not None:
                # this key frame into the ring (the hand-back's records and the poses pose(f) just wrote), then
                # requestForBA(5, 2, 2, 30): the numCams * 2 oldest key cameras held, 2 points held, maxIter 2, inner 10
                ba_win.push_dev(pose_s.cuda_stream, hb_args[b], d_K1.data_ptr(), 1, d_R[i & 1].data_ptr(), d_t[i & 1].data_ptr(), i)
                ba_win.solve_async(ba_ws, pose_s.cuda_stream, d_map.data_ptr(), 2 * nc, 2, 6.0, 2, 10)
                if upd_pts is not None:
                    upd_pts["requested"] += 1
                    upd_pts["first_key"] = i - 4 * args.key_every
            else:
                ba_ws.solve_async(pose_s.cuda_stream, d_jR.data_ptr(), d_jT.data_ptr(), d_jM.data_ptr(), joint["n_cams_con"],
                                  joint["n_pts_con"], 6.0, 2, 10)

    _step_core = step

    def step(i, key_frame, upload=False):   # noqa: F811
        # the NCC matching leg goes LAST on the pose stream: behind the event that frees the tracker's dest buffer and behind
        # the events the key-frame solves wait for (it consumes the frame's records; nothing of this frame waits for it)
        _step_core(i, key_frame, upload)
        if ncc is not None and i % NCC_EVERY == 0:
            ncc_leg(order[i % len(order)])

    def barrier():
        ic_ws.wait()      # the worker threads' queues are part of the timed work
        ba_ws.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # first frame: detect (GPUKLT::first, reference src/tracking/GPUKLT.cpp:133-142), map association, first hand-back
    grp.detect_dev(img_ptrs[order[0]], dest_ptrs[0], cnt_ptrs)
    grp.advanceFrame()
    grp.synchronize()
    for i, c in enumerate(my_cams):
        d = d_dests[0][i].cpu().numpy().view(coslam_amd.KLT_TrackedFeature)
        s2m = associate(sc, c, order[0], d)
        d_slot2map[i].copy_(torch.from_numpy(s2m))
    with torch.cuda.stream(pose_s):
        handback_dev(pose_s.cuda_stream, hb_args[0], N_FEAT, W, H, N_COL_BLK, N_ROW_BLK, PTS_STRIDE, device=local_rank,
                     frame=0)
    torch.cuda.synchronize()
    for i, c in enumerate(my_cams):   # the first hand-back starts every track as new (unmapped): put the map back
        d = d_dests[0][i].cpu().numpy().view(coslam_amd.KLT_TrackedFeature)
        d_slot2map[i].copy_(torch.from_numpy(associate(sc, c, order[0], d)))
    torch.cuda.synchronize()

    # set-up, not warm-up: one key-frame interval so that everything that happens once per process is behind us -- the BA
    # workers capture and instantiate their graphs on first use, the runtime loads each kernel's code object at its first
    # launch -- whatever W the caller asks for
    import gc

    gc.collect()
    gc.disable()   # no collector pauses on the launching thread from here to the end of the timed region
    n_setup = max(args.key_every, 1) + 1
    if ba_win is not None and args.key_every > 0:
        n_setup = 5 * args.key_every + 1      # (fills the ring: every timed solve then has its 5 key frames = 40 cameras)
    for i in range(n_setup):
        step(i + 1, args.key_every > 0 and i % max(args.key_every, 1) == 0)
    barrier()
    # ... and the device out of its idle power state: a fresh process on a fresh box measured 2177 frames/s on the driver's
    # 20-step command where the second and third process measured 2460 / 2489 (profiles/r03_bench_lines.txt) -- the 12 ms of
    # set-up above plus W = 5 frames are over before the clocks have ramped.  Keep the loop running (same cadence, untimed)
    # for SETUP_SECONDS of wall time; K and W are untouched.
    t_su = time.perf_counter()
    ke = max(args.key_every, 1)
    rounds = 0
    # (N > 1: every rank must take the same number of steps -- each carries an all-gather -- so the count is fixed there, not timed)
    while (rounds < 12) if world > 1 else (time.perf_counter() - t_su < SETUP_SECONDS):
        for i in range(n_setup, n_setup + 4 * ke):
            step(i + 1, args.key_every > 0 and i % ke == 0)
        n_setup += 4 * ke
        rounds += 1
        barrier()
    base0 = n_setup    # (the frame sequence continues through set-up, warm-up and the timed region: no jump for the tracker)
    for i in range(args.warmup):
        step(base0 + i + 1, args.key_every > 0 and i % args.key_every == 0)
    barrier()
    ba_ws.worker_stats(), ic_ws.worker_stats()   # (reset: the sums below cover the timed region only)
    if upd_pts is not None:
        upd_pts["runs"] = 0
    t_begin = time.perf_counter()
    t_step_max, i_step_max, t_prev = 0.0, -1, t_begin
    for i in range(args.steps):
        # key frames: the first frame of the timed region and every KEY_EVERY-th after it (K / KEY_EVERY solves of each
        # kind in K frames, all completed before the clock stops)
        step(base0 + args.warmup + i + 1, args.key_every > 0 and i % args.key_every == 0)
        t_now = time.perf_counter()
        if t_now - t_prev > t_step_max:
            t_step_max, i_step_max = t_now - t_prev, i
        t_prev = t_now
    t_host = time.perf_counter() - t_begin
    barrier()
    dt = time.perf_counter() - t_begin
    wj, wi = ba_ws.worker_stats(), ic_ws.worker_stats()
    upd_info = None
    if upd_pts is not None:
        cnt = upd_pts["d_counts"].cpu().tolist()
        upd_info = {"what": "RobustBundleRTS::updateNewPosesPoints behind every finished joint BA (cs_ba_completed read between frames): "
                            "one launch (cs_update_new_poses_points_dev) over all map points, on a copy of the map",
                    "runs_in_timed_region": upd_pts["runs"], "static_points_retriangulated_last_run": cnt[0],
                    "dynamic_points_retriangulated_last_run": cnt[1]}
    solve_duty = {"what": "time the key-frame solves held their workspaces' streams inside the timed region (GPU clock, from the moment the "
                          "frame they wait for was done), against the region's length: which chain bounds the loop",
                  "joint_ba": {"solves": wj[0], "ms_total": wj[1], "ms_max": wj[3], "ms_parse_total": wj[4], "share_of_timed_region": wj[1] / (dt * 1e3)},
                  "inter_camera": {"solves": wi[0], "ms_total": wi[1], "ms_max": wi[3], "share_of_timed_region": wi[1] / (dt * 1e3)}}
    with_upload = None
    if not args.no_upload_leg and not args.serial:
        # the same loop once more, the images coming from pinned host memory every frame (same key-frame cadence, same drain)
        # one pinned ring entry per frame: the cameras' images back to back, as capture threads writing into cs_pinned_alloc'd
        # memory would leave them -> ONE host-to-device copy per frame
        h_frames = torch.from_numpy(np.stack([frames[c] for c in my_cams], axis=1).copy()).pin_memory()   # [frame][camera][H][W]
        i0 = base0 + args.warmup + args.steps + 1
        stage(i0)
        stage(i0 + 1)
        for i in range(args.warmup):
            step(i0 + i, args.key_every > 0 and i % args.key_every == 0, upload=True)
        barrier()
        tu = time.perf_counter()
        for i in range(args.steps):
            step(i0 + args.warmup + i, args.key_every > 0 and i % args.key_every == 0, upload=True)
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
        replay_base = i0 + args.warmup + args.steps - 1
    else:
        replay_base = base0 + args.warmup + args.steps
    gc.enable()
    pg_info = None
    if pg is not None:
        pg.status(ba_s.cuda_stream)      # raises if a graph failed
        moved = (d_pgNT - d_pgT).abs().max().item()
        pg_info = dict(pg.counts(), max_non_key_translation_change=moved)
    gathered_info = None
    if world > 1:
        # what the last frame's all-gather delivered, checked against its owners: every rank sums the words of the records of its
        # OWN cameras as it packed them; the sums travel through torch.distributed; every rank compares all 8 gathered records
        last_b = replay_base & 1
        w1 = multicam.record_words(N_FEAT)
        own = []
        for i in range(nc):
            ws_ = d_dests[last_b][i].to(torch.int64).sum() + d_R[last_b][i].view(torch.int32).to(torch.int64).sum() + \
                d_t[last_b][i].view(torch.int32).to(torch.int64).sum()
            own.append(ws_)
        own_t = torch.stack(own).to("cpu" if dist_backend != "nccl" else dev)
        all_t = [torch.zeros_like(own_t) for _ in range(world)]
        dist.all_gather(all_t, own_t)
        sums = torch.cat(all_t).cpu().tolist()
        ok = True
        for g in range(N_CAMS):
            fw, Rg, tg = xchg.unpack(g, device=local_rank)
            got = int(fw.to(torch.int64).sum().item() + Rg.view(torch.int32).to(torch.int64).sum().item() + tg.view(torch.int32).to(torch.int64).sum().item())
            ok = ok and got == int(sums[g])
        snapR = torch.cat([ic_start_snap[12 * g: 12 * g + 9] for g in range(N_CAMS)])
        gathered_info = {"cameras_checked": N_CAMS, "records_match_owner": bool(ok), "record_bytes": 4 * w1,
                         "intercam_start_is_gathered_pose": bool(torch.equal(d_iR, snapR)),
                         "intercam_start_differs_from_prebaked": bool((d_iR.cpu().numpy() != ic["Rs0"].reshape(-1)).any())}
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    grp.synchronize()
    last = replay_base & 1
    n_live = [int((d.cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum()) for d in d_dests[last]]
    pose_ok = d_ok.cpu().numpy().tolist()
    from coslam_amd.pose import IntraCamPoseOption
    _opts = [IntraCamPoseOption.from_buffer_copy(d_opt[i].cpu().numpy().tobytes()) for i in range(nc)]
    pose_iters = [[o.nIterRW, o.nIterLM] for o in _opts]     # re-weighting rounds, LM steps of the last round (last frame)
    pose_npts = d_npts.cpu().numpy().tolist()
    # how far the device-resident poses are from the synthetic ground truth of the last frame (data-coupled pose leg)
    f_last = order[replay_base % len(order)]
    Rl, tl_ = d_R[last].cpu().numpy(), d_t[last].cpu().numpy()
    pose_err = max(float(np.abs(tl_[i] - sc.pose(c, f_last)[1]).max()) for i, c in enumerate(my_cams))
    win_info = None
    if ba_win is not None:
        wC, wP, wO, _, wkf = ba_win.last_problem()
        ba_ws.set_sizes(wC, wP, wO)
        win_info = {"cameras": wC, "points": wP, "measurements": wO, "key_frames": wkf,
                    "what": "parsed on the device from the last 5 key frames' hand-back records and solved poses (cs_ba_window_*)"}
    _, _, _, _, st_j = ba_ws.download() if (world == 1 or not ba_sliced) else (None, None, None, None, None)
    _, _, _, _, st_i = ic_ws.download()

    # ---- roofline of the dominant kernel: the persistent gain tracker of all cameras of this rank (one launch per frame).
    # Timed with HIP events on the stream it is launched on, over a replay of the same frames after the timed region.
    roof = None
    replayed = 100
    if rank == 0:
        trks[0].set_profiling(True)
        n_prof = 100   # (also for a short --steps run: the first frames after the switch to profiling are slower)
        base = replay_base
        for i in range(n_prof):
            f, fn = order[(base + i + 1) % len(order)], order[(base + i + 2) % len(order)]
            if prefetch:
                grp.prefetch_dev(img_ptrs[fn])
            grp.redetect_dev(img_ptrs[f], dest_ptrs[0], cnt_ptrs)
            grp.advanceFrame()
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
        pmc_file = os.path.join(ROOT, "profiles", "r03_tracker_pmc.json")
        if not os.path.exists(pmc_file):
            pmc_file = os.path.join(ROOT, "profiles", "r02_tracker_pmc.json")
        pj = json.load(open(pmc_file)) if (nc == N_CAMS and os.path.exists(pmc_file)) else None
        if pj is not None:   # HBM bytes per frame's worth of launches from the committed --pmc passes (8 cameras per launch there)
            traffic = pj["traffic_bytes_per_launch"] / launches
            traffic_src = os.path.relpath(pmc_file, ROOT) + " (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE: bytes of all 8 cameras / launches per frame)"
            if "valu_wave_insts_per_launch" in pj:
                # the second roofline: the kernel is bound by VALU issue, not by HBM.  A wave64 VALU
                # instruction holds its SIMD's issue port for 4 cycles; 1024 SIMDs.
                insts = pj["valu_wave_insts_per_launch"] / launches
                floor_us = insts * 4.0 / (1024 * pj.get("sclk_ghz", 2.4) * 1e3)
                valu = {"insts": insts, "floor_us": floor_us, "frac": floor_us / avg_us, "unit": "wave64 VALU instructions per launch",
                        "source": os.path.relpath(pmc_file, ROOT) + " (SQ_INSTS_VALU, separate --pmc pass)"}
        fused_kernel = prof["launches_per_frame"] <= N_CAMS
        roof = {"bound": "hbm", "kernel": "k_track_rows_fused" if fused_kernel else "k_track_rows_pass",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes / launches,
                "avg_launch_us": avg_us, "launches_per_frame": launches, "frames_timed": prof["frames"],
                "cameras_per_launch": nc / launches, "valu": valu}
        if launches > 1 and fused_kernel:
            # the same kernel with every camera in ONE launch (two waves per SIMD: its own best configuration, slower for the loop)
            for t in trks:
                t.set_cu_count(256)
            for i in range(20):   # (the frame sequence continues where the replay above stopped: base + 100)
                f, fn = order[(base + 100 + i + 1) % len(order)], order[(base + 100 + i + 2) % len(order)]
                if prefetch:
                    grp.prefetch_dev(img_ptrs[fn])
                grp.redetect_dev(img_ptrs[f], dest_ptrs[0], cnt_ptrs)
                grp.advanceFrame()
            trks[0].set_profiling(True)
            for i in range(n_prof):
                f, fn = order[(base + 120 + i + 1) % len(order)], order[(base + 120 + i + 2) % len(order)]
                if prefetch:
                    grp.prefetch_dev(img_ptrs[fn])
                grp.redetect_dev(img_ptrs[f], dest_ptrs[0], cnt_ptrs)
                grp.advanceFrame()
            p1 = trks[0].get_profile()
            trks[0].set_profiling(False)
            replayed = 100 + 20 + n_prof
            if p1["launches_per_frame"] == 1:
                us1 = p1["tracker_us_total"] / max(p1["frames"], 1)
                a1 = alg_bytes / (us1 * 1e-6) / 1e9
                roof["all_cameras_in_one_launch"] = {"avg_launch_us": us1, "achieved": a1, "frac": a1 / HBM_PEAK_GBS,
                                                     "algorithmic_bytes_per_launch": alg_bytes,
                                                     "valu_frac": None if valu is None else valu["floor_us"] * launches / us1}

    # ---- secondary key: cfg2 (BASELINE.json configs[1]) = ONE camera on the GPU, KLT + hand-back + pose per frame, no
    # key-frame solves; same kernels through the single-handle entry points.  Not the headline; kept for continuity.
    cfg2 = None
    if rank == 0 and n_gpus == 1 and not args.serial and not args.no_pose:
        if not args.klt_cus:
            for t in trks:
                t.set_cu_count(256)   # (the headline loop may have budgeted the tracker for fewer cameras per launch)
        n2 = min(args.steps, 200)
        base = replay_base + replayed   # (continues the frame sequence of the replays above)
        k0 = trks[0]
        hb1 = [handback_cams(hb_cams(0)[:1]), handback_cams(hb_cams(1)[:1])]

        def step1(i):
            f, fn = order[i % len(order)], order[(i + 1) % len(order)]
            b = i & 1
            klt_s.wait_event(dest_free[b])
            if prefetch:
                k0.prefetch_dev(img_ptrs[fn][0])
            k0.redetect_dev(img_ptrs[f][0], dest_ptrs[b][0], cnt_ptrs[0])
            k0.advanceFrame()
            klt_done[b].record(klt_s)
            pose_s.wait_event(klt_done[b])
            handback_dev(pose_s.cuda_stream, hb1[b], N_FEAT, W, H, N_COL_BLK, N_ROW_BLK, PTS_STRIDE,
                         device=local_rank, frame=i)
            src, dst = (i + 1) & 1, i & 1
            intraCamEstimate_batch_dev(pose_s.cuda_stream, 1, PTS_STRIDE, d_K.data_ptr(), d_R[src].data_ptr(),
                                       d_t[src].data_ptr(), d_npts.data_ptr(), 0, d_Ms.data_ptr(), d_ms.data_ptr(), 10.0,
                                       d_R[dst].data_ptr(), d_t[dst].data_ptr(), d_opt.data_ptr(), d_ok.data_ptr(),
                                       device=local_rank)
            dest_free[b].record(pose_s)

        for i in range(20):
            step1(base + i + 1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(n2):
            step1(base + 20 + i + 1)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        cfg2 = {"workload": "cfg2: 1 camera 640x480 x 2000 slots, KLT (redetect, prefetch) + hand-back + intraCamEstimate "
                            "per frame, no key-frame solves", "camera_frames_per_s": n2 / dt2, "frames": n2}

    # ---- secondary key: the sliding-window BA of cfg5 (BASELINE.json configs[4]: 4 cameras x 30 key frames = 120 poses of which 8
    # fixed, 5000 points, every point in every key frame: 600 k measurements, reduced system of order 672), one full robust solve
    cfg5 = None
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        from coslam_amd.synth import make_ba_problem

        pr5 = make_ba_problem(n_cams=120, n_pts=5000, W=1920, H=1080, noise=0.3, outlier_frac=0.01, outlier_mag=40.0, n_cams_con=8,
                              n_pts_con=2, seed=55)
        p5, c5, x5 = csr(pr5)
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

    # ---- secondary key: the KLT stage of cfg5 (BASELINE.json configs[4]: 4 cameras 1920 x 1080 x 5000 slots (100 x 50), here all
    # four on ONE GPU as a camera group): redetect + prefetch per frame, HIP-event time of the tracker stage
    cfg5_klt = None
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        from coslam_amd.synth import Scene as _Scene

        W5, H5, L5, FW5, FH5, C5, NF5 = 1920, 1080, 4, 100, 50, 4, 4
        sc5 = _Scene(C5, W5, H5, 12000, seed=0xC051A + 5)
        fr5 = [torch.from_numpy(np.stack([sc5.render(c, f) for f in range(NF5)])).to(dev) for c in range(C5)]
        ord5 = list(range(NF5)) + list(range(NF5 - 2, 0, -1))
        t5s = []
        for _ in range(C5):
            t = coslam_amd.KLT_SequenceTracker(coslam_amd.KLT_SequenceTrackerConfig(
                nIterations=10, nLevels=L5, levelSkip=1, windowWidth=7, trackWithGain=1, minCornerness=3000.0, convergenceThreshold=1.0,
                SSD_Threshold=20000.0, minDistance=8), device=local_rank)
            t.allocate(W5, H5, L5, FW5, FH5)
            t.set_concurrent_handles(C5)   # (the headline's 8 trackers are still alive but idle: only this group's launches overlap)
            t5s.append(t)
        g5 = coslam_amd.KLT_TrackerGroup(t5s)
        g5.set_stream(klt_s.cuda_stream)
        dd5 = [torch.zeros(FW5 * FH5 * 5, dtype=torch.int32, device=dev) for _ in range(C5)]
        cc5 = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(C5)]
        dp5, cp5 = [d.data_ptr() for d in dd5], [c.data_ptr() for c in cc5]
        g5.detect_dev([f[0].data_ptr() for f in fr5], dp5, cp5)
        g5.advanceFrame()

        def frame5(i):
            a, b = ord5[(i + 1) % len(ord5)], ord5[(i + 2) % len(ord5)]
            g5.prefetch_dev([f[b].data_ptr() for f in fr5])
            g5.redetect_dev([f[a].data_ptr() for f in fr5], dp5, cp5)
            g5.advanceFrame()

        for i in range(8):
            frame5(i)
        g5.synchronize()
        t5s[0].set_profiling(True)
        n5 = 40
        tk = time.perf_counter()
        for i in range(n5):
            frame5(8 + i)
        g5.synchronize()
        dtk = time.perf_counter() - tk
        p5 = t5s[0].get_profile()
        t5s[0].set_profiling(False)
        live5 = [int((d.cpu().numpy().view(coslam_amd.KLT_TrackedFeature)["status"] >= 0).sum()) for d in dd5]
        per_feat5 = L5 * 2 * (2 * 3 + 2) ** 2 * 6 + 2 * 12
        trk_us5 = p5["tracker_us_total"] / max(p5["frames"], 1)
        cfg5_klt = {"workload": "cfg5 KLT: 4 cameras 1920x1080 x 5000 slots (100x50) as one camera group on one GPU, 4 levels, 7x7, 10 it/level "
                                "with gain, redetect + prefetch per frame", "frames_per_s": n5 / dtk, "camera_frames_per_s": C5 * n5 / dtk,
                    "us_per_frame": dtk / n5 * 1e6, "tracker_stage_us": trk_us5, "tracker_launches_per_frame": p5["launches_per_frame"],
                    "tracker_algorithmic_GBps": per_feat5 * FW5 * FH5 * C5 / (trk_us5 * 1e-6) / 1e9, "live_features": live5,
                    "hbm_frac_per_launch": per_feat5 * FW5 * FH5 * C5 / (trk_us5 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "pmc": "profiles/r03_cfg5_klt_pmc.json (traffic 2.98 x the algorithmic bytes, VALU issue 0.50 of the launch)"}
        g5.close()
        for t in t5s:
            t.close()
        del fr5

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cores = os.cpu_count() or 1
        nt = min(cores, N_CAMS)
        v1, n1, dt1 = cpu_baseline(sc, frames, joint, ic, 1, 12.0, not args.no_register, not args.no_posegraph)
        vN, nN, dtN = cpu_baseline(sc, frames, joint, ic, nt, 12.0, not args.no_register, not args.no_posegraph) if nt > 1 else (v1, n1, dt1)
        cpu = {"value": vN, "unit": "frames/s", "cores": nt, "kind": "port",
               "sample": f"{nN} frames of the same 8-camera workload on {nt} threads (cameras in parallel) in {dtN:.1f} s; "
                         f"{n1} frames on 1 thread in {dt1:.1f} s (oracle/: C restatement, gcc -O2); host has {cores} cores",
               "value_1_thread": v1, "host_cores": cores}

    # ---- the same loop driven from C++ through the C-ABI only (north_star: "Host stays C++"): tools/cxx/frame_loop.cpp, its
    # own process, the workload handed over as a file; same steps / warm-up / key-frame cadence / drain
    cxx = None
    if rank == 0 and n_gpus == 1 and not args.no_cxx_loop and not args.serial and args.key_every == KEY_EVERY:
        import subprocess
        import tempfile

        exe = os.path.join(ROOT, "tools", "cxx", "frame_loop.bin")
        if os.path.exists(exe):
            torch.cuda.synchronize()
            with tempfile.TemporaryDirectory() as td:
                wl = os.path.join(td, "workload.bin")
                export_workload(wl, sc, frames, joint, ic, args.klt_cams_per_launch)
                pr = subprocess.run([exe, wl, str(args.steps), str(args.warmup), str(args.klt_cams_per_launch),
                                     "1" if ba_win is not None else "0"], capture_output=True, text=True, timeout=600)
            if pr.returncode == 0 and pr.stdout.strip().startswith("{"):
                cxx = json.loads(pr.stdout.strip().splitlines()[-1])
                cxx["what"] = ("tools/cxx/frame_loop.cpp: the headline loop from C++ through include/coslam_hip.h only (no Python, no "
                               "torch), own process, images resident in HBM")
            else:
                cxx = {"error": (pr.stderr or pr.stdout)[-400:]}
        else:
            cxx = {"error": "tools/cxx/frame_loop.bin not built (python -c 'import __graft_entry__ as g; g.build()')"}

    if rank == 0:
        out = {
            "metric": "frames/sec for track+local-BA loop, 8 cams 640x480 x 2000 feats (one frame = all 8 cameras)",
            "value": args.steps / dt, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32 (KLT, f16 pyramid storage) + f64 (pose, BA)",
            "data": "synthetic",
            "config": {"workload": "8 cams 640x480 x 2000 KLT slots (50x40), 4-level pyramid, 7x7 window, 10 it/level with "
                                   "gain, redetect every frame; on-device hand-back + intraCamEstimate of all 8 cameras "
                                   "every frame (fed by the tracker's output)"
                                   + ("" if pose_upd is None else ", then poseUpdate3D's Mahalanobis gate + seqTriangulate refinement of the "
                                      "map points and the dynamic-point test (64-frame history)") + "; map-point registration search every frame "
                                   f"(active + current static, {P_REG} points each x 8 cams x 2000 slots); "
                                   f"every {KEY_EVERY}th frame: joint local BA "
                                   + (f"C=40 (16 fixed), parsed on the device from the last 5 key frames' tracked features and poses "
                                      f"(last: {win_info['points']} pts x {win_info['measurements']} meas), maxIter 2 / inner 10"
                                      if win_info is not None else
                                      f"C=40 (16 fixed) x {len(joint['pts0'])} pts x {len(joint['obs_cam'])} meas (pre-baked), maxIter 2 / inner 10")
                                   + ("" if args.no_posegraph else ", followed on its stream by the pose-graph relaxation of the "
                                      "window's non-key frames (21-frame chain per camera, 5 key frames fixed)") + ", and "
                                   f"inter-camera solve C=8 free, {ic['n_static']} static pts fixed + {ic['n_dynamic']} dynamic, "
                                   "sigma 6, 3 x 40; N>1: cameras sharded 8/N per GPU, all-gather of features+pose per frame, "
                                   + ("joint BA sliced by points with an all-reduce per LM step" if ba_sliced else
                                      "joint BA solved by every rank from the gathered measurements (latency-bound: no collective; --ba-sliced 1 slices it)"),
                       "cameras": N_CAMS, "cameras_per_gpu": nc, "camera_frames_per_s": N_CAMS * args.steps / dt,
                       "live_features_last_frame": n_live, "pose_ok": pose_ok, "pose_correspondences": pose_npts, "pose_rounds_and_last_lm_steps": pose_iters,
                       "pose_translation_error_vs_truth": pose_err,
                       "joint_ba_problem": win_info if win_info is not None else {
                           "cameras": len(joint["Rs0"]), "points": len(joint["pts0"]), "measurements": len(joint["obs_cam"]),
                           "what": "pre-baked synthetic problem, re-solved from the same start at every key frame"},
                       "joint_ba_last": None if st_j is None else {"lm_steps": st_j.nIterTotal, "outliers": st_j.nOutliers,
                                                                  "cost0": st_j.cost0, "cost": st_j.cost},
                       "intercam_last": {"lm_steps": st_i.nIterTotal, "outliers": st_i.nOutliers, "cost0": st_i.cost0,
                                         "cost": st_i.cost},
                       "frame_front_prefetch": bool(prefetch), "secondary_cfg2": cfg2, "secondary_cfg5_ba": cfg5, "secondary_cfg5_klt": cfg5_klt,
                       "posegraph_last": pg_info,
                       "register_candidates_last_frame": None if args.no_register else
                       {"active": int((reg_out[0]["slot"] >= 0).sum().item()), "current_static": int((reg_out[1]["slot"] >= 0).sum().item()),
                        "already_attached": int((reg_out[1]["slot"] == -1).sum().item()),
                        "current_static_mergeable_over_the_whole_track": None if pose_upd is None else int((d_mergeable == 1).sum().item())},
                       "pose_update": None if pose_upd is None else {
                           "what": "poseUpdate3D's gate + seqTriangulate over all static mapped features and detectDynamicFeaturePoints over "
                                   "all unmapped / dynamic tracks, every frame, one launch for the rank's cameras (cs_pose_update_frame_dev)",
                           "history_frames": pose_upd.frames, "map_points_uncertain": int((d_mapflags & 4).ne(0).sum().item()),
                           "map_points_classify": None if args.no_classify else {
                               "what": "CoSLAM::mapPointsClassify(12.0) every frame behind the gate (cs_map_points_classify_dev)",
                               "examined_last_frame": int(d_cls_counts[0].item()), "became_false_last_frame": int(d_cls_counts[1].item()),
                               "map_points_false": int((d_mapflags & 2).ne(0).sum().item()),
                               "map_points_dynamic": int(((d_mapflags & 3) == 1).sum().item())},
                           "map_points_refined": int((d_map - torch.from_numpy(sc.points).to(dev)).abs().amax(dim=1).gt(0).sum().item()),
                           "features_dynamic_last_frame": [int(v) for v in ((d_isstatic == 0) & (d_state >= 0)).sum(dim=1).cpu().tolist()],
                           "static_mapped_features_last_frame": [int(v) for v in ((d_state >= 0) & (d_slot2map >= 0)).sum(dim=1).cpu().tolist()]},
                       "update_new_poses_points": upd_info,
                       "key_frame_solves_duty": solve_duty,
                       "host_enqueue_ms_per_step": t_host / args.steps * 1e3, "host_enqueue_ms_max_step": t_step_max * 1e3, "host_enqueue_max_at_step": i_step_max, "tracker_stream_cus": args.klt_cus or "all",
                       "ncc_matching": None if ncc is None else {
                           "every_frames": NCC_EVERY, "camera_pairs_per_run": nc - 1, "runs": ncc["runs"],
                           "output": "dense matrices (cs_ncc_epi_mat_dev)" if args.ncc_dense else "list of the passing pairs (cs_ncc_epi_pairs_dev)",
                           "pairs_kept_last_run": (int((ncc["score"] != -1.0).sum().item()) if args.ncc_dense
                                                   else [int(v) for v in ncc["pair_count"].cpu().tolist()]),
                           "unmapped_features_last_run": [int(v) for v in ncc["valid"].sum(dim=1).cpu().tolist()]},
                       "gathered_records": gathered_info, "with_upload": with_upload, "cxx_frame_loop": cxx,
                       "collectives": None if world == 1 else ("libcoslam_hip RCCL (C-ABI)" if native else "torch.distributed " + dist_backend),
                       "streams": "one stream (--serial)" if args.serial else
                       "tracker group | hand-back + pose (event-ordered behind the tracker of the same frame) | "
                       "inter-camera solve and joint local BA each on its workspace's worker thread + stream "
                       "(cs_ba_solve_async, like the reference's BA worker thread)"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
