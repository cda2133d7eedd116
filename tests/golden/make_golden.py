#!/usr/bin/env python
"""Generates tests/golden/*.npz.

pose_golden.npz  -- inputs and outputs of the REFERENCE's own intraCamEstimate (src/slam/SL_IntraCamPose.cpp compiled in
                    place into oracle/_ref/libintracam_ref.so by oracle/Makefile).  Needs /root/reference; run in the build
                    container only.  These vectors pin oracle/pose_oracle.c and the HIP pose kernel to the reference.
klt_golden.npz   -- small KLT cases (pyramid texels, detection list, two tracked frames) produced by oracle/klt_oracle.c
                    (the with-gain case in the oracle's "tree" summation mode, which the HIP tracker matches bit for bit).
                    A regression fixture in the KERNELS' summation order; the shaders' own outputs are in cgklt_golden.npz.
cgklt_golden.npz -- the KLT passes run by the REFERENCE's own fragment programs (src/tracking/CGKLT/Shaders/*.cg compiled in place
                    into oracle/_ref/libcgklt_ref.so by oracle/build_cgref.sh; rasteriser + GL texture model in
                    oracle/ref_shim/cg/): pyramids of three frames (two scenes), cornerness after klt_detector_pass1/2, after the
                    suppression + non-max passes, the HistoPyramid point list, and detect / redetect / redetect sequences with and
                    without gain in which EVERY pass was produced by the shaders (the frame logic around them is
                    oracle.SequenceTracker's restatement of v3d_gpuklt.cpp:650-889, asserted here to reproduce the shader
                    outputs bit for bit at every step).  Pins oracle/klt_oracle.c (serial mode) and, on the GPU box, the HIP
                    pyramid / detector (bit for bit) and trackers (<= 0.02 px, status flips only at recorded threshold margins).
ba_golden.npz    -- cfg1-shaped BA problem solved by oracle/ba_oracle.c (our definition; parity unpinned).
register_golden.npz -- a feature list, 400 (m, var, maxDist) queries and the answers of the REFERENCE's own
                    searchMahaNearestFeatPt (src/app/SL_SingleSLAM.cpp:1141-1164 compiled in place into
                    oracle/_ref/ref_register_test, `golden` mode, CPU).  Pins oracle/register_oracle.c's search and,
                    on the GPU box, the registration kernel.
ncc_golden.npz   -- two small images, feature positions, F and the REFERENCE's own NCC blocks (NCCBlock::computeScaled,
                    src/slam/SL_NCCBlock.cpp:15-54) and epipolar / NCC matrices (getEpiNccMat,
                    src/slam/SL_FeatureMatching.cpp:3-46; matchNCCBlock) from oracle/_ref/ref_ncc_test `golden` (CPU).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from coslam_amd.klt import KLT_SequenceTrackerConfig  # noqa: E402
from coslam_amd.synth import Scene, make_ba_problem, rodrigues  # noqa: E402


def pose_cases():
    out = {}
    sc = Scene(1, 640, 480, 3000, seed=77)
    rng = np.random.default_rng(5)
    n_cases = 6
    for c in range(n_cases):
        npts = [192, 192, 64, 12, 300, 192][c]
        R, t = sc.pose(0, 2 * c)
        uv, vis = sc.project(0, 2 * c)
        idx = np.nonzero(vis)[0][:npts]
        Ms = sc.points[idx]
        ms = uv[idx] + 0.5 * rng.standard_normal((len(idx), 2))
        ms[: max(1, npts // 20)] += 30 * rng.standard_normal((max(1, npts // 20), 2))
        R0 = R @ rodrigues(0.01 * rng.standard_normal(3))
        t0 = t + 0.03 * rng.standard_normal(3)
        prev = None if c % 2 == 0 else np.abs(rng.standard_normal(len(idx))) * 5
        ok, Rr, tr, st = oracle.ref_intracam_estimate(sc.K, R0, t0, len(idx), prev, Ms, ms, 10.0)
        out[f"K{c}"], out[f"R0{c}"], out[f"t0{c}"], out[f"Ms{c}"], out[f"ms{c}"] = sc.K, R0, t0, Ms, ms
        out[f"prev{c}"] = np.zeros(0) if prev is None else prev
        out[f"ok{c}"], out[f"R{c}"], out[f"t{c}"] = np.array([ok]), Rr, tr
        out[f"stats{c}"] = np.array([st["err"], st["errRW"], st["lambda_"], st["nIterLM"], st["nIterRW"], st["retTypeLM"]])
    out["n"] = np.array([n_cases])
    return out


def klt_cases():
    W, H, L, fw, fh = 160, 120, 3, 10, 8
    sc = Scene(1, W, H, 260, seed=9, sigma=1.4)
    imgs = np.stack([sc.render(0, f) for f in range(3)])
    out = {"images": imgs, "dims": np.array([W, H, L, fw, fh])}
    for gain in (0, 1):
        cfg = KLT_SequenceTrackerConfig(nIterations=6, nLevels=L, levelSkip=1, windowWidth=7, trackWithGain=gain,
                                        minCornerness=800.0, convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=5)
        # window sums in the HIP trackers' fixed order ("tree" mode), with and without gain -> the GPU test is bit-exact
        o = oracle.SequenceTracker(cfg, sum_mode=1)
        o.allocate(W, H, L, fw, fh)
        n0, d0 = o.detect(imgs[0])
        out[f"pyr{gain}"] = o.read_pyramid()
        out[f"corner{gain}"] = o.read_cornerness()
        o.advanceFrame()
        n1, d1 = o.redetect(imgs[1])
        o.advanceFrame()
        n2, d2 = o.track(imgs[2])
        out[f"n{gain}"] = np.array([n0, n1, n2])
        for k, d in enumerate((d0, d1, d2)):
            out[f"status{gain}_{k}"], out[f"pos{gain}_{k}"], out[f"gain{gain}_{k}"] = d["status"], d["pos"], d["gain"]
    return out


def ba_case():
    pr = make_ba_problem()
    P = len(pr["pts0"])
    ptr, cam, xy, _ = oracle.csr_by_point(P, pr["obs_pt"], pr["obs_cam"], pr["obs_xy"])
    R, T, M, outl, st = oracle.ba_robust(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy, 2, 2, 6.0, 2, 10)
    return dict(Ks=pr["Ks"], Rs0=pr["Rs0"], ts0=pr["ts0"], pts0=pr["pts0"], ptr=ptr, cam=cam, xy=xy, R=R, T=T, M=M,
                outlier=outl, stats=np.array([st.cost0, st.cost, st.nIterTotal, st.nOuter, st.nOutliers]))


def register_case():
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_register_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_register_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "reg.bin")
        subprocess.run([exe, "golden", path], check=True)
        raw = open(path, "rb").read()
    N, Q, none, _ = np.frombuffer(raw, dtype=np.int32, count=4)
    o = 16
    xy = np.frombuffer(raw, dtype=np.float64, count=2 * N, offset=o)
    o += 16 * N
    state = np.frombuffer(raw, dtype=np.int32, count=N, offset=o)
    o += 4 * N
    q = np.frombuffer(raw, dtype=np.float64, count=7 * Q, offset=o).reshape(Q, 7)
    o += 56 * Q
    ans = np.frombuffer(raw, dtype=np.int32, count=Q, offset=o)
    return dict(xy=xy.copy(), state=state.copy(), m=q[:, 0:2].copy(), var=q[:, 2:6].copy(), maxDist=q[:, 6].copy(),
                slot=ans.copy(), empty_frame_returns_null=np.int32(1 - none))


def ncc_case():
    """the reference's own NCCBlock::computeScaled / getEpiNccMat (oracle/_ref/ref_ncc_test golden, CPU)"""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_ncc_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_ncc_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ncc.bin")
        subprocess.run([exe, "golden", path], check=True)
        raw = open(path, "rb").read()
    W, H, M, N = np.frombuffer(raw, dtype=np.int32, count=4)
    o = 16
    F = np.frombuffer(raw, dtype=np.float64, count=9, offset=o)
    o += 72
    scale, epiMax, nccMin = np.frombuffer(raw, dtype=np.float64, count=3, offset=o)
    o += 24
    out = dict(F=F.copy(), scale=scale, epiMax=epiMax, nccMin=nccMin)
    for tag, n in (("1", M), ("2", N)):
        out["img" + tag] = np.frombuffer(raw, dtype=np.uint8, count=W * H, offset=o).reshape(H, W).copy()
        o += W * H
        out["x" + tag] = np.frombuffer(raw, dtype=np.float64, count=n, offset=o).copy()
        o += 8 * n
        out["y" + tag] = np.frombuffer(raw, dtype=np.float64, count=n, offset=o).copy()
        o += 8 * n
        out["blocks" + tag] = np.frombuffer(raw, dtype=np.uint8, count=128 * n, offset=o).reshape(n, 128).copy()
        o += 128 * n
        out["abc" + tag] = np.frombuffer(raw, dtype=np.float64, count=4 * n, offset=o).reshape(n, 4).copy()
        o += 32 * n
        out["valid" + tag] = np.frombuffer(raw, dtype=np.int32, count=n, offset=o).copy()
        o += 4 * n
    out["epi"] = np.frombuffer(raw, dtype=np.float64, count=M * N, offset=o).reshape(M, N).copy()
    o += 8 * M * N
    out["ncc"] = np.frombuffer(raw, dtype=np.float64, count=M * N, offset=o).reshape(M, N).copy()
    return out


def posegraph_case():
    """the reference's own GlobalPoseGraph::computeNewCameraRotations / computeNewCameraTranslations on 9 graphs
    (oracle/_ref/ref_posegraph_test golden, CPU).  Flat arrays: graph g owns nodes [node_ptr[g], node_ptr[g+1]) and edges
    [edge_ptr[g], edge_ptr[g+1]); id1 / id2 are local to the graph."""
    import struct
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_posegraph_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_posegraph_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "pg.bin")
        subprocess.run([exe, "golden", path], check=True)
        raw = open(path, "rb").read()
    ng, _ = struct.unpack_from("ii", raw, 0)
    o = 8
    node_ptr, edge_ptr = [0], [0]
    fixed, R, t, nR, nt, id1, id2, eR, eT = [], [], [], [], [], [], [], [], []
    for _g in range(ng):
        n, e = struct.unpack_from("ii", raw, o)
        o += 8
        for _i in range(n):
            fixed.append(struct.unpack_from("i", raw, o)[0])
            v = np.frombuffer(raw, np.float64, 24, o + 4)
            o += 4 + 192
            R.append(v[:9]), t.append(v[9:12]), nR.append(v[12:21]), nt.append(v[21:24])
        for _k in range(e):
            a, b = struct.unpack_from("ii", raw, o)
            v = np.frombuffer(raw, np.float64, 12, o + 8)
            o += 8 + 96
            id1.append(a), id2.append(b), eR.append(v[:9]), eT.append(v[9:])
        node_ptr.append(node_ptr[-1] + n), edge_ptr.append(edge_ptr[-1] + e)
    assert o == len(raw)
    return dict(node_ptr=np.array(node_ptr, np.int32), edge_ptr=np.array(edge_ptr, np.int32), fixed=np.array(fixed, np.uint8),
                nodeR=np.array(R), nodeT=np.array(t), newR=np.array(nR), newT=np.array(nt), id1=np.array(id1, np.int32),
                id2=np.array(id2, np.int32), edgeR=np.array(eR), edgeT=np.array(eT))


def export_case():
    """the reference's own CoSLAM::exportResults (oracle/_ref/ref_export_test golden, CPU): the six files it writes and the
    arrays cs_export_results_v1 needs to write them again.  Map point ids are the addresses of that run's objects."""
    import struct
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_export_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_export_test missing: run `make -C oracle` where /root/reference exists")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        subprocess.run([exe, "golden", td], check=True, stdout=subprocess.DEVNULL)
        raw = open(os.path.join(td, "inputs.bin"), "rb").read()
        for n in ("input_videos.txt", "mappts.txt", "0_campose.txt", "1_campose.txt", "0_featpts.txt", "1_featpts.txt"):
            out["file_" + n] = np.frombuffer(open(os.path.join(td, "slam_results", "ref", n), "rb").read(), dtype=np.uint8).copy()
    nC, cur, nP, _ = struct.unpack_from("iiii", raw, 0)
    o = 16

    def take(dt, n):
        nonlocal o
        v = np.frombuffer(raw, dtype=dt, count=n, offset=o).copy()
        o += v.nbytes
        return v

    out.update(nCams=np.int32(nC), curFrame=np.int32(cur), ptId=take(np.int64, nP), ptM=take(np.float64, 3 * nP).reshape(nP, 3),
               ptCov=take(np.float64, 9 * nP).reshape(nP, 9))
    for c in range(nC):
        pl = int(take(np.int32, 1)[0])
        out[f"c{c}_path"] = take(np.uint8, pl)
        W, H, start, nposes = take(np.int32, 4)
        out[f"c{c}_whs"] = np.array([W, H, start], np.int32)
        out[f"c{c}_K"], out[f"c{c}_kc"] = take(np.float64, 9), take(np.float64, 5)
        out[f"c{c}_poseFrame"] = take(np.int32, nposes)
        out[f"c{c}_poseR"], out[f"c{c}_poseT"] = take(np.float64, 9 * nposes).reshape(-1, 9), take(np.float64, 3 * nposes).reshape(-1, 3)
        out[f"c{c}_featPtr"] = take(np.int32, int(take(np.int32, 1)[0]))
        nf = int(take(np.int32, 1)[0])
        out[f"c{c}_featId"], out[f"c{c}_featXY"] = take(np.int64, nf), take(np.float64, 2 * nf).reshape(-1, 2)
    assert o == len(raw)
    return out


def intercam_case():
    """the reference's own InterCamPoseEstimator::addMapPoints (oracle/_ref/ref_intercam_test golden, CPU): three scenes of cameras'
    records and the flattened vecPts3D / vecMeas2D it built from them."""
    import struct
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_intercam_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_intercam_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "ic.bin")
        subprocess.run([exe, "golden", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = [0]

    def ints(n):
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o[0]).copy()
        o[0] += 4 * n
        return v

    def dbls(n):
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o[0]).copy()
        o[0] += 8 * n
        return v

    out = {}
    (ns,) = ints(1)
    out["n_scenes"] = np.int32(ns)
    for sc in range(ns):
        nc, N, nMap, frame, W, H, ncb, nrb, pts_stride = (int(v) for v in ints(9))
        k = lambda n: f"s{sc}_{n}"   # noqa: E731
        out[k("dims")] = np.array([nc, N, nMap, frame, W, H, ncb, nrb, pts_stride], np.int32)
        out[k("mapPts")] = dbls(3 * nMap).reshape(nMap, 3)
        out[k("mapFlags")] = ints(nMap).astype(np.uint8)
        out[k("newPt")] = ints(nMap).astype(np.uint8)
        out[k("pointFeat")] = ints(nMap * nc).reshape(nMap, nc)
        xy, st, s2m, ft, sp = [], [], [], [], []
        for _ in range(nc):
            xy.append(dbls(2 * N)), st.append(ints(N)), s2m.append(ints(N)), ft.append(ints(N).astype(np.uint8)), sp.append(ints(2 * N))
        out[k("xy")], out[k("state")], out[k("slot2map")] = np.stack(xy), np.stack(st), np.stack(s2m)
        out[k("isStatic")], out[k("trackSpan")] = np.stack(ft), np.stack(sp)
        n_static, n_dyn, P, n_obs = (int(v) for v in ints(4))
        out[k("counts")] = np.array([n_static, n_dyn, P, n_obs], np.int32)
        out[k("pts")] = dbls(3 * P).reshape(P, 3)
        out[k("obs_ptr")], out[k("obs_cam")] = ints(P + 1), ints(n_obs)
        out[k("obs_xy")] = dbls(2 * n_obs).reshape(n_obs, 2)
        out[k("point_map")] = ints(P)
        cams = dbls(24 * nc).reshape(nc, 24)
        out[k("Rs")], out[k("Ts")], out[k("curR")], out[k("curT")] = cams[:, :9], cams[:, 9:12], cams[:, 12:21], cams[:, 21:24]
    assert o[0] == len(raw)
    return out


def newpts_case():
    """the reference's own featTracksFromMatches + NewMapPtsNCC::reconstructTracks + decidePointType (oracle/_ref/ref_newpts_test golden,
    CPU): three scenes of cameras, candidate features, given matches, dynamic points' features; the tracks in the reference's
    numbering and the new map points it made of them."""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_newpts_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_newpts_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "np.bin")
        subprocess.run([exe, "golden", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = [0]

    def ints(n):
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o[0]).copy()
        o[0] += 4 * n
        return v

    def dbls(n):
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o[0]).copy()
        o[0] += 8 * n
        return v

    out = {}
    (ns,) = ints(1)
    out["n_scenes"] = np.int32(ns)
    for sc in range(ns):
        nc, N, frame, W, H = (int(v) for v in ints(5))
        k = lambda n: f"s{sc}_{n}"   # noqa: E731
        out[k("dims")] = np.array([nc, N, frame, W, H], np.int32)
        out[k("K")] = dbls(9)
        Rt = dbls(12 * nc).reshape(nc, 12)
        out[k("R")], out[k("t")] = Rt[:, :9], Rt[:, 9:]
        xy, st = [], []
        for c in range(nc):
            xy.append(dbls(2 * N)), st.append(ints(N).astype(np.uint8))
            (nd,) = ints(1)
            out[k(f"dyn{c}")] = dbls(2 * int(nd)).reshape(-1, 2)
            (no,) = ints(1)
            out[k(f"other{c}")] = dbls(2 * int(no)).reshape(-1, 2)
        out[k("xy")], out[k("isStatic")] = np.stack(xy), np.stack(st)
        for a in range(nc - 1):
            (nm,) = ints(1)
            out[k(f"match{a}")] = ints(2 * int(nm)).reshape(-1, 2)
        (nt,) = ints(1)
        lens, flat = [], []
        for _ in range(int(nt)):
            (ln,) = ints(1)
            lens.append(int(ln)), flat.append(ints(2 * int(ln)).reshape(-1, 2))
        out[k("track_len")] = np.asarray(lens, np.int32)
        out[k("track_views")] = np.concatenate(flat) if flat else np.zeros((0, 2), np.int32)
        (nn,) = ints(1)
        M, cov, fl, ff, feat = [], [], [], [], []
        for _ in range(int(nn)):
            M.append(dbls(3)), cov.append(dbls(9))
            a_, b_ = ints(2)
            fl.append(int(a_)), ff.append(int(b_)), feat.append(ints(nc))
        out[k("new_M")], out[k("new_cov")] = np.stack(M), np.stack(cov)
        out[k("new_flags")], out[k("new_first")], out[k("new_feat")] = np.asarray(fl, np.uint8), np.asarray(ff, np.int32), np.stack(feat)
        out[k("reproj")] = dbls(nc * N).reshape(nc, N)
    assert o[0] == len(raw)
    return out


def decide_case():
    """the reference's own CoSLAM::curStaticPointsRegInGroup -- and, in the last two scenes, curDynamicPointsRegInGroup behind it, the
    order of currentMapPointsRegister -- (oracle/_ref/ref_decide_test golden, CPU): five scenes of cameras, pose histories, feature
    tracks and map points; which feature carries which point afterwards, every point's position / covariance."""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_decide_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_decide_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "d.bin")
        subprocess.run([exe, "golden", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = [0]

    def ints(n):
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o[0]).copy()
        o[0] += 4 * n
        return v

    def dbls(n):
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o[0]).copy()
        o[0] += 8 * n
        return v

    out = {}
    (ns,) = ints(1)
    out["n_scenes"] = np.int32(ns)
    for sc in range(ns):
        nC, Hh, N, nP, cur, W, H, with_dyn, with_merge = (int(v) for v in ints(9))
        (pv,) = dbls(1)
        k = lambda n: f"s{sc}_{n}"   # noqa: E731
        out[k("dims")] = np.array([nC, Hh, N, nP, cur, W, H], np.int32)
        out[k("with_dynamic")] = np.int32(with_dyn)   # curDynamicPointsRegInGroup ran behind the static points' registration
        out[k("with_merge")] = np.int32(with_merge)   # bMerge == true: checkUnify at a conflict, the points unified on a yes
        out[k("pixelVar")] = np.float64(pv)
        K, hR, hT = np.zeros((nC, 9)), np.zeros((nC, Hh, 9)), np.zeros((nC, Hh, 3))
        for c in range(nC):
            K[c] = dbls(9)
            for j in range(Hh):
                hR[c, j], hT[c, j] = dbls(9), dbls(3)
        out[k("K")], out[k("histR")], out[k("histT")] = K, hR, hT
        hXY = np.zeros((nC, Hh, 2 * N))
        span = np.full((nC, 2 * N), -1, np.int32)
        state = np.full((nC, N), -1, np.int32)
        is_static = np.ones((nC, N), np.uint8)
        s2m = np.full((nC, N), -1, np.int32)
        for c in range(nC):
            for s in range(N):
                L, st, m = (int(v) for v in ints(3))
                if L == 0:
                    continue
                xy = dbls(2 * L).reshape(L, 2)
                hXY[c, :L, s], hXY[c, :L, N + s] = xy[:, 0], xy[:, 1]
                span[c, s], span[c, N + s] = cur - L + 1, cur
                state[c, s], is_static[c, s], s2m[c, s] = (1 if L == 1 else 0), st, m
        out[k("histXY")], out[k("trackSpan")], out[k("state")], out[k("isStatic")], out[k("slot2map")] = hXY, span, state, is_static, s2m
        M, cov, fl, pf = np.zeros((nP, 3)), np.zeros((nP, 9)), np.zeros(nP, np.uint8), np.zeros((nP, nC), np.int32)
        for p_ in range(nP):
            M[p_], cov[p_] = dbls(3), dbls(9)
            (f_,) = ints(1)
            fl[p_] = f_
            pf[p_] = ints(nC)
        out[k("M")], out[k("cov")], out[k("flags")], out[k("pointFeat")] = M, cov, fl, pf
        nreg, nreg_dyn = (int(v) for v in ints(2))
        out[k("ref_regged")], out[k("ref_regged_dynamic")] = np.int32(nreg), np.int32(nreg_dyn)
        out[k("ref_slot2map")] = ints(nC * N).reshape(nC, N)
        R = dbls(12 * nP).reshape(nP, 12)
        out[k("ref_M")], out[k("ref_cov")] = R[:, :3], R[:, 3:]
        after = ints(nP * (1 + nC)).reshape(nP, 1 + nC)
        out[k("ref_flags")], out[k("ref_pointFeat")] = after[:, 0].astype(np.uint8), after[:, 1:].copy()
    assert o[0] == len(raw)
    return out


def decide_relink_case():
    """the same with MapPoint::pFeatures as time leaves them (oracle/_ref/ref_decide_test golden_relink, CPU): stale heads and chains that
    jump into older tracks, in two bMerge scenes and two plain ones.  Re-laid out the way the device holds them: every dead-track segment
    on a slot of its own behind the frame's N slots (state -1 in this frame, its pixels in the history), featRef [nP][nC][4] = {slot, frame,
    first, seg}, segPool [nC][cap][4] = {slot, last, first, next}, refStatic; deadOwner [nC][Ntot] = the point that owned a dead slot's chain.
    Afterwards per (point, camera): ref_staleOwner (whose stale feature the point holds there, -1 none), ref_preOwner (whose dead-track chain
    hangs directly behind the live feature it holds there, -1 none)."""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_decide_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_decide_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "d.bin")
        subprocess.run([exe, "golden_relink", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = [0]

    def ints(n):
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o[0]).copy()
        o[0] += 4 * n
        return v

    def dbls(n):
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o[0]).copy()
        o[0] += 8 * n
        return v

    out = {}
    (ns,) = ints(1)
    out["n_scenes"] = np.int32(ns)
    for sc in range(ns):
        nC, Hh, N, nP, cur, W, H, with_dyn, with_merge = (int(v) for v in ints(9))
        (pv,) = dbls(1)
        k = lambda n: f"s{sc}_{n}"   # noqa: E731
        out[k("dims")] = np.array([nC, Hh, N, nP, cur, W, H], np.int32)
        out[k("with_dynamic")] = np.int32(with_dyn)   # curDynamicPointsRegInGroup ran behind the static points' registration
        out[k("with_merge")] = np.int32(with_merge)   # bMerge == true: checkUnify at a conflict, the points unified on a yes
        out[k("pixelVar")] = np.float64(pv)
        K, hR, hT = np.zeros((nC, 9)), np.zeros((nC, Hh, 9)), np.zeros((nC, Hh, 3))
        for c in range(nC):
            K[c] = dbls(9)
            for j in range(Hh):
                hR[c, j], hT[c, j] = dbls(9), dbls(3)
        out[k("K")], out[k("histR")], out[k("histT")] = K, hR, hT
        hXY = np.zeros((nC, Hh, 2 * N))
        span = np.full((nC, 2 * N), -1, np.int32)
        state = np.full((nC, N), -1, np.int32)
        is_static = np.ones((nC, N), np.uint8)
        s2m = np.full((nC, N), -1, np.int32)
        for c in range(nC):
            for s in range(N):
                L, st, m = (int(v) for v in ints(3))
                if L == 0:
                    continue
                xy = dbls(2 * L).reshape(L, 2)
                hXY[c, :L, s], hXY[c, :L, N + s] = xy[:, 0], xy[:, 1]
                span[c, s], span[c, N + s] = cur - L + 1, cur
                state[c, s], is_static[c, s], s2m[c, s] = (1 if L == 1 else 0), st, m
        M, cov, fl, pf = np.zeros((nP, 3)), np.zeros((nP, 9)), np.zeros(nP, np.uint8), np.zeros((nP, nC), np.int32)
        chains = {}
        for p_ in range(nP):
            M[p_], cov[p_] = dbls(3), dbls(9)
            (f_,) = ints(1)
            fl[p_] = f_
            pf[p_] = ints(nC)
            for c in range(nC):
                stale, stat, nx = (int(v) for v in ints(3))
                segs = []
                for _ in range(nx):
                    j0, L = (int(v) for v in ints(2))
                    segs.append((j0, L, dbls(2 * L).reshape(L, 2)))
                chains[(p_, c)] = (stale, stat, segs)
        n_dead = [sum(len(chains[(p_, c)][2]) for p_ in range(nP)) for c in range(nC)]
        Nt = N + max(n_dead)
        hXY2 = np.full((nC, Hh, 2 * Nt), np.nan)
        hXY2[:, :, :N], hXY2[:, :, Nt:Nt + N] = hXY[:, :, :N], hXY[:, :, N:]
        span2 = np.full((nC, 2 * Nt), -1, np.int32)
        span2[:, :N], span2[:, Nt:Nt + N] = span[:, :N], span[:, N:]
        state2, stat2, s2m2 = np.full((nC, Nt), -1, np.int32), np.ones((nC, Nt), np.uint8), np.full((nC, Nt), -1, np.int32)
        state2[:, :N], stat2[:, :N], s2m2[:, :N] = state, is_static, s2m
        ref = np.full((nP, nC, 4), -1, np.int32)
        ref[:, :, 1:3] = 0
        pool = np.full((nC, max(max(n_dead), 1), 4), -1, np.int32)
        rstat = np.ones((nP, nC), np.uint8)
        dead_owner = np.full((nC, Nt), -1, np.int32)
        nxt_slot, npool = [N] * nC, [0] * nC
        for p_ in range(nP):
            for c in range(nC):
                stale, stat, segs = chains[(p_, c)]
                placed = []
                for j0, L, xy in segs:
                    s_ = nxt_slot[c]
                    nxt_slot[c] += 1
                    hXY2[c, j0:j0 + L, s_], hXY2[c, j0:j0 + L, Nt + s_] = xy[:, 0], xy[:, 1]
                    span2[c, s_], span2[c, Nt + s_] = cur - (j0 + L - 1), cur - j0
                    stat2[c, s_] = stat
                    dead_owner[c, s_] = p_
                    placed.append((s_, cur - j0, cur - (j0 + L - 1)))
                behind = placed[1:] if stale else placed       # (a stale head's own run is its first segment)
                nxt = -1
                for s_, last, first in reversed(behind):       # the oldest first: each names the one behind it
                    pool[c, npool[c]] = (s_, last, first, nxt)
                    nxt = npool[c]
                    npool[c] += 1
                if stale:
                    ref[p_, c] = (placed[0][0], placed[0][1], placed[0][2], nxt)
                    rstat[p_, c] = stat
                elif pf[p_, c] >= 0:
                    s_ = int(pf[p_, c])
                    ref[p_, c] = (s_, cur, span[c, s_], nxt)
                    rstat[p_, c] = is_static[c, s_]
        out[k("dims")] = np.array([nC, Hh, Nt, nP, cur, W, H], np.int32)
        out[k("n_live_slots")] = np.int32(N)
        out[k("histXY")], out[k("trackSpan")], out[k("state")], out[k("isStatic")], out[k("slot2map")] = hXY2, span2, state2, stat2, s2m2
        out[k("featRef")], out[k("segPool")], out[k("refStatic")], out[k("deadOwner")] = ref, pool, rstat, dead_owner
        out[k("M")], out[k("cov")], out[k("flags")], out[k("pointFeat")] = M, cov, fl, pf
        nreg, nreg_dyn = (int(v) for v in ints(2))
        out[k("ref_regged")], out[k("ref_regged_dynamic")] = np.int32(nreg), np.int32(nreg_dyn)
        rs2m = np.full((nC, Nt), -1, np.int32)
        rs2m[:, :N] = ints(nC * N).reshape(nC, N)
        out[k("ref_slot2map")] = rs2m
        R = dbls(12 * nP).reshape(nP, 12)
        out[k("ref_M")], out[k("ref_cov")] = R[:, :3], R[:, 3:]
        rfl, rpf = np.zeros(nP, np.uint8), np.zeros((nP, nC), np.int32)
        r_stale, r_pre = np.full((nP, nC), -1, np.int32), np.full((nP, nC), -1, np.int32)
        for p_ in range(nP):
            after = ints(1 + nC)
            rfl[p_], rpf[p_] = after[0], after[1:]
            so = ints(2 * nC).reshape(nC, 2)
            r_stale[p_], r_pre[p_] = so[:, 0], so[:, 1]
        out[k("ref_flags")], out[k("ref_pointFeat")], out[k("ref_staleOwner")], out[k("ref_preOwner")] = rfl, rpf, r_stale, r_pre
    assert o[0] == len(raw)
    return out


def mergability_case():
    """the reference's own CoSLAM::staticCheckMergability (oracle/_ref/ref_mergability_test golden, CPU): 150 tracks of 1..24
    frames, newest first, and its verdicts."""
    import struct
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_mergability_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_mergability_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.bin")
        subprocess.run([exe, "golden", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    (n,) = struct.unpack_from("i", raw, 0)
    o = 4
    Ls, sig, K, M, cov, rec, verdict = [], [], [], [], [], [], []
    for _ in range(n):
        (L,) = struct.unpack_from("i", raw, o)
        o += 4
        v = np.frombuffer(raw, dtype=np.float64, count=1 + 9 + 3 + 9 + 14 * L, offset=o).copy()
        o += v.nbytes
        (vd,) = struct.unpack_from("i", raw, o)
        o += 4
        Ls.append(L), sig.append(v[0]), K.append(v[1:10]), M.append(v[10:13]), cov.append(v[13:22]), rec.append(v[22:]), verdict.append(vd)
    assert o == len(raw)
    return dict(L=np.array(Ls, np.int32), sigma=np.array(sig), K=np.array(K), M=np.array(M), cov=np.array(cov),
                rec=np.concatenate(rec), rec_ptr=np.concatenate([[0], np.cumsum(14 * np.array(Ls))]).astype(np.int64),
                verdict=np.array(verdict, np.int32))


def mergability_long_case():
    """the reference's own CoSLAM::staticCheckMergability on LONG tracks (oracle/_ref/ref_mergability_test golden_long, CPU): 3 cameras
    x 420 frames of poses, 48 tracks per camera (most 200-420 frames long, a few inside the 64-frame window), each with its own map
    point, laid out frame by frame the way the device's history is fed; pixels of frames before a track's first are NaN."""
    import struct
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_mergability_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_mergability_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.bin")
        subprocess.run([exe, "golden_long", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    nC, T, nT = struct.unpack_from("iii", raw, 0)
    (sigma,) = struct.unpack_from("d", raw, 12)
    o = 20
    K, R, t = np.zeros((nC, 9)), np.zeros((nC, T, 9)), np.zeros((nC, T, 3))
    for c in range(nC):
        v = np.frombuffer(raw, dtype=np.float64, count=9 + 12 * T, offset=o)
        o += v.nbytes
        K[c] = v[:9]
        R[c], t[c] = v[9:].reshape(T, 12)[:, :9], v[9:].reshape(T, 12)[:, 9:]
    M, cov = np.zeros((nC, nT, 3)), np.zeros((nC, nT, 9))
    f1, bad, verdict = np.zeros((nC, nT), np.int32), np.zeros((nC, nT), np.int32), np.zeros((nC, nT), np.int32)
    xy = np.full((nC, nT, T, 2), np.nan)
    for c in range(nC):
        for k in range(nT):
            v = np.frombuffer(raw, dtype=np.float64, count=12, offset=o)
            o += 96
            M[c, k], cov[c, k] = v[:3], v[3:]
            (f1[c, k],) = struct.unpack_from("i", raw, o)
            o += 4
            n = T - f1[c, k]
            xy[c, k, f1[c, k]:] = np.frombuffer(raw, dtype=np.float64, count=2 * n, offset=o).reshape(n, 2)
            o += 16 * n
            bad[c, k], verdict[c, k] = struct.unpack_from("ii", raw, o)
            o += 8
    assert o == len(raw)
    return dict(sigma=np.float64(sigma), K=K, R=R, t=t, M=M, cov=cov, f1=f1, bad_frame=bad, verdict=verdict, xy=xy.astype(np.float64))


def update_points_case():
    """the reference's own RobustBundleRTS::updateNewPosesPoints over updateStaticPointPosition / updateDynamicPointPosition
    (oracle/_ref/ref_update_points_test golden, CPU): 6 scenes of 60 map points, re-laid out the way the device holds them (slot =
    point index, ring entry 0 = the current frame), and the reference's points / covariances afterwards."""
    import struct
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_update_points_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_update_points_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "u.bin")
        subprocess.run([exe, "golden", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = 0

    def ints(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o).copy()
        o += 4 * n
        return v

    def dbls(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o).copy()
        o += 8 * n
        return v

    out = {}
    (nS,) = ints(1)
    out["n_scenes"] = np.int32(nS)
    for sc in range(nS):
        nC, H, nP, firstKey, cur = ints(5)
        (sigma,) = dbls(1)
        K, iK = np.zeros((nC, 9)), np.zeros((nC, 9))
        for c in range(nC):
            K[c], iK[c] = dbls(9), dbls(9)
        hR, hT = np.zeros((nC, H, 9)), np.zeros((nC, H, 3))
        for c in range(nC):
            for j in range(H):
                hR[c, j], hT[c, j] = dbls(9), dbls(3)
        N = nP
        hXY = np.zeros((nC, H, 2 * N))
        span = np.full((nC, 2 * N), -1, np.int32)
        fstat = np.ones((nC, N), np.uint8)
        pf = np.full((nP, nC), -1, np.int32)
        M0, cov0 = np.zeros((nP, 3)), np.zeros((nP, 9))
        flags, lastF, isCur = np.zeros(nP, np.uint8), np.zeros(nP, np.int32), np.zeros(nP, np.uint8)
        for p in range(nP):
            M0[p], cov0[p] = dbls(3), dbls(9)
            ltype, unc, lastF[p], isCur[p] = ints(4)
            flags[p] = (1 if ltype == 1 else 0) | (2 if ltype == -2 else 0) | (4 if unc else 0)
            for c in range(nC):
                L, dyn = ints(2)
                m = dbls(2 * L).reshape(L, 2)
                if L:
                    pf[p, c] = p
                    span[c, p], span[c, N + p] = cur - L + 1, cur
                    fstat[c, p] = 0 if dyn else 1
                    hXY[c, :L, p], hXY[c, :L, N + p] = m[:, 0], m[:, 1]
        Mr, covr = dbls(3 * nP).reshape(nP, 3), np.zeros((nP, 9))
        # (the driver writes M, cov per point interleaved)
        o -= 8 * 3 * nP
        for p in range(nP):
            Mr[p], covr[p] = dbls(3), dbls(9)
        rsel, Mf, covf = np.zeros(nP, np.uint8), np.zeros((nP, 3)), np.zeros((nP, 9))
        for p in range(nP):
            (rs_,) = ints(1)
            rsel[p], Mf[p], covf[p] = rs_, dbls(3), dbls(9)
        (nU,) = ints(1)
        up = np.zeros((nU, 2), np.int32)
        uh1, uh2 = np.zeros((nU, nC), np.uint8), np.zeros((nU, nC), np.uint8)
        uM1, uM2, uM, ucov, uok = np.zeros((nU, 3)), np.zeros((nU, 3)), np.zeros((nU, 3)), np.zeros((nU, 9)), np.zeros(nU, np.uint8)
        for q in range(nU):
            up[q] = ints(2)
            hh = ints(2 * nC).reshape(nC, 2)
            uh1[q], uh2[q] = hh[:, 0], hh[:, 1]
            uM1[q], uM2[q] = dbls(3), dbls(3)
            (uok[q],) = ints(1)
            uM[q], ucov[q] = dbls(3), dbls(9)
        pre = f"s{sc}_"
        for k, v in dict(unify_pts=up, unify_has1=uh1, unify_has2=uh2, unify_M1=uM1, unify_M2=uM2, unify_ok=uok, unify_M=uM,
                         unify_cov=ucov).items():
            out[pre + k] = v
        for k, v in dict(K=K, iK=iK, histR=hR, histT=hT, histXY=hXY, trackSpan=span, featStatic=fstat, pointFeat=pf, M0=M0, cov0=cov0,
                         flags=flags, lastFrame=lastF, isCurrent=isCur, firstKey=np.int32(firstKey), curFrame=np.int32(cur),
                         sigma=np.float64(sigma), M_ref=Mr, cov_ref=covr, refine_select=rsel, M_refine=Mf, cov_refine=covf).items():
            out[pre + k] = v
    assert o == len(raw)
    return out


def update_points_relink_case():
    """re-linked and stale feature chains (oracle/_ref/ref_update_points_test golden_relink, CPU): the reference's own
    updateNewPosesPoints / refineMapPoint / checkUnify over chains as `pFeat->preFrame = p->pFeatures[iCam]` (SL_CoSLAM.cpp:775-779) and a
    lost track leave them, re-laid out the way the device holds them: every segment of consecutive frames on a slot of its own (slot =
    running number per camera), featRef [nP][nC][4] = {slot, frame, first, seg}, segPool [nC][cap][4] = {slot, last, first, next}."""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_update_points_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_update_points_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "u.bin")
        subprocess.run([exe, "golden_relink", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = 0

    def ints(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o).copy()
        o += 4 * n
        return v

    def dbls(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o).copy()
        o += 8 * n
        return v

    out = {}
    (nS,) = ints(1)
    out["n_scenes"] = np.int32(nS)
    for sc in range(nS):
        nC, H, nP, firstKey, cur = ints(5)
        (sigma,) = dbls(1)
        K, iK = np.zeros((nC, 9)), np.zeros((nC, 9))
        for c in range(nC):
            K[c], iK[c] = dbls(9), dbls(9)
        hR, hT = np.zeros((nC, H, 9)), np.zeros((nC, H, 3))
        for c in range(nC):
            for j in range(H):
                hR[c, j], hT[c, j] = dbls(9), dbls(3)
        N = 3 * nP                                           # at most three segments per (point, camera), each on its own slot
        hXY = np.full((nC, H, 2 * N), np.nan)                # (a pixel nobody should read)
        fstat = np.ones((nC, N), np.uint8)
        ref = np.full((nP, nC, 4), -1, np.int32)
        pool = np.full((nC, 2 * nP, 4), -1, np.int32)
        nslot, npool = [0] * nC, [0] * nC
        M0, cov0 = np.zeros((nP, 3)), np.zeros((nP, 9))
        flags, lastF, isCur = np.zeros(nP, np.uint8), np.zeros(nP, np.int32), np.zeros(nP, np.uint8)
        for p in range(nP):
            M0[p], cov0[p] = dbls(3), dbls(9)
            ltype, unc, lastF[p], isCur[p] = ints(4)
            flags[p] = (1 if ltype == 1 else 0) | (2 if ltype == -2 else 0) | (4 if unc else 0)
            for c in range(nC):
                nSeg, dyn = ints(2)
                segs = []
                for q in range(nSeg):
                    j0, L = ints(2)
                    m = dbls(2 * L).reshape(L, 2)
                    s_ = nslot[c]
                    nslot[c] += 1
                    hXY[c, j0:j0 + L, s_], hXY[c, j0:j0 + L, N + s_] = m[:, 0], m[:, 1]
                    segs.append((s_, cur - j0, cur - (j0 + L - 1)))
                    if q == 0:
                        fstat[c, s_] = 0 if dyn else 1
                nxt = -1
                for s_, last, first in reversed(segs[1:]):   # the oldest segment first: each names the one behind it
                    pool[c, npool[c]] = (s_, last, first, nxt)
                    nxt = npool[c]
                    npool[c] += 1
                if segs:
                    ref[p, c] = (segs[0][0], segs[0][1], segs[0][2], nxt)
        Mr, covr = np.zeros((nP, 3)), np.zeros((nP, 9))
        for p in range(nP):
            Mr[p], covr[p] = dbls(3), dbls(9)
        rsel, Mf, covf = np.zeros(nP, np.uint8), np.zeros((nP, 3)), np.zeros((nP, 9))
        for p in range(nP):
            (rs_,) = ints(1)
            rsel[p], Mf[p], covf[p] = rs_, dbls(3), dbls(9)
        (nU,) = ints(1)
        up = np.zeros((nU, 2), np.int32)
        uh1, uh2 = np.zeros((nU, nC), np.uint8), np.zeros((nU, nC), np.uint8)
        uM1, uM2, uM, ucov, uok = np.zeros((nU, 3)), np.zeros((nU, 3)), np.zeros((nU, 3)), np.zeros((nU, 9)), np.zeros(nU, np.uint8)
        for q in range(nU):
            up[q] = ints(2)
            hh = ints(2 * nC).reshape(nC, 2)
            uh1[q], uh2[q] = hh[:, 0], hh[:, 1]
            uM1[q], uM2[q] = dbls(3), dbls(3)
            (uok[q],) = ints(1)
            uM[q], ucov[q] = dbls(3), dbls(9)
        pre = f"s{sc}_"
        for k, v in dict(unify_pts=up, unify_has1=uh1, unify_has2=uh2, unify_M1=uM1, unify_M2=uM2, unify_ok=uok, unify_M=uM,
                         unify_cov=ucov).items():
            out[pre + k] = v
        for k, v in dict(K=K, iK=iK, histR=hR, histT=hT, histXY=hXY, featStatic=fstat, featRef=ref, segPool=pool, M0=M0, cov0=cov0,
                         flags=flags, lastFrame=lastF, isCurrent=isCur, firstKey=np.int32(firstKey), curFrame=np.int32(cur),
                         sigma=np.float64(sigma), M_ref=Mr, cov_ref=covr, refine_select=rsel, M_refine=Mf, cov_refine=covf).items():
            out[pre + k] = v
    assert o == len(raw)
    return out


def keyframe_case():
    """the reference's own key-frame decision (oracle/_ref/ref_keyframe_test golden, CPU): CoSLAM::IsReadyForKeyFrame with its helpers,
    compiled in place, on 6 scenes of 3-6 cameras x 400 slots; the device's tables (state, slot2map) and the reference's answers"""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_keyframe_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_keyframe_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "k.bin")
        subprocess.run([exe, "golden", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = 0

    def ints(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o).copy()
        o += 4 * n
        return v

    def dbls(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o).copy()
        o += 8 * n
        return v

    out = {}
    (nS,) = ints(1)
    out["n_scenes"] = np.int32(nS)
    for sc in range(nS):
        nC, N, nMap, cur = ints(4)
        ratio, ang, trans = dbls(3)
        M, flags, first = np.zeros((nMap, 3)), np.zeros(nMap, np.uint8), np.zeros(nMap, np.int32)
        for p in range(nMap):
            M[p] = dbls(3)
            ltype, unc, first[p] = ints(3)
            flags[p] = (1 if ltype == 1 else 0) | (2 if ltype == -2 else 0) | (4 if unc else 0)
        R, t, sR, sT = np.zeros((nC, 9)), np.zeros((nC, 3)), np.zeros((nC, 9)), np.zeros((nC, 3))
        kf, km = np.zeros(nC, np.int32), np.zeros(nC, np.int32)
        state, s2m = np.full((nC, N), -1, np.int32), np.full((nC, N), -1, np.int32)
        for c in range(nC):
            R[c], t[c], sR[c], sT[c] = dbls(9), dbls(3), dbls(9), dbls(3)
            kf[c], km[c] = ints(2)
            mo = ints(N)
            state[c] = np.where(mo == -2, -1, 0)
            s2m[c] = np.where(mo >= 0, mo, -1)
        ready, nstat, cen = np.zeros(nC, np.int32), np.zeros(nC, np.int32), np.zeros((nC, 3))
        for c in range(nC):
            ready[c], nstat[c] = ints(2)
            cen[c] = dbls(3)
        pre = f"s{sc}_"
        for k, v in dict(mapPts=M, mapFlags=flags, firstFrame=first, R=R, t=t, selfR=sR, selfT=sT, keyFrame=kf, keyMapped=km, state=state,
                         slot2map=s2m, curFrame=np.int32(cur), ratio=np.float64(ratio), minViewAngle=np.float64(ang),
                         minTranslation=np.float64(trans), ready_ref=ready, nMappedStatic_ref=nstat, center_ref=cen).items():
            out[pre + k] = v
    assert o == len(raw)
    return out


def intracam_newpts_case():
    """the reference's own SingleSLAM::newMapPoints (oracle/_ref/ref_intracam_newpts_test golden, CPU): three one-camera scenes of 160 slots;
    the device's tables (history, state, slot2map, trackSpan, isStatic) and the reference's new points in slot order"""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_intracam_newpts_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_intracam_newpts_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "n.bin")
        subprocess.run([exe, "golden", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = 0

    def ints(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o).copy()
        o += 4 * n
        return v

    def dbls(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o).copy()
        o += 8 * n
        return v

    out = {}
    (nS,) = ints(1)
    out["n_scenes"] = np.int32(nS)
    for sc in range(nS):
        H, N, cur = ints(3)
        K, iK = dbls(9), dbls(9)
        sigma, max_epi = dbls(2)
        (min_len,) = ints(1)
        hR, hT = np.zeros((H, 9)), np.zeros((H, 3))
        for j in range(H):
            hR[j], hT[j] = dbls(9), dbls(3)
        hXY = np.zeros((H, 2 * N))
        state, s2m = np.full(N, -1, np.int32), np.full(N, -1, np.int32)
        span, fstat = np.full(2 * N, -1, np.int32), np.ones(N, np.uint8)
        for k in range(N):
            L, mapped, dyn = ints(3)
            m = dbls(2 * L).reshape(L, 2)
            if L:
                state[k] = 0
                s2m[k] = 7 if mapped else -1
                fstat[k] = 0 if dyn else 1
                span[k], span[N + k] = cur - L + 1, cur
                hXY[:L, k], hXY[:L, N + k] = m[:, 0], m[:, 1]
        (n_new,) = ints(1)
        slot, first, M, cov = np.zeros(n_new, np.int32), np.zeros(n_new, np.int32), np.zeros((n_new, 3)), np.zeros((n_new, 9))
        for q in range(n_new):
            slot[q], first[q] = ints(2)
            M[q], cov[q] = dbls(3), dbls(9)
        pre = f"s{sc}_"
        for k, v in dict(K=K, iK=iK, sigma=np.float64(sigma), maxEpiErr=np.float64(max_epi), minTrackLen=np.int32(min_len), curFrame=np.int32(cur),
                         histR=hR, histT=hT, histXY=hXY, state=state, slot2map=s2m, trackSpan=span, isStatic=fstat, new_slot=slot,
                         new_first=first, new_M=M, new_cov=cov).items():
            out[pre + k] = v
    assert o == len(raw)
    return out


def classify_case():
    """the reference's own CoSLAM::mapPointsClassify over isStaticPoint / isStaticPointExclude / isDynamicPoint / isLittleMove /
    isStaticRemovable (oracle/_ref/ref_classify_test golden, CPU): 3 scenes of 72 map points walking every branch, re-laid out the
    way the device holds them (slot = point index, ring entry 0 = the current frame), and the points' states afterwards."""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_classify_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_classify_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "c.bin")
        subprocess.run([exe, "golden", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = 0

    def ints(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o).copy()
        o += 4 * n
        return v

    def dbls(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o).copy()
        o += 8 * n
        return v

    def flags_of(ltype, unc):
        return (1 if ltype == 1 else 0) | (2 if ltype == -2 else 0) | (4 if unc else 0)

    out = {}
    (nS,) = ints(1)
    out["n_scenes"] = np.int32(nS)
    for sc in range(nS):
        nC, H, nP, cur = ints(4)
        (pixelVar,) = dbls(1)
        K, iK = np.zeros((nC, 9)), np.zeros((nC, 9))
        for c in range(nC):
            K[c], iK[c] = dbls(9), dbls(9)
        hR, hT = np.zeros((nC, H, 9)), np.zeros((nC, H, 3))
        for c in range(nC):
            for j in range(H):
                hR[c, j], hT[c, j] = dbls(9), dbls(3)
        N = nP
        hXY = np.zeros((nC, H, 2 * N), np.float32).astype(np.float64)
        span = np.full((nC, 2 * N), -1, np.int32)
        fstat = np.ones((nC, N), np.uint8)
        pf = np.full((nP, nC), -1, np.int32)
        fframe = np.full((nP, nC), -1, np.int32)
        ffirst = np.full((nP, nC), -1, np.int32)
        M0, cov0 = np.zeros((nP, 3)), np.zeros((nP, 9))
        flags, newpt, sfn, first = np.zeros(nP, np.uint8), np.zeros(nP, np.uint8), np.zeros(nP, np.int32), np.zeros(nP, np.int32)
        for p in range(nP):
            M0[p], cov0[p] = dbls(3), dbls(9)
            ltype, unc, newpt[p], sfn[p], first[p] = ints(5)
            flags[p] = flags_of(ltype, unc)
            for c in range(nC):
                L, back, dyn = ints(3)
                m = dbls(2 * L).reshape(L, 2)
                if L:
                    pf[p, c], fframe[p, c], ffirst[p, c] = p, cur - back, cur - back - L + 1
                    span[c, p], span[c, N + p] = cur - back - L + 1, cur - back
                    fstat[c, p] = 0 if dyn else 1
                    hXY[c, back:back + L, p], hXY[c, back:back + L, N + p] = m[:, 0], m[:, 1]
        Mr, covr = np.zeros((nP, 3)), np.zeros((nP, 9))
        flr, newr, sfr = np.zeros(nP, np.uint8), np.zeros(nP, np.uint8), np.zeros(nP, np.int32)
        has_r, fstat_r = np.zeros((nP, nC), np.uint8), fstat.copy()
        for p in range(nP):
            Mr[p], covr[p] = dbls(3), dbls(9)
            ltype, unc, newr[p], sfr[p] = ints(4)
            flr[p] = flags_of(ltype, unc)
            for c in range(nC):
                has_r[p, c], dyn = ints(2)
                if has_r[p, c] and fframe[p, c] == cur:
                    fstat_r[c, p] = 0 if dyn else 1
        pre = f"s{sc}_"
        for k, v in dict(K=K, iK=iK, histR=hR, histT=hT, histXY=hXY, trackSpan=span, featStatic=fstat, pointFeat=pf, featFrame=fframe,
                         featFirst=ffirst, M0=M0, cov0=cov0, flags=flags, newPt=newpt, staticFrameNum=sfn, firstFrame=first,
                         curFrame=np.int32(cur), pixelVar=np.float64(pixelVar), M_ref=Mr, cov_ref=covr, flags_ref=flr, newPt_ref=newr,
                         staticFrameNum_ref=sfr, hasFeature_ref=has_r, featStatic_ref=fstat_r).items():
            out[pre + k] = v
    assert o == len(raw)
    return out


def classify_relink_case():
    """the reference's own mapPointsClassify over re-linked and stale feature chains (oracle/_ref/ref_classify_test golden_relink, CPU),
    re-laid out the way the device holds them: every segment of consecutive frames on a slot of its own (running number per camera),
    featRef [nP][nC][4] = {slot, frame, first, seg}, segPool [nC][cap][4] = {slot, last, first, next}; pointFeat names the heads that are
    of this frame.  featDyn_ref [nP][nC]: the type the reference left on the feature each pointer names (live or stale)."""
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "ref_classify_test")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/ref_classify_test missing: run `make -C oracle` where /root/reference exists")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "c.bin")
        subprocess.run([exe, "golden_relink", path], check=True, stdout=subprocess.DEVNULL)
        raw = open(path, "rb").read()
    o = 0

    def ints(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.int32, count=n, offset=o).copy()
        o += 4 * n
        return v

    def dbls(n):
        nonlocal o
        v = np.frombuffer(raw, dtype=np.float64, count=n, offset=o).copy()
        o += 8 * n
        return v

    def flags_of(ltype, unc):
        return (1 if ltype == 1 else 0) | (2 if ltype == -2 else 0) | (4 if unc else 0)

    out = {}
    (nS,) = ints(1)
    out["n_scenes"] = np.int32(nS)
    for sc in range(nS):
        nC, H, nP, cur = ints(4)
        (pixelVar,) = dbls(1)
        K, iK = np.zeros((nC, 9)), np.zeros((nC, 9))
        for c in range(nC):
            K[c], iK[c] = dbls(9), dbls(9)
        hR, hT = np.zeros((nC, H, 9)), np.zeros((nC, H, 3))
        for c in range(nC):
            for j in range(H):
                hR[c, j], hT[c, j] = dbls(9), dbls(3)
        N = 3 * nP                                           # at most three segments per (point, camera), each on its own slot
        hXY = np.full((nC, H, 2 * N), np.nan)                # (a pixel nobody should read)
        span = np.full((nC, 2 * N), -1, np.int32)
        fstat = np.ones((nC, N), np.uint8)
        rstat = np.ones((nP, nC), np.uint8)
        pf = np.full((nP, nC), -1, np.int32)
        s2m = np.full((nC, N), -1, np.int32)
        ref = np.full((nP, nC, 4), -1, np.int32)
        ref[:, :, 1:3] = 0
        pool = np.full((nC, 2 * nP, 4), -1, np.int32)
        nslot, npool = [0] * nC, [0] * nC
        M0, cov0 = np.zeros((nP, 3)), np.zeros((nP, 9))
        flags, newpt, sfn, first = np.zeros(nP, np.uint8), np.zeros(nP, np.uint8), np.zeros(nP, np.int32), np.zeros(nP, np.int32)
        for p in range(nP):
            M0[p], cov0[p] = dbls(3), dbls(9)
            ltype, unc, newpt[p], sfn[p], first[p] = ints(5)
            flags[p] = flags_of(ltype, unc)
            for c in range(nC):
                nSeg, dyn = ints(2)
                segs = []
                for q in range(nSeg):
                    j0, L = ints(2)
                    m = dbls(2 * L).reshape(L, 2)
                    s_ = nslot[c]
                    nslot[c] += 1
                    hXY[c, j0:j0 + L, s_], hXY[c, j0:j0 + L, N + s_] = m[:, 0], m[:, 1]
                    segs.append((s_, cur - j0, cur - (j0 + L - 1)))
                    span[c, s_], span[c, N + s_] = cur - (j0 + L - 1), cur - j0
                nxt = -1
                for s_, last, first_ in reversed(segs[1:]):   # the oldest segment first: each names the one behind it
                    pool[c, npool[c]] = (s_, last, first_, nxt)
                    nxt = npool[c]
                    npool[c] += 1
                if segs:
                    ref[p, c] = (segs[0][0], segs[0][1], segs[0][2], nxt)
                    rstat[p, c] = 0 if dyn else 1
                    if segs[0][1] == cur:
                        pf[p, c] = segs[0][0]
                        s2m[c, segs[0][0]] = p
                        fstat[c, segs[0][0]] = 0 if dyn else 1
        Mr, covr = np.zeros((nP, 3)), np.zeros((nP, 9))
        flr, newr, sfr = np.zeros(nP, np.uint8), np.zeros(nP, np.uint8), np.zeros(nP, np.int32)
        has_r, dyn_r = np.zeros((nP, nC), np.uint8), np.zeros((nP, nC), np.uint8)
        for p in range(nP):
            Mr[p], covr[p] = dbls(3), dbls(9)
            ltype, unc, newr[p], sfr[p] = ints(4)
            flr[p] = flags_of(ltype, unc)
            for c in range(nC):
                has_r[p, c], dyn_r[p, c] = ints(2)
        pre = f"s{sc}_"
        for k, v in dict(K=K, iK=iK, histR=hR, histT=hT, histXY=hXY, trackSpan=span, featStatic=fstat, refStatic=rstat, pointFeat=pf,
                         slot2map=s2m, featRef=ref, segPool=pool, M0=M0, cov0=cov0, flags=flags, newPt=newpt, staticFrameNum=sfn,
                         firstFrame=first, curFrame=np.int32(cur), pixelVar=np.float64(pixelVar), M_ref=Mr, cov_ref=covr, flags_ref=flr,
                         newPt_ref=newr, staticFrameNum_ref=sfr, hasFeature_ref=has_r, featDyn_ref=dyn_r).items():
            out[pre + k] = v
    assert o == len(raw)
    return out


def cgklt_cases():
    """see the module docstring.  Every array named *_cg comes out of libcgklt_ref.so."""
    from oracle import cgref
    if not cgref.have():
        raise SystemExit("oracle/_ref/libcgklt_ref.so missing: run oracle/build_cgref.sh where /root/reference exists")
    out = {}
    scenes = [  # W, H, L, fw, fh, window, iterations, levelSkip, minDistance, minCornerness, scene seed
        (160, 120, 3, 10, 8, 7, 6, 1, 5, 800.0, 9),
        (256, 192, 4, 16, 8, 5, 8, 1, 7, 1500.0, 13),   # 2:1 slot grid: the gain shader's neighbour taps land on texel edges
    ]
    out["scenes"] = np.array([s[:9] for s in scenes], dtype=np.int64)
    out["minCornerness"] = np.array([s[9] for s in scenes], dtype=np.float32)
    for si, (W, H, L, fw, fh, win, iters, skip, mind, minc, seed) in enumerate(scenes):
        sc = Scene(1, W, H, int(W * H / 75), seed=seed, sigma=1.4)
        imgs = np.stack([sc.render(0, f) for f in range(3)])
        out[f"s{si}_images"] = imgs
        pyr = [cgref.pyramid_build(imgs[f], W, H, L, 0) for f in range(3)]
        import hashlib
        sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
        for f in range(3):
            # the texels in full for scene 0 and the first frame of scene 1, SHA-256 of the same bytes for the rest (fixture size)
            out[f"s{si}_pyr{f}_cg" if (si == 0 or f == 0) else f"s{si}_pyr{f}_sha_cg"] = pyr[f] if (si == 0 or f == 0) else sha(pyr[f])
            assert np.array_equal(pyr[f], oracle.pyramid_build(imgs[f], W, H, L, 0)), "oracle pyramid != shader pyramid"
        # the other reading of the decimation taps (NEAREST exactly on a texel edge resolves downwards); even sizes only
        cen = cgref.pyramid_build(imgs[0], W, H, L, 1)
        out[f"s{si}_pyr0_centered_cg" if si == 0 else f"s{si}_pyr0_centered_sha_cg"] = cen if si == 0 else sha(cen)
        assert np.array_equal(cen, oracle.pyramid_build(imgs[0], W, H, L, 1))
        hw = win // 2
        for gain in (0, 1):
            pre = f"s{si}_g{gain}_"
            cfg = KLT_SequenceTrackerConfig(nIterations=iters, nLevels=L, levelSkip=skip, windowWidth=win, trackWithGain=gain,
                                            minCornerness=minc, convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=mind)
            o = oracle.SequenceTracker(cfg, sum_mode=0)   # the shader's serial summation order
            o.allocate(W, H, L, fw, fh)
            plw, cap = 2 * fw, 4 * fw * fh
            margin_t, margin_d = cfg.trackBorderMargin, 10.0   # v3d_gpuklt.h:114, KLT_SequenceTracker::allocate
            for f in range(3):
                if f:
                    o.advanceFrame()
                    mbuf = np.full(fw * fh, 1e30, np.float32)
                    oracle.set_threshold_margin_buffer(mbuf)
                    try:
                        n, d = o.redetect(imgs[f])
                    finally:
                        oracle.set_threshold_margin_buffer(None)
                    # --- the tracker, by the shaders
                    if gain:
                        trk = cgref.track_gain(pyr[f - 1], pyr[f], W, H, L, skip, hw, iters, fw, fh, margin_t, 1.0, 20000.0, prov, prov)
                    else:
                        trk = cgref.track_nogain(pyr[f - 1], pyr[f], W, H, L, skip, hw, fw, fh, margin_t, 1.0, 20000.0, prov)
                    alive = trk[:, 0] >= 0
                    assert np.array_equal(alive, d["status"] == 0), "tracked set: oracle != shaders"
                    assert np.array_equal(trk[alive, :2], d["pos"][alive]) and np.array_equal(trk[alive, 2], d["gain"][alive])
                    out[pre + f"tracked{f}_cg"] = trk
                    out[pre + f"margin{f}"] = mbuf
                    present = np.where(alive[:, None], np.concatenate([trk[:, :2], np.zeros((fw * fh, 1), np.float32)], 1),
                                       np.array([-1, -1, 0], np.float32)).astype(np.float32)
                else:
                    n, d = o.detect(imgs[0])
                    present = None
                # --- the detector, by the shaders
                lvl0 = oracle.level_view(pyr[f], W, H, L, 0)
                c = cgref.cornerness(lvl0, W, H, minc, margin_d)
                if f == 0 and gain == 0:
                    out[f"s{si}_cornerness0_cg"] = c
                    assert np.array_equal(c.view(np.uint32), oracle.cornerness(lvl0, W, H, minc, margin_d).view(np.uint32))
                if present is not None:
                    c = cgref.suppress_present(c, present)
                c = cgref.nonmax(c, mind)
                assert np.array_equal(c.view(np.uint32), o.read_cornerness().view(np.uint32)), "non-max map: oracle != shaders"
                cnt, lst = cgref.extract(c, plw, cap)
                cnt_o, lst_o = oracle.extract(c, cap)
                assert cnt == cnt_o and np.array_equal(lst.view(np.uint32), lst_o.view(np.uint32)), "point list: oracle != shaders"
                new = d["status"] == 1
                assert set(map(tuple, d["pos"][new].tolist())) <= set(map(tuple, lst[:, :2].tolist()))
                out[pre + f"nonmax{f}_cg"], out[pre + f"list{f}_cg"] = c, lst
                out[pre + f"n{f}"] = np.array([n])
                out[pre + f"status{f}"], out[pre + f"pos{f}"], out[pre + f"gain{f}"] = d["status"], d["pos"], d["gain"]
                prov = o.read_features()
                out[pre + f"provided{f}"] = prov
            o.close()
    # pass-level vectors: one launch of klt_tracker_with_gain.cg per level on a list with dead slots and real thresholds
    si, (W, H, L, fw, fh, win) = 0, scenes[0][:6]
    rng = np.random.default_rng(31)
    N = fw * fh
    feat0 = out["s0_g1_provided0"].copy()
    cur = feat0.copy()
    cur[:, :2] += (rng.uniform(-0.4, 0.4, (N, 2)) / np.array([W, H])).astype(np.float32)
    cur[:, 2] = rng.uniform(0.9, 1.1, N).astype(np.float32)
    cur[feat0[:, 0] < 0] = -1.0
    cur[rng.random(N) < 0.05] = -1.0
    vr = np.array([4.0 / W, 4.0 / H, 1 - 4.0 / W, 1 - 4.0 / H], np.float32)
    out["pass_feat0"], out["pass_cur"], out["pass_vr"] = feat0, cur, vr
    p0, p1 = out["s0_pyr0_cg"], out["s0_pyr1_cg"]
    for level in range(L):
        r = cgref.track_gain_pass(p0, p1, W, H, L, level, fw, fh, win // 2, feat0, cur, 4.0, 20000.0, vr, 1.0, 200.0)
        assert np.array_equal(r.view(np.uint32),
                              cgref.okl_track_gain_pass(p0, p1, W, H, L, level, fw, fh, win // 2, feat0, cur, 4.0, 20000.0, vr).view(np.uint32))
        out[f"pass_level{level}_cg"] = r
    return out


if __name__ == "__main__":
    if not oracle.have_ref():
        raise SystemExit("oracle/_ref/libintracam_ref.so missing: run `make -C oracle` where /root/reference exists")
    which = sys.argv[1:] or ["pose", "klt", "ba", "register", "ncc", "posegraph", "export", "mergability", "mergability_long", "update_points", "update_points_relink", "keyframe", "intracam_newpts", "classify", "classify_relink", "intercam", "decide_relink", "newpts", "decide", "cgklt"]
    if "pose" in which:
        np.savez_compressed(os.path.join(HERE, "pose_golden.npz"), **pose_cases())
    if "klt" in which:
        np.savez_compressed(os.path.join(HERE, "klt_golden.npz"), **klt_cases())
    if "ba" in which:
        np.savez_compressed(os.path.join(HERE, "ba_golden.npz"), **ba_case())
    if "register" in which:
        np.savez_compressed(os.path.join(HERE, "register_golden.npz"), **register_case())
    if "ncc" in which:
        np.savez_compressed(os.path.join(HERE, "ncc_golden.npz"), **ncc_case())
    if "posegraph" in which:
        np.savez_compressed(os.path.join(HERE, "posegraph_golden.npz"), **posegraph_case())
    if "export" in which:
        np.savez_compressed(os.path.join(HERE, "export_golden.npz"), **export_case())
    if "intercam" in which:
        np.savez_compressed(os.path.join(HERE, "intercam_golden.npz"), **intercam_case())
    if "mergability" in which:
        np.savez_compressed(os.path.join(HERE, "mergability_golden.npz"), **mergability_case())
    if "mergability_long" in which:
        np.savez_compressed(os.path.join(HERE, "mergability_long_golden.npz"), **mergability_long_case())
    if "update_points" in which:
        np.savez_compressed(os.path.join(HERE, "update_points_golden.npz"), **update_points_case())
    if "intracam_newpts" in which:
        np.savez_compressed(os.path.join(HERE, "intracam_newpts_golden.npz"), **intracam_newpts_case())
    if "keyframe" in which:
        np.savez_compressed(os.path.join(HERE, "keyframe_golden.npz"), **keyframe_case())
    if "classify_relink" in which:
        np.savez_compressed(os.path.join(HERE, "classify_relink_golden.npz"), **classify_relink_case())
    if "decide_relink" in which:
        np.savez_compressed(os.path.join(HERE, "decide_relink_golden.npz"), **decide_relink_case())
    if "update_points_relink" in which:
        np.savez_compressed(os.path.join(HERE, "update_points_relink_golden.npz"), **update_points_relink_case())
    if "classify" in which:
        np.savez_compressed(os.path.join(HERE, "classify_golden.npz"), **classify_case())
    if "decide" in which:
        np.savez_compressed(os.path.join(HERE, "decide_golden.npz"), **decide_case())
    if "newpts" in which:
        np.savez_compressed(os.path.join(HERE, "newpts_golden.npz"), **newpts_case())
    if "cgklt" in which:
        np.savez_compressed(os.path.join(HERE, "cgklt_golden.npz"), **cgklt_cases())
    print("golden fixtures written")
