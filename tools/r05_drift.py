#!/usr/bin/env python
"""r05_drift.py -- where does the closed loop's pose error come from?  ("next round" item 1)

Runs the headline frame loop (coslam_amd.frameloop.FrameLoop: bench.py's loop, one rank) for --frames frames under ONE named
variant and, every --every frames, drains the device and logs

    pose error vs the synthetic truth     raw (bench.py's figure: max |t_est - t_true|; camera centres) AND after a similarity /
                                          a rigid alignment of the 8 estimated camera centres onto the true ones -- a motion of the
                                          whole map + rig (the gauge: nothing ties a SLAM map to the world frame it started in) is
                                          thereby separated from a distortion of the rig
    the alignment itself                  rotation angle, translation, scale of the gauge motion
    map error                             over the map points this frame's static features USE (of the 7000 initial points: their
                                          true positions are known): raw, after the cameras' alignment, after their own alignment
    joint BA                              converged cost per measurement of the last window solved, LM steps, outliers
    bookkeeping                           static mapped features per camera (on initial / on new points), pose correspondences, LM
                                          steps of intraCamEstimate, map points false / in use, attachments of the registration

as JSON lines (one per sample) to --out.  tools/r05_drift.sh runs the variants one process each; profiles/r05_drift_*.jsonl are
its outputs, profiles/r05_drift_summary.md the table DESIGN.md quotes.  GPU only (the HIP path has no CPU fallback).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    # name: (LoopConfig overrides, BA apply mask or None, extra)
    "full": ({}, None),
    "no_decide": (dict(with_decide=False), None),
    "no_ncc": (dict(with_ncc=False), None),
    "no_decide_no_ncc": (dict(with_decide=False, with_ncc=False), None),
    "no_writeback": ({}, 0),                       # the window BAs run, nothing is applied
    "r03_like": (dict(with_decide=False, with_ncc=False), 0),
    "lag1": (dict(ba_lag=1), None),
    "lag4": (dict(ba_lag=4), None),
    "points_only": ({}, 2 | 4),                    # adjusted points + false flags; key poses untouched, no re-triangulation
    "poses_only": ({}, 1 | 8),                     # key poses + relaxation + updateNewPosesPoints; adjusted points dropped
    "no_false": ({}, 1 | 2 | 8),                   # everything but "a point with an outlier measurement becomes false"
    "no_update": ({}, 1 | 2 | 4),                  # everything but updateNewPosesPoints
    "no_classify": (dict(with_classify=False), None),
    "no_merge": (dict(merge_every=0), None),
    "no_intercam": (dict(with_intercam=False), None),
    "no_chains": (dict(feature_chains=False), None),
    "classify_plain": (dict(classify_refs=False), None),                 # round 6: mapPointsClassify back on this frame's features
    "no_rounds": (dict(revisit_rounds=0), None),                         # round 6: the single pass alone (no second visits)
    "kf_drives": (dict(keyframe_drives=True), None),                     # the key frames where the decision puts them (m_mappedPtsReduceRatio = 0.93)
    "kf_drives_r098": (dict(keyframe_drives=True, keyframe_ratio=0.98), None),
    "kf_drives_r100": (dict(keyframe_drives=True, keyframe_ratio=1.0), None),
    "kf_drives_r102": (dict(keyframe_drives=True, keyframe_ratio=1.02), None),
    "intracam": (dict(intracam_mapping=True, map_spare=50000), None),   # + SingleSLAM::newMapPoints for the cameras that are ready for a key frame   # this frame's features on their own tracks (the state before the feature references)
}


def umeyama(X, Y, with_scale=True):
    """similarity (s, R, t) minimising sum |s R X_i + t - Y_i|^2 (Umeyama 1991); X, Y: [n][3]"""
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    S = Yc.T @ Xc / len(X)
    U, D, Vt = np.linalg.svd(S)
    E = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        E[2, 2] = -1
    R = U @ E @ Vt
    s = float((D * np.diag(E)).sum() / (Xc ** 2).sum() * len(X)) if with_scale else 1.0
    t = my - s * R @ mx
    return s, R, t


def rot_angle_deg(R):
    return float(np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="full", choices=sorted(VARIANTS))
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--hist", type=int, default=64)
    ap.add_argument("--hist-store", type=int, default=0, help="frames the track history keeps (0: LoopConfig's 4096)")
    ap.add_argument("--map-spare", type=int, default=0, help="room for new map points behind the initial map (0: LoopConfig's 8192)")
    ap.add_argument("--min-distance", type=int, default=-1, help="KLT minDistance (-1: bench.py's)")
    ap.add_argument("--out", default="")
    ap.add_argument("--pixel-err-reading", choices=["variance", "std"], default="variance")
    ap.add_argument("--time-intracam", action="store_true", help="at the end: k_intracam alone on the loop's last inputs (HIP events)")
    ap.add_argument("--count-attach", action="store_true", help="sum the registration's attachments over the run (a device read-back per frame)")
    args = ap.parse_args()

    import bench

    frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)

    import torch

    import coslam_amd
    from coslam_amd.frameloop import FrameLoop, LoopConfig
    from coslam_amd.pose import IntraCamPoseOption

    if not torch.cuda.is_available():
        raise SystemExit("r05_drift.py needs an MI355X")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sc = bench.build_scene()
    NA, N = bench.N_CAMS, bench.N_FEAT
    video = {c: torch.from_numpy(frames[c]).to(dev) for c in range(NA)}
    over, mask = VARIANTS[args.variant]
    kw = dict(n_cams=NA, W=bench.W, H=bench.H, levels=bench.LEVELS, fw=bench.FW, fh=bench.FH, pts_stride=bench.PTS_STRIDE,
              n_col_blk=bench.N_COL_BLK, n_row_blk=bench.N_ROW_BLK, key_every=bench.KEY_EVERY, p_reg=bench.P_REG, hist=args.hist,
              pixel_err_reading=args.pixel_err_reading)
    kw.update(over)
    if args.hist_store > 0:
        kw["hist_store"] = args.hist_store
    if args.map_spare > 0:
        kw["map_spare"] = args.map_spare
    cfg = LoopConfig(**kw)
    kc = bench.klt_config()
    if args.min_distance > 0:
        kc.minDistance = args.min_distance
    n_map0 = len(sc.points)
    loop = FrameLoop(cfg, sc, video, None, kc, bench.reg_covariances(n_map0), rank=0, world=1, device=0, associate=bench.associate)
    if mask is not None and loop.out is not None:
        loop.out.set_apply_mask(mask)
    loop.first_frame()
    truth_pts = np.ascontiguousarray(sc.points, dtype=np.float64)
    out = open(args.out, "w") if args.out else sys.stdout
    ke = cfg.key_every
    attached_total = [0]

    def sample(i):
        loop.drain()
        dst = i & 1
        R = loop.d_R[dst].cpu().numpy().reshape(NA, 3, 3)
        t = loop.d_t[dst].cpu().numpy().reshape(NA, 3)
        f = loop.vid(i)
        Rt, tt = zip(*[sc.pose(c, f) for c in range(NA)])
        Rt, tt = np.array(Rt), np.array(tt)
        Ce = -np.einsum("cji,cj->ci", R, t)
        Ct = -np.einsum("cji,cj->ci", Rt, tt)
        rec = {"variant": args.variant, "frame": i, "t_err_max": float(np.abs(t - tt).max()),
               "centre_err_raw_max": float(np.linalg.norm(Ce - Ct, axis=1).max())}
        # rotation error raw: angle of R_est R_true^T
        rec["rot_err_raw_deg_max"] = max(rot_angle_deg(R[c] @ Rt[c].T) for c in range(NA))
        for name, ws in (("sim", True), ("rigid", False)):
            s, Ra, ta = umeyama(Ce, Ct, ws)
            Ca = s * Ce @ Ra.T + ta
            rec[f"centre_err_{name}_max"] = float(np.linalg.norm(Ca - Ct, axis=1).max())
            rec[f"centre_err_{name}_rms"] = float(np.sqrt((np.linalg.norm(Ca - Ct, axis=1) ** 2).mean()))
            # a world transform x' = s Ra x + ta turns the world->camera rotation R into R Ra^T
            rec[f"rot_err_{name}_deg_max"] = max(rot_angle_deg(R[c] @ Ra.T @ Rt[c].T) for c in range(NA))
            rec[f"gauge_{name}"] = {"scale": s, "rot_deg": rot_angle_deg(Ra), "trans": float(np.linalg.norm(ta))}
            if name == "sim":
                cam_align = (s, Ra, ta)
        # the map points this frame's static features use
        state = loop.d_state.cpu().numpy()
        s2m = loop.d_slot2map.cpu().numpy()
        isst = loop.d_isstatic.cpu().numpy()
        flags = loop.d_mapflags.cpu().numpy()
        M = loop.d_map.cpu().numpy()
        live = (state >= 0) & (s2m >= 0)
        used = np.unique(s2m[live & (isst != 0)])
        used = used[(flags[used] & 3) == 0]                 # local static, not false
        u0 = used[used < n_map0]
        rec["static_mapped_features_per_cam"] = [int(v) for v in live.sum(1)]
        rec["features_on_initial_points"] = int((live & (s2m < n_map0)).sum())
        rec["features_on_new_points"] = int((live & (s2m >= n_map0)).sum())
        rec["used_points_initial"], rec["used_points_new"] = int(len(u0)), int(len(used) - len(u0))
        if len(u0) >= 4:
            e = np.linalg.norm(M[u0] - truth_pts[u0], axis=1)
            rec["map_err_raw"] = {"median": float(np.median(e)), "p90": float(np.percentile(e, 90)), "max": float(e.max())}
            s, Ra, ta = cam_align
            e = np.linalg.norm(s * M[u0] @ Ra.T + ta - truth_pts[u0], axis=1)
            rec["map_err_cam_aligned"] = {"median": float(np.median(e)), "p90": float(np.percentile(e, 90))}
            s, Ra, ta = umeyama(M[u0], truth_pts[u0], True)
            e = np.linalg.norm(s * M[u0] @ Ra.T + ta - truth_pts[u0], axis=1)
            rec["map_err_self_aligned"] = {"median": float(np.median(e)), "p90": float(np.percentile(e, 90)),
                                           "gauge": {"scale": s, "rot_deg": rot_angle_deg(Ra), "trans": float(np.linalg.norm(ta))}}
            # the poses in the MAP's gauge: what intraCamEstimate can know
            Ca = s * Ce @ Ra.T + ta
            rec["centre_err_map_aligned_max"] = float(np.linalg.norm(Ca - Ct, axis=1).max())
        rec["map_points_false"] = int(((flags & 2) != 0).sum())
        rec["map_points_in_use"] = int(loop.d_mapcount.item())
        rec["pose_correspondences"] = loop.d_npts.cpu().numpy().tolist()
        opts = [IntraCamPoseOption.from_buffer_copy(loop.d_opt[c].cpu().numpy().tobytes()) for c in range(NA)]
        rec["pose_rounds_lm"] = [[o.nIterRW, o.verboseRW] for o in opts]
        if loop.win is not None and loop.n_my_solves > 0:
            wC, wP, wO, _, wkf = loop.win.last_problem()
            loop.ba_ws.set_sizes(wC, wP, wO)
            st = loop.ba_ws.download()[4]
            rec["joint_ba"] = {"points": wP, "meas": wO, "lm_steps": st.nIterTotal, "outliers": st.nOutliers, "cost0": st.cost0, "cost": st.cost,
                               "cost_per_meas": st.cost / max(wO, 1), "cost0_per_meas": st.cost0 / max(wO, 1)}
        if loop.n_my_ic > 0 and loop.icam is not None:
            iC, iP, iO, iS, _ = loop.icam.last_problem()
            loop.ic_ws.set_sizes(iC, iP, iO)
            st = loop.ic_ws.download()[4]
            rec["intercam"] = {"static": iS, "dynamic": iP - iS, "outliers": st.nOutliers, "cost_per_meas": st.cost / max(iO, 1)}
        if hasattr(loop, "_dec"):
            rec["decide_counts"] = loop._dec["cnt"].cpu().tolist()
            cand = loop.reg_out["slot"] >= 0
            rec["current_points_listed"] = int(loop.d_curcount.item())
            rec["candidates"] = int(cand.sum().item())
            rec["mergeable_verdicts"] = {str(v): int(((loop.d_mergeable == v) & cand).sum().item()) for v in (0, 1, 2)}
            rec["running_verdict_counts"] = loop.d_merge_counts.cpu().tolist()   # hits, full tail walks, unjudged, tail terms (run totals)
            rec["attached_total"] = attached_total[0]
        if loop.out is not None:
            rec["apply_counts"] = loop.d_apply_counts.cpu().tolist()
            rec["apply_errors"] = loop.out.wait_errors()
        out.write(json.dumps(rec) + "\n")
        out.flush()

    for i in range(1, args.frames + 1):
        loop.step(i, ke > 0 and (i - 1) % ke == 0)
        if args.count_attach and hasattr(loop, "_dec"):   # (a device read-back per frame: diagnostic runs only)
            attached_total[0] += int(loop._dec["cnt"][0].item())
        if i % args.every == 0 or i == 1:
            sample(i)
    loop.drain()
    if getattr(loop, "kf", None):
        print(json.dumps({"keyframe_stats": loop.keyframe_stats()}), file=out, flush=True)
    if args.time_intracam:
        # k_intracam ALONE on the loop's own last inputs (this frame's correspondences, the previous frame's poses): is its time in the loop
        # (~105 us) the co-residency with the tracker / the solves, or the LM steps its data ask for?
        src = (args.frames + 1) & 1
        from coslam_amd.pose import intraCamEstimate_batch_dev

        ps = loop.pose_s
        opt0 = loop.d_opt.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        Ro, to = torch.zeros_like(loop.d_R[0]), torch.zeros_like(loop.d_t[0])
        tot, n = 0.0, 200

        def one():
            with torch.cuda.stream(ps):
                loop.d_opt.copy_(opt0)
            intraCamEstimate_batch_dev(ps.cuda_stream, NA, cfg.pts_stride, loop.d_K.data_ptr(), loop.d_R[src].data_ptr(), loop.d_t[src].data_ptr(),
                                       loop.d_npts.data_ptr(), 0, loop.d_Ms.data_ptr(), loop.d_ms.data_ptr(), 10.0, Ro.data_ptr(), to.data_ptr(),
                                       loop.d_opt.data_ptr(), loop.d_ok.data_ptr(), device=0)

        for _ in range(20):
            one()
        torch.cuda.synchronize()
        for _ in range(n):
            with torch.cuda.stream(ps):
                loop.d_opt.copy_(opt0)
            e0.record(ps)
            intraCamEstimate_batch_dev(ps.cuda_stream, NA, cfg.pts_stride, loop.d_K.data_ptr(), loop.d_R[src].data_ptr(), loop.d_t[src].data_ptr(),
                                       loop.d_npts.data_ptr(), 0, loop.d_Ms.data_ptr(), loop.d_ms.data_ptr(), 10.0, Ro.data_ptr(), to.data_ptr(),
                                       loop.d_opt.data_ptr(), loop.d_ok.data_ptr(), device=0)
            e1.record(ps)
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        opts = [IntraCamPoseOption.from_buffer_copy(loop.d_opt[c].cpu().numpy().tobytes()) for c in range(NA)]
        (out if args.out else sys.stdout).write(json.dumps({"variant": args.variant, "intracam_alone_us": tot / n * 1e3, "npts": loop.d_npts.cpu().tolist(),
                                                            "rounds_lm": [[o.nIterRW, o.verboseRW] for o in opts]}) + "\n")
    if args.out:
        out.close()


if __name__ == "__main__":
    main()
