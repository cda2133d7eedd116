"""Deterministic synthetic multi-camera sequence.

The reference ships no data, fixtures or tests; every parity test and the bench run on this generator.
World: P static points uniform in the box [-5,5]x[-3,3]x[6,14], drawn as Gaussian blobs over a faint
band-limited background; C cameras on a 1 m-radius arc looking at the box centre, each advancing
1.5 cm/frame with 0.1 deg/frame yaw.  K = [[0.82W,0,W/2],[0,0.82W,H/2],[0,0,1]], no distortion.
Image coordinates put pixel centres at half-integers (feature pos * W, src/tracking/GPUKLT.cpp:43-44).
Ground-truth poses, points and projections are available next to the images.
"""
import math

import numpy as np


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def look_at(cam_pos, target):
    """World->camera rotation with +z towards target, +y down (image convention)."""
    z = target - cam_pos
    z = z / np.linalg.norm(z)
    up = np.array([0.0, -1.0, 0.0])
    x = np.cross(-up, z)  # right
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=0)


class Scene:
    def __init__(self, n_cams=1, W=640, H=480, n_points=3000, seed=0xC051A, sigma=1.2, bg_amp=5.0, loop_period=0):
        """loop_period > 0: every camera moves on a smooth CLOSED curve with that period in frames (peak speed 1.5 cm/frame, peak
        yaw rate 0.1 deg/frame: the figures of the straight path) -- a video of any length that never reverses and never jumps:
        frame f and frame f + loop_period are the same image."""
        self.loop_period = int(loop_period)
        self.C, self.W, self.H, self.P = n_cams, W, H, n_points
        self.rng = np.random.Generator(np.random.MT19937(seed))
        rng = self.rng
        self.points = np.stack(
            [rng.uniform(-5, 5, n_points), rng.uniform(-3, 3, n_points), rng.uniform(6, 14, n_points)], axis=1
        )
        self.amp = rng.uniform(60, 160, n_points) * rng.choice([-1.0, 1.0], n_points)
        self.sigma = sigma
        f = 0.82 * W
        self.K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]], dtype=np.float64)
        self.target = np.array([0.0, 0.0, 10.0])
        # faint static background (band-limited noise): below any sensible minCornerness
        bg = rng.standard_normal((H, W))
        k = np.array([1, 4, 6, 4, 1], dtype=np.float64) / 16.0
        for _ in range(3):
            bg = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, bg)
            bg = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, bg)
        bg = bg / (np.abs(bg).max() + 1e-12)
        self.background = 128.0 + bg_amp * bg

    # ---- geometry ----
    def pose(self, cam, frame):
        """(R, t) with x_cam = R x_world + t."""
        a0 = (cam - (self.C - 1) / 2.0) * math.radians(12.0)  # cameras spread on the arc
        if self.loop_period > 0:
            T = self.loop_period
            ph = 2.0 * math.pi * (frame % T) / T
            amp = 0.015 * T / (2.0 * math.pi)          # peak speed amp * 2 pi / T = 1.5 cm / frame
            yaw = math.radians(0.1) * T / (2.0 * math.pi) * math.sin(ph + 0.9)
            pos = np.array([math.sin(a0) * 1.0 + amp * math.sin(ph), 0.05 * math.sin(0.7 * cam) + 0.25 * amp * math.sin(2.0 * ph),
                            -math.cos(a0) * 1.0 + 1.0 + 0.5 * amp * (1.0 - math.cos(ph))])
        else:
            yaw = math.radians(0.1) * frame
            pos = np.array([math.sin(a0) * 1.0 + 0.015 * frame, 0.05 * math.sin(0.7 * cam), -math.cos(a0) * 1.0 + 1.0])
        R = _rot_y(yaw).T @ look_at(pos, self.target)
        t = -R @ pos
        return R, t

    def project(self, cam, frame, points=None):
        R, t = self.pose(cam, frame)
        X = (self.points if points is None else points) @ R.T + t
        z = X[:, 2]
        uv = (X @ self.K.T)[:, :2] / z[:, None]
        vis = (z > 0.1) & (uv[:, 0] >= 0) & (uv[:, 0] < self.W) & (uv[:, 1] >= 0) & (uv[:, 1] < self.H)
        return uv, vis

    # ---- rendering ----
    def render(self, cam, frame):
        uv, vis = self.project(cam, frame)
        img = self.background.copy()
        r = int(math.ceil(4 * self.sigma))
        idx = np.nonzero(vis)[0]
        u, v, a = uv[idx, 0], uv[idx, 1], self.amp[idx]
        # pixel (i,j) has its centre at (i+0.5, j+0.5)
        ci = np.floor(u).astype(np.int64)
        cj = np.floor(v).astype(np.int64)
        inv2s2 = 1.0 / (2.0 * self.sigma * self.sigma)
        for dj in range(-r, r + 1):
            jj = cj + dj
            okj = (jj >= 0) & (jj < self.H)
            dy2 = (jj + 0.5 - v) ** 2
            for di in range(-r, r + 1):
                ii = ci + di
                ok = okj & (ii >= 0) & (ii < self.W)
                w = a * np.exp(-((ii + 0.5 - u) ** 2 + dy2) * inv2s2)
                np.add.at(img, (jj[ok], ii[ok]), w[ok])
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def shift_image(img, dx, dy):
    """Sub-pixel shift by bilinear resampling (known-answer flow tests): out(x,y) = img(x-dx, y-dy)."""
    H, W = img.shape
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    sx, sy = xs - dx, ys - dy
    x0 = np.clip(np.floor(sx).astype(int), 0, W - 1)
    y0 = np.clip(np.floor(sy).astype(int), 0, H - 1)
    x1 = np.clip(x0 + 1, 0, W - 1)
    y1 = np.clip(y0 + 1, 0, H - 1)
    fx = np.clip(sx - np.floor(sx), 0, 1)
    fy = np.clip(sy - np.floor(sy), 0, 1)
    f = img.astype(np.float64)
    out = (f[y0, x0] * (1 - fx) + f[y0, x1] * fx) * (1 - fy) + (f[y1, x0] * (1 - fx) + f[y1, x1] * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def blob_image(W, H, centers, sigma=2.0, amp=120.0, base=100.0):
    """Isolated Gaussian blobs at continuous positions (pixel centres at half-integers)."""
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.full((H, W), base)
    for (u, v) in centers:
        img += amp * np.exp(-(((xs + 0.5 - u) ** 2) + ((ys + 0.5 - v) ** 2)) / (2 * sigma * sigma))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ---- gauge-free comparison with the ground truth ----
def umeyama(X, Y, with_scale=True):
    """(s, R, t) minimising sum |s R X_i + t - Y_i|^2 (Umeyama 1991); X, Y: [n][3].  A SLAM map is defined up to such a transform of
    the world frame (a similarity once the scale floats with the map): estimated camera centres / map points are compared with the
    truth after it has been removed."""
    X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    U, D, Vt = np.linalg.svd(Yc.T @ Xc / len(X))
    E = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        E[2, 2] = -1
    R = U @ E @ Vt
    s = float((D * np.diag(E)).sum() / max((Xc ** 2).sum(), 1e-300) * len(X)) if with_scale else 1.0
    return s, R, my - s * R @ mx


def rig_error_vs_truth(scene, frame, R_est, t_est):
    """How far the estimated rig of `frame` is from the synthetic truth: raw (max |t - t_true| as rounds 2-4 reported it; camera centres),
    and after a similarity / a rigid alignment of the camera centres -- what is left is the rig's DISTORTION, what the alignment took
    out is a motion of the whole map + rig (nothing ties the map to the frame it started in once its points are estimated)."""
    nC = len(R_est)
    R_est, t_est = np.asarray(R_est, dtype=np.float64).reshape(nC, 3, 3), np.asarray(t_est, dtype=np.float64).reshape(nC, 3)
    Rt, tt = zip(*[scene.pose(c, frame) for c in range(nC)])
    Rt, tt = np.array(Rt), np.array(tt)
    Ce, Ct = -np.einsum("cji,cj->ci", R_est, t_est), -np.einsum("cji,cj->ci", Rt, tt)
    out = {"raw_max_abs_t": float(np.abs(t_est - tt).max()), "centres_raw_max": float(np.linalg.norm(Ce - Ct, axis=1).max())}
    for name, ws in (("sim3", True), ("rigid", False)):
        s, Ra, ta = umeyama(Ce, Ct, ws)
        out[f"centres_after_{name}_max"] = float(np.linalg.norm(s * Ce @ Ra.T + ta - Ct, axis=1).max())
        ang = float(np.degrees(np.arccos(np.clip((np.trace(Ra) - 1) / 2, -1, 1))))
        out[f"gauge_{name}"] = {"scale": s, "rot_deg": ang, "trans": float(np.linalg.norm(ta))}
    return out


# ---- bundle-adjustment problem generator (cfg1: 10 key frames x 500 points) ----
def rodrigues(w):
    th = np.linalg.norm(w)
    if th == 0:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)


def make_ba_problem(n_cams=10, n_pts=500, W=640, H=480, noise=0.5, rot_pert=0.01, trans_pert=0.03, pt_pert=0.05,
                    outlier_frac=0.05, outlier_mag=20.0, seed=0xC051A + 1, visibility=1.0, n_cams_con=2, n_pts_con=2):
    """Synthetic local-BA problem in the flat layout of cs_ba_robust.
    Returns dict with ground truth and perturbed initial values."""
    rng = np.random.Generator(np.random.MT19937(seed))
    f = 0.82 * W
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    pts = np.stack([rng.uniform(-5, 5, n_pts), rng.uniform(-3, 3, n_pts), rng.uniform(6, 14, n_pts)], axis=1)
    target = np.array([0.0, 0.0, 10.0])
    Rs, ts = [], []
    for c in range(n_cams):
        a = (c - (n_cams - 1) / 2.0) * math.radians(4.0)
        pos = np.array([math.sin(a) * 2.0, 0.1 * math.sin(1.3 * c), -math.cos(a) * 2.0 + 2.0])
        R = look_at(pos, target)
        Rs.append(R)
        ts.append(-R @ pos)
    Rs, ts = np.array(Rs), np.array(ts)
    obs_cam, obs_pt, obs_xy = [], [], []
    for i in range(n_pts):
        for c in range(n_cams):
            if visibility < 1.0 and rng.uniform() > visibility:
                continue
            X = Rs[c] @ pts[i] + ts[c]
            uv = (K @ X)[:2] / X[2]
            obs_cam.append(c)
            obs_pt.append(i)
            obs_xy.append(uv)
    obs_cam = np.array(obs_cam, dtype=np.int32)
    obs_pt = np.array(obs_pt, dtype=np.int32)
    obs_xy = np.array(obs_xy, dtype=np.float64)
    n_obs = len(obs_cam)
    obs_xy_noisy = obs_xy + noise * rng.standard_normal(obs_xy.shape)
    is_out = rng.uniform(size=n_obs) < outlier_frac
    sign = rng.choice([-1.0, 1.0], size=(n_obs, 2))
    obs_xy_noisy[is_out] += outlier_mag * sign[is_out]
    Rs0 = np.array([Rs[c] @ rodrigues(rot_pert * rng.standard_normal(3)) for c in range(n_cams)])
    ts0 = ts + trans_pert * rng.standard_normal(ts.shape)
    pts0 = pts + pt_pert * rng.standard_normal(pts.shape)
    # the held-fixed cameras / points define the gauge: they keep their true values
    Rs0[:n_cams_con] = Rs[:n_cams_con]
    ts0[:n_cams_con] = ts[:n_cams_con]
    pts0[:n_pts_con] = pts[:n_pts_con]
    return dict(K=K, Ks=np.repeat(K[None], n_cams, 0), Rs_gt=Rs, ts_gt=ts, pts_gt=pts, Rs0=Rs0, ts0=ts0, pts0=pts0,
                obs_cam=obs_cam, obs_pt=obs_pt, obs_xy=obs_xy_noisy, obs_xy_clean=obs_xy, is_outlier=is_out)


# ---- the headline workload's BA problems (8 synchronised cameras) ------------------------------------------------
def make_joint_ba_problem(scene, n_kf=5, kf_step=5, pts_per_cam=500, pool=1500, noise=0.5, rot_pert=0.005,
                          trans_pert=0.02, pt_pert=0.05, outlier_frac=0.03, outlier_mag=20.0, n_old_kf=2, n_pts_con=2,
                          seed=0xC051A + 7):
    """The joint local BA CoSLAM queues at a key frame (reference src/app/SL_CoSLAM.cpp:1345,1731-1784 ->
    RobustBundleRTS, src/app/SL_CoSLAMRobustBA.cpp:37-78,109-165): the last `n_kf` key frames of ALL cameras are the
    cameras of ONE problem, ordered key frame by key frame (addKeyFrames), the first numCams * n_old_kf of them held
    fixed; the points are the static map points seen in those key frames, each measured in every (key frame, camera) it
    was tracked in; points with a single measurement are dropped (parseInputs: nfpts > 1).
    Same dict layout as make_ba_problem, plus n_cams_con / n_pts_con."""
    rng = np.random.Generator(np.random.MT19937(seed))
    nc = scene.C
    C = n_kf * nc
    K = scene.K
    Rs, ts = [], []
    for kf in range(n_kf):
        for c in range(nc):
            R, t = scene.pose(c, kf * kf_step)
            Rs.append(R)
            ts.append(t)
    Rs, ts = np.array(Rs), np.array(ts)
    # a pool of map points; every camera tracks `pts_per_cam` of those it sees in all its key frames
    vis_all = []
    for c in range(nc):
        v = np.ones(scene.P, dtype=bool)
        for kf in range(n_kf):
            _, vk = scene.project(c, kf * kf_step)
            v &= vk
        vis_all.append(v)
    seen = np.nonzero(np.any(vis_all, axis=0))[0]
    pool_idx = rng.choice(seen, size=min(pool, len(seen)), replace=False)
    tracked = np.zeros((nc, len(pool_idx)), dtype=bool)
    for c in range(nc):
        cand = np.nonzero(vis_all[c][pool_idx])[0]
        pick = rng.choice(cand, size=min(pts_per_cam, len(cand)), replace=False)
        tracked[c, pick] = True
    keep = np.nonzero(tracked.sum(0) * n_kf > 1)[0]
    pts = scene.points[pool_idx[keep]]
    tracked = tracked[:, keep]
    obs_cam, obs_pt, obs_xy = [], [], []
    for i in range(len(pts)):
        for kf in range(n_kf):
            for c in range(nc):
                if not tracked[c, i]:
                    continue
                j = kf * nc + c
                X = Rs[j] @ pts[i] + ts[j]
                obs_cam.append(j)
                obs_pt.append(i)
                obs_xy.append((K @ X)[:2] / X[2])
    obs_cam = np.array(obs_cam, dtype=np.int32)
    obs_pt = np.array(obs_pt, dtype=np.int32)
    obs_xy = np.array(obs_xy, dtype=np.float64)
    n_obs = len(obs_cam)
    noisy = obs_xy + noise * rng.standard_normal(obs_xy.shape)
    is_out = rng.uniform(size=n_obs) < outlier_frac
    noisy[is_out] += outlier_mag * rng.choice([-1.0, 1.0], size=(n_obs, 2))[is_out]
    n_con = nc * n_old_kf
    Rs0 = np.array([Rs[j] @ rodrigues(rot_pert * rng.standard_normal(3)) for j in range(C)])
    ts0 = ts + trans_pert * rng.standard_normal(ts.shape)
    pts0 = pts + pt_pert * rng.standard_normal(pts.shape)
    Rs0[:n_con], ts0[:n_con] = Rs[:n_con], ts[:n_con]
    pts0[:n_pts_con] = pts[:n_pts_con]
    return dict(K=K, Ks=np.repeat(K[None], C, 0), Rs_gt=Rs, ts_gt=ts, pts_gt=pts, Rs0=Rs0, ts0=ts0, pts0=pts0,
                obs_cam=obs_cam, obs_pt=obs_pt, obs_xy=noisy, obs_xy_clean=obs_xy, is_outlier=is_out,
                n_cams_con=n_con, n_pts_con=n_pts_con)


def csr_of_problem(pr):
    """(obs_ptr, obs_cam, obs_xy) of a generated problem's measurements grouped by point (the flat layout of cs_ba_upload)"""
    P = len(pr["pts0"])
    obs_pt = np.asarray(pr["obs_pt"])
    order = np.argsort(obs_pt, kind="stable")
    ptr = np.zeros(P + 1, dtype=np.int32)
    np.add.at(ptr, obs_pt + 1, 1)
    return np.cumsum(ptr).astype(np.int32), pr["obs_cam"][order], pr["obs_xy"][order]


def make_intercam_problem(scene, frame=10, n_static=192, n_dyn=60, noise=0.5, rot_pert=0.004, trans_pert=0.015,
                          dyn_pert=0.05, outlier_frac=0.02, outlier_mag=20.0, seed=0xC051A + 11):
    """The inter-camera pose solve (reference src/app/SL_InterCamPoseEstimator.cpp:18-95): cameras = the current pose
    of every camera, all free (nCamsCon 0); points = per camera the static feature points chosen for pose estimation
    (<= 192, ONE measurement each, held fixed: nPtsCon = numStatic) followed by <= 61 dynamic points measured in every
    camera that sees them (free).  bundleAdjustRobust(0, ..., numStatic, ..., sigma 6, maxIter 3, 40 inner)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    nc = scene.C
    K = scene.K
    Rs, ts = zip(*[scene.pose(c, frame) for c in range(nc)])
    Rs, ts = np.array(Rs), np.array(ts)
    uv, vis = zip(*[scene.project(c, frame) for c in range(nc)])
    pts, obs_cam, obs_pt, obs_xy = [], [], [], []
    used = np.zeros(scene.P, dtype=bool)
    for c in range(nc):
        cand = np.nonzero(vis[c])[0]
        pick = rng.choice(cand, size=min(n_static, len(cand)), replace=False)
        used[pick] = True
        for p in pick:
            obs_cam.append(c)
            obs_pt.append(len(pts))
            obs_xy.append(uv[c][p])
            pts.append(scene.points[p])
    n_stat = len(pts)
    nvis = np.sum(vis, axis=0)
    cand = np.nonzero((nvis >= 2) & ~used)[0]
    for p in rng.choice(cand, size=min(n_dyn, len(cand)), replace=False):
        for c in range(nc):
            if vis[c][p]:
                obs_cam.append(c)
                obs_pt.append(len(pts))
                obs_xy.append(uv[c][p])
        pts.append(scene.points[p])
    pts = np.array(pts)
    obs_cam = np.array(obs_cam, dtype=np.int32)
    obs_pt = np.array(obs_pt, dtype=np.int32)
    obs_xy = np.array(obs_xy, dtype=np.float64)
    n_obs = len(obs_cam)
    noisy = obs_xy + noise * rng.standard_normal(obs_xy.shape)
    is_out = rng.uniform(size=n_obs) < outlier_frac
    noisy[is_out] += outlier_mag * rng.choice([-1.0, 1.0], size=(n_obs, 2))[is_out]
    Rs0 = np.array([Rs[c] @ rodrigues(rot_pert * rng.standard_normal(3)) for c in range(nc)])
    ts0 = ts + trans_pert * rng.standard_normal(ts.shape)
    pts0 = pts.copy()
    pts0[n_stat:] += dyn_pert * rng.standard_normal((len(pts) - n_stat, 3))
    return dict(K=K, Ks=np.repeat(K[None], nc, 0), Rs_gt=Rs, ts_gt=ts, pts_gt=pts, Rs0=Rs0, ts0=ts0, pts0=pts0,
                obs_cam=obs_cam, obs_pt=obs_pt, obs_xy=noisy, obs_xy_clean=obs_xy, is_outlier=is_out,
                n_cams_con=0, n_pts_con=n_stat, n_static=n_stat, n_dynamic=len(pts) - n_stat)


def make_pose_graphs(n_cams=8, n_frames=21, key_every=5, rot_adj=0.01, trans_adj=0.05, seed=0, loop_edges=0, loop_noise=0.01,
                     free_tail=0):
    """Camera pose graphs the way RobustBundleRTS::constructCameraGraphs / output() build them (reference
    src/app/SL_CoSLAMRobustBA.cpp:182-229, 283-294): per camera a chain of n_frames poses, every key_every-th one a key frame
    (fixed), edges = relative transform of consecutive poses BEFORE the adjustment, key poses then moved by a small rigid
    correction (what a BA does).  loop_edges extra (i -> j) edges per camera with noisy relative transforms; free_tail frames
    after the last key frame.  Returns dict(graphs=[(fixed, id1, id2)], nodeR0, nodeT0 (before), nodeR, nodeT (fixed nodes
    adjusted), ge1, ge2 (edges' ends as flat node indices), node_ptr, edge_ptr)."""
    rng = np.random.default_rng(seed)
    graphs, R0, T0, R1, T1 = [], [], [], [], []
    for _c in range(n_cams):
        n = n_frames + free_tail
        w = rng.uniform(-0.3, 0.3, 3)
        p = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), 4 + rng.uniform(-1, 1)])
        dw, dp = rng.uniform(-0.02, 0.02, 3), rng.uniform(-0.05, 0.05, 3)
        fixed = np.zeros(n, dtype=np.uint8)
        fixed[0:n_frames:key_every] = 1
        for i in range(n):
            R = rodrigues(w)
            R0.append(R.reshape(9)), T0.append(p.copy())
            if fixed[i]:
                R1.append((rodrigues(rng.uniform(-rot_adj, rot_adj, 3)) @ R).reshape(9))
                T1.append(p + rng.uniform(-trans_adj, trans_adj, 3))
            else:
                R1.append(R.reshape(9)), T1.append(p.copy())
            dw += rng.uniform(-0.004, 0.004, 3)
            dp += rng.uniform(-0.01, 0.01, 3)
            w = w + dw
            p = p + dp
        id1 = list(range(n - 1))
        id2 = list(range(1, n))
        for _k in range(loop_edges):
            a, b = rng.choice(n, 2, replace=False)
            if abs(int(a) - int(b)) > 6:
                b = int(a) + int(np.sign(int(b) - int(a))) * int(rng.integers(2, 6))
            id1.append(int(a)), id2.append(int(b))
        graphs.append((fixed, np.array(id1, np.int32), np.array(id2, np.int32)))
    node_ptr = np.concatenate([[0], np.cumsum([len(g[0]) for g in graphs])]).astype(np.int32)
    edge_ptr = np.concatenate([[0], np.cumsum([len(g[1]) for g in graphs])]).astype(np.int32)
    base = np.repeat(node_ptr[:-1], np.diff(edge_ptr))
    ge1 = np.concatenate([g[1] for g in graphs]) + base
    ge2 = np.concatenate([g[2] for g in graphs]) + base
    out = dict(graphs=graphs, nodeR0=np.array(R0), nodeT0=np.array(T0), nodeR=np.array(R1), nodeT=np.array(T1), ge1=ge1, ge2=ge2,
               node_ptr=node_ptr, edge_ptr=edge_ptr)
    # relative transforms of the chain edges follow from the poses; the loop edges get a measurement error on top
    nchain = [n_frames + free_tail - 1] * n_cams
    eR, eT = [], []
    for e in range(len(ge1)):
        Ra, Rb = out["nodeR0"][ge1[e]].reshape(3, 3), out["nodeR0"][ge2[e]].reshape(3, 3)
        R = Rb @ Ra.T
        t = out["nodeT0"][ge2[e]] - R @ out["nodeT0"][ge1[e]]
        g = int(np.searchsorted(edge_ptr, e, side="right") - 1)
        if e - edge_ptr[g] >= nchain[g]:
            R = rodrigues(rng.uniform(-loop_noise, loop_noise, 3)) @ R
            t = t + rng.uniform(-loop_noise, loop_noise, 3)
        eR.append(R.reshape(9)), eT.append(t)
    out["edgeR"], out["edgeT"] = np.array(eR), np.array(eT)
    return out
