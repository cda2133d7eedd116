"""bundleAdjustRobust over the C-ABI (call contract: reference src/app/SL_CoSLAMRobustBA.cpp:170-180,
SL_InterCamPoseEstimator.cpp:92-95).  The reference passes vector<Mat_d> Ks,Rs,Ts, vector<Point3d> pts and
vector<vector<Meas2D>> meas; here meas is either that nested list [(viewId, x, y), ...] per point or the
flat CSR arrays the C-ABI takes."""
import ctypes as C

import numpy as np

from ._lib import CoslamHipError, check, lib


class BAStats(C.Structure):
    _fields_ = [("cost0", C.c_double), ("cost", C.c_double), ("nIterTotal", C.c_int), ("nOuter", C.c_int),
                ("nOutliers", C.c_int), ("flags", C.c_int)]   # flags: CS_BA_FLAG_CHOL_FAILED 1 | NO_PROGRESS 2 | SOLVER_TIMEOUT 4


def flatten_meas(meas2Ds):
    """vector<vector<Meas2D>> -> (obs_ptr, obs_cam, obs_xy)"""
    ptr = np.zeros(len(meas2Ds) + 1, dtype=np.int32)
    cams, xy = [], []
    for i, ms in enumerate(meas2Ds):
        for (v, x, y) in ms:
            cams.append(v)
            xy.append((x, y))
        ptr[i + 1] = len(cams)
    return ptr, np.asarray(cams, dtype=np.int32), np.asarray(xy, dtype=np.float64).reshape(-1, 2)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def bundleAdjustRobust(nCamsCon, Ks, Rs, Ts, nPtsCon, pt3Ds, meas, maxErr, maxIter, innerMaxIter, device=0):
    """In-place on Rs (C,3,3), Ts (C,3), pt3Ds (P,3) like the reference; returns (outlier[nObs], stats).
    meas: nested list per point of (viewId, x, y) or a tuple (obs_ptr, obs_cam, obs_xy)."""
    L = lib()
    if isinstance(meas, tuple):
        obs_ptr, obs_cam, obs_xy = meas
    else:
        obs_ptr, obs_cam, obs_xy = flatten_meas(meas)
    Cn, P = len(Rs), len(pt3Ds)
    Ksf = _d(Ks).reshape(Cn, 9)
    Rsf = _d(Rs).reshape(Cn, 9).copy()
    Tsf = _d(Ts).reshape(Cn, 3).copy()
    ptsf = _d(pt3Ds).reshape(P, 3).copy()
    obs_ptr = np.ascontiguousarray(obs_ptr, dtype=np.int32)
    obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
    obs_xy = _d(obs_xy).reshape(-1, 2)
    n = len(obs_cam)
    out = np.zeros(max(n, 1), dtype=np.int32)
    st = BAStats()
    check(L.cs_ba_robust(Cn, P, n, _vp(Ksf), _vp(Rsf), _vp(Tsf), _vp(ptsf), _vp(obs_ptr), _vp(obs_cam), _vp(obs_xy),
                         int(nCamsCon), int(nPtsCon), C.c_double(maxErr), int(maxIter), int(innerMaxIter), _vp(out),
                         C.byref(st), int(device)), "cs_ba_robust")
    np.asarray(Rs).reshape(Cn, 9)[:] = Rsf
    np.asarray(Ts).reshape(Cn, 3)[:] = Tsf
    np.asarray(pt3Ds).reshape(P, 3)[:] = ptsf
    return out[:n], st


class BAWorkspace:
    """Device-resident BA: upload once, re-solve on a stream from device-resident initial estimates."""

    def __init__(self, device=0):
        self._L = lib()
        self._L.cs_ba_create.restype = C.c_void_p
        h = self._L.cs_ba_create(int(device))
        if not h:
            raise CoslamHipError("cs_ba_create: " + self._L.cs_last_error().decode())
        self._h = C.c_void_p(h)

    def upload(self, Ks, Rs, Ts, pts, obs_ptr, obs_cam, obs_xy):
        self.C, self.P, self.nObs = len(Rs), len(pts), len(obs_cam)
        a = [_d(Ks).reshape(-1), _d(Rs).reshape(-1), _d(Ts).reshape(-1), _d(pts).reshape(-1),
             np.ascontiguousarray(obs_ptr, dtype=np.int32), np.ascontiguousarray(obs_cam, dtype=np.int32),
             _d(obs_xy).reshape(-1)]
        check(self._L.cs_ba_upload(self._h, self.C, self.P, self.nObs, *[_vp(x) for x in a]), "cs_ba_upload")

    def solve_dev(self, stream_ptr, d_Rs0, d_Ts0, d_pts0, nCamsCon, nPtsCon, maxErr, maxIter, innerMaxIter):
        vp = C.c_void_p
        check(self._L.cs_ba_solve_dev(self._h, vp(stream_ptr), self.C, self.P, self.nObs, vp(d_Rs0), vp(d_Ts0),
                                      vp(d_pts0), int(nCamsCon), int(nPtsCon), C.c_double(maxErr), int(maxIter),
                                      int(innerMaxIter)), "cs_ba_solve_dev")

    def solve_async(self, after_stream_ptr, d_Rs0, d_Ts0, d_pts0, nCamsCon, nPtsCon, maxErr, maxIter, innerMaxIter):
        """cs_ba_solve_async: the solve on the workspace's worker thread (the reference's BA thread), started once the work
        enqueued on `after_stream_ptr` so far has finished; returns at once."""
        vp = C.c_void_p
        check(self._L.cs_ba_solve_async(self._h, vp(after_stream_ptr), self.C, self.P, self.nObs, vp(d_Rs0), vp(d_Ts0),
                                        vp(d_pts0), int(nCamsCon), int(nPtsCon), C.c_double(maxErr), int(maxIter),
                                        int(innerMaxIter)), "cs_ba_solve_async")

    def set_stream(self, stream_ptr):
        """cs_ba_set_stream: the workspace (and its asynchronous worker) enqueue on the caller's stream -- e.g. one confined
        to a CU range; 0 restores the workspace's own.  The caller keeps the stream alive."""
        check(self._L.cs_ba_set_stream(self._h, C.c_void_p(stream_ptr)), "cs_ba_set_stream")

    def worker_stats(self):
        """(solves completed by the worker since the last call, GPU ms they held the stream in total, the last one, the longest, the
        window parses' part of the total)"""
        j, t, l, m, p = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
        check(self._L.cs_ba_worker_stats(self._h, C.byref(j), C.byref(t), C.byref(l), C.byref(m), C.byref(p)), "cs_ba_worker_stats")
        return j.value, t.value, l.value, m.value, p.value

    def wait(self):
        check(self._L.cs_ba_wait(self._h), "cs_ba_wait")

    def pending(self):
        """asynchronous solves still queued or running (never blocks): RobustBundleRTS::processed as CoSLAM's main loop reads it"""
        n = self._L.cs_ba_pending(self._h)
        if n < 0:
            check(n, "cs_ba_pending")
        return n

    def completed(self):
        """asynchronous solves finished since the workspace was created (never blocks)"""
        self._L.cs_ba_completed.restype = C.c_longlong
        return int(self._L.cs_ba_completed(self._h))

    def result_buffers(self):
        """device addresses (ints) of the workspace's current estimate: (Rs [C,9], Ts [C,3], pts [P,3])"""
        r, t, m = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self._L.cs_ba_result_buffers(self._h, C.byref(r), C.byref(t), C.byref(m)), "cs_ba_result_buffers")
        return r.value, t.value, m.value

    def set_followup(self, fn_ptr, user_ptr):
        """cs_ba_set_followup: a NATIVE function (address, e.g. coslam_amd.posegraph.after_ba_function()) and its record's
        address; enqueued behind every solve on the solve's stream.  The caller keeps the record alive.  (0, 0) removes it."""
        check(self._L.cs_ba_set_followup(self._h, C.c_void_p(fn_ptr), C.c_void_p(user_ptr)), "cs_ba_set_followup")

    def problem_buffers(self):
        """device addresses (ints) of the flat problem in the workspace: (Ks, obs_ptr, obs_cam, obs_xy)"""
        a, b, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self._L.cs_ba_problem_buffers(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "cs_ba_problem_buffers")
        return a.value, b.value, c.value, d.value

    def set_sizes(self, n_cams, n_pts, n_obs):
        """the sizes download() copies (a problem parsed on the device: BAWindow.last_problem())"""
        self.C, self.P, self.nObs = int(n_cams), int(n_pts), int(n_obs)

    def download(self):
        Rs, Ts, pts = np.zeros((self.C, 9)), np.zeros((self.C, 3)), np.zeros((max(self.P, 1), 3))
        out = np.zeros(max(self.nObs, 1), dtype=np.int32)
        st = BAStats()
        check(self._L.cs_ba_download(self._h, self.C, self.P, self.nObs, _vp(Rs), _vp(Ts), _vp(pts), _vp(out),
                                     C.byref(st)), "cs_ba_download")
        return Rs.reshape(self.C, 3, 3), Ts, pts[: self.P], out[: self.nObs], st

    def close(self):
        if self._h:
            self._L.cs_ba_destroy.argtypes = [C.c_void_p]
            self._L.cs_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BAWindow:
    """cs_ba_window: a ring of key frames on the device -- per key frame and camera the hand-back's records (undistorted pixels,
    slot -> map point) and K, R, t -- from which the bundle adjuster's inputs are parsed on the device
    (RobustBundleRTS::addKeyFrames / addPoints / parseInputs, reference src/app/SL_CoSLAMRobustBA.cpp:37-78,109-165)."""

    def __init__(self, n_cams, n_key_frames, n_slots, n_map_pts, device=0):
        self._L = lib()
        self._L.cs_ba_window_create.restype = C.c_void_p
        h = self._L.cs_ba_window_create(int(device), int(n_cams), int(n_key_frames), int(n_slots), int(n_map_pts))
        if not h:
            raise CoslamHipError("cs_ba_window_create: " + self._L.cs_last_error().decode())
        self._h = C.c_void_p(h)
        self.n_cams, self.n_key_frames, self.device = n_cams, n_key_frames, device

    def push_dev(self, stream_ptr, hb_cams, d_K, k_shared, d_R, d_t, frame):
        """hb_cams: the ctypes array handback_cams() built (its xy / state / slot2map are read); device pointers as ints"""
        vp = C.c_void_p
        check(self._L.cs_ba_window_push_dev(self._h, vp(stream_ptr), hb_cams, vp(d_K), int(k_shared), vp(d_R), vp(d_t), int(frame)),
              "cs_ba_window_push_dev")

    def solve_async(self, ws, after_stream_ptr, d_map_pts, n_cams_con, n_pts_con, max_err, max_iter, inner_max_iter, d_map_static=0):
        vp = C.c_void_p
        check(self._L.cs_ba_solve_window_async(ws._h, self._h, vp(after_stream_ptr), vp(d_map_pts), vp(d_map_static), int(n_cams_con),
                                               int(n_pts_con), C.c_double(max_err), int(max_iter), int(inner_max_iter)),
              "cs_ba_solve_window_async")

    def solve_flags_async(self, ws, after_stream_ptr, d_map_pts, d_map_flags, n_cams_con, n_pts_con, max_err, max_iter, inner_max_iter):
        """cs_ba_solve_window_flags_async: only the points that are isLocalStatic() under the map's CS_MAP_* flag bytes take part"""
        vp = C.c_void_p
        check(self._L.cs_ba_solve_window_flags_async(ws._h, self._h, vp(after_stream_ptr), vp(d_map_pts), vp(d_map_flags), int(n_cams_con),
                                                     int(n_pts_con), C.c_double(max_err), int(max_iter), int(inner_max_iter)),
              "cs_ba_solve_window_flags_async")

    def reserve(self, ws):
        """cs_ba_reserve_for_window: ws sized and bound for this window's largest problem (result_buffers() then stay valid)"""
        check(self._L.cs_ba_reserve_for_window(ws._h, self._h), "cs_ba_reserve_for_window")

    def last_problem(self):
        """(C, P, nObs, device address of the points' map indices, key-frame numbers oldest first); call after ws.wait()"""
        c, p, o, pm = C.c_int(0), C.c_int(0), C.c_int(0), C.c_void_p()
        kf = (C.c_int * self.n_key_frames)()
        check(self._L.cs_ba_window_last_problem(self._h, C.byref(c), C.byref(p), C.byref(o), C.byref(pm), kf), "cs_ba_window_last_problem")
        return c.value, p.value, o.value, pm.value, [f for f in kf if f >= 0]

    def close(self):
        if self._h:
            self._L.cs_ba_window_destroy.argtypes = [C.c_void_p]
            self._L.cs_ba_window_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BAOutput:
    """cs_ba_output: RobustBundleRTS::output() (reference src/app/SL_CoSLAMRobustBA.cpp:273-316) in two halves -- the solve's worker
    packs every window solve's result into a record of a small ring; the stream that owns the map applies a record between two
    frames (key poses into the pose history / the window / the camera graphs, points into the map, relaxation of the non-key
    frames, updateNewPosesPoints)."""

    def __init__(self, n_cams, n_key_frames, n_map_pts, n_slots=8, device=0):
        self._L = lib()
        self._L.cs_ba_output_create.restype = C.c_void_p
        h = self._L.cs_ba_output_create(int(device), int(n_cams), int(n_key_frames), int(n_map_pts), int(n_slots))
        if not h:
            raise CoslamHipError("cs_ba_output_create: " + self._L.cs_last_error().decode())
        self._h = C.c_void_p(h)
        self._L.cs_ba_output_record_bytes.restype = C.c_size_t
        self._L.cs_ba_output_record_bytes.argtypes = [C.c_void_p]
        self.record_bytes = int(self._L.cs_ba_output_record_bytes(self._h))
        self._L.cs_ba_output_packed.restype = C.c_longlong
        self._L.cs_ba_output_packed.argtypes = [C.c_void_p]
        self._L.cs_ba_output_wait.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p]
        self._L.cs_ba_output_slot.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p]
        self.n_cams, self.n_key_frames, self.n_map_pts, self.device = n_cams, n_key_frames, n_map_pts, device

    def attach(self, ws):
        check(self._L.cs_ba_output_attach(self._h, ws._h), "cs_ba_output_attach")

    def packed(self):
        return int(self._L.cs_ba_output_packed(self._h))

    def wait(self, seq):
        """blocks until record `seq` is complete on the device -> its device address"""
        p = C.c_void_p()
        check(self._L.cs_ba_output_wait(self._h, int(seq), C.byref(p)), "cs_ba_output_wait")
        return p.value

    def wait_dev(self, seq, stream_ptr, timeout_ms=0):
        """the wait on the device: work enqueued on the stream afterwards runs once record `seq` is complete -> its address, at once"""
        p = C.c_void_p()
        check(self._L.cs_ba_output_wait_dev(self._h, C.c_longlong(int(seq)), C.c_void_p(stream_ptr), int(timeout_ms), C.byref(p)),
              "cs_ba_output_wait_dev")
        return p.value

    def wait_errors(self):
        return int(self._L.cs_ba_output_wait_errors(self._h))

    def slot(self, seq):
        p = C.c_void_p()
        check(self._L.cs_ba_output_slot(self._h, int(seq), C.byref(p)), "cs_ba_output_slot")
        return p.value

    def header(self, d_record, stream_ptr=0):
        """dict(C, P, nObs, nKf, nCams, seq, ok, key_frames) of a record (synchronises the stream)"""
        h8, kf = (C.c_int * 8)(), (C.c_int * 16)()
        check(self._L.cs_ba_output_header(self._h, C.c_void_p(d_record), C.c_void_p(stream_ptr), h8, kf), "cs_ba_output_header")
        return dict(C=h8[0], P=h8[1], nObs=h8[2], nKf=h8[3], nCams=h8[4], seq=h8[5], ok=h8[6], key_frames=[f for f in kf if f >= 0])

    def arrays(self, d_record):
        """device addresses (ints): Rs, Ts, pts, pointMap, ptOutlier of a record"""
        a = [C.c_void_p() for _ in range(5)]
        check(self._L.cs_ba_output_arrays(self._h, C.c_void_p(d_record), *[C.byref(x) for x in a]), "cs_ba_output_arrays")
        return [x.value for x in a]

    def apply_dev(self, d_record, stream_ptr, history, window, pu_cams, d_pointFeat, n_map, d_mapPts, d_mapCov, d_mapFlags, pixelErrVar,
                  first_key_frame, key_every, d_Rcur, d_tcur, d_counts=0, seq=-1):
        """seq >= 0: the record's sequence number on the rank that solved the window -- a slot holding another window's record (a
        device-side wait that gave up) then applies nothing and is counted (wait_errors)"""
        vp = C.c_void_p
        check(self._L.cs_ba_output_apply_seq_dev(self._h, vp(d_record), C.c_longlong(int(seq)), vp(stream_ptr), vp(history._h),
                                                 window._h if window is not None else None,
                                                 pu_cams, vp(d_pointFeat), int(n_map), vp(d_mapPts), vp(d_mapCov), vp(d_mapFlags),
                                                 C.c_double(pixelErrVar), int(first_key_frame), int(key_every), vp(d_Rcur), vp(d_tcur),
                                                 vp(d_counts)), "cs_ba_output_apply_seq_dev")

    def apply_frames_dev(self, d_record, stream_ptr, history, window, pu_cams, d_pointFeat, n_map, d_mapPts, d_mapCov, d_mapFlags,
                         pixelErrVar, key_frames, d_Rcur, d_tcur, d_counts=0, seq=-1):
        """apply_dev for key frames that are not equally spaced (the reference's fall where genNewMapPoints' decision puts them,
        SL_CoSLAM.cpp:1294-1346): key_frames = the window's frame numbers, ascending; with seq >= 0 every one of them is compared
        with the record's header."""
        vp = C.c_void_p
        kf = (C.c_int * len(key_frames))(*[int(f) for f in key_frames])
        check(self._L.cs_ba_output_apply_frames_dev(self._h, vp(d_record), C.c_longlong(int(seq)), vp(stream_ptr), vp(history._h),
                                                    window._h if window is not None else None,
                                                    pu_cams, vp(d_pointFeat), int(n_map), vp(d_mapPts), vp(d_mapCov), vp(d_mapFlags),
                                                    C.c_double(pixelErrVar), kf, len(key_frames), vp(d_Rcur), vp(d_tcur),
                                                    vp(d_counts)), "cs_ba_output_apply_frames_dev")

    def set_feat_refs(self, d_featRef, d_refStatic=None):
        """updateNewPosesPoints of every later apply over feature references (cs_feat_ref_advance_dev's table; None: this frame's features)"""
        check(self._L.cs_ba_output_set_feat_refs(self._h, C.c_void_p(d_featRef),
                                                 C.c_void_p(d_refStatic)), "cs_ba_output_set_feat_refs")

    def set_apply_mask(self, mask):
        """diagnostic (tools/r05_drift.py): 1 key poses + relaxation, 2 points, 4 outlier points false, 8 updateNewPosesPoints; 15 = all"""
        self._L.cs_ba_output_set_apply_mask.argtypes = [C.c_void_p, C.c_int]
        check(self._L.cs_ba_output_set_apply_mask(self._h, int(mask)), "cs_ba_output_set_apply_mask")

    def close(self):
        if self._h:
            self._L.cs_ba_output_destroy.argtypes = [C.c_void_p]
            self._L.cs_ba_output_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class InterCamCam(C.Structure):
    """== cs_intercam_cam (include/coslam_hip.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("K", "xy", "state", "slot2map", "trackSpan", "isStatic")]


def intercam_cams(cams):
    arr = (InterCamCam * len(cams))()
    for a, c in zip(arr, cams):
        for n, _ in InterCamCam._fields_:
            setattr(a, n, int(c[n]))
    return arr


class BAInterCam:
    """cs_ba_intercam: InterCamPoseEstimator::addMapPoints (reference src/app/SL_InterCamPoseEstimator.cpp:18-91) built on the device
    from the frame's records when the solve is requested; the solve on a workspace's worker thread."""

    def __init__(self, n_cams, n_slots, pts_stride, n_map_pts, max_dyn=60, device=0):
        self._L = lib()
        self._L.cs_ba_intercam_create.restype = C.c_void_p
        h = self._L.cs_ba_intercam_create(int(device), int(n_cams), int(n_slots), int(pts_stride), int(n_map_pts), int(max_dyn))
        if not h:
            raise CoslamHipError("cs_ba_intercam_create: " + self._L.cs_last_error().decode())
        self._h = C.c_void_p(h)
        self.n_cams = n_cams

    def solve_async(self, ws, after_stream_ptr, cams, W, H, n_col_blk, n_row_blk, d_R, d_t, d_mapPts, d_mapFlags, d_newPt, d_pointFeat,
                    max_err=6.0, max_iter=3, inner_max_iter=40):
        """cams: the array intercam_cams() built; InterCamPoseEstimator's constants: sigma 6, maxIter 3, 40 inner (SL_InterCamPoseEstimator.h)"""
        vp = C.c_void_p
        check(self._L.cs_ba_solve_intercam_async(ws._h, self._h, vp(after_stream_ptr), cams, int(W), int(H), int(n_col_blk), int(n_row_blk),
                                                 vp(d_R), vp(d_t), vp(d_mapPts), vp(d_mapFlags), vp(d_newPt), vp(d_pointFeat),
                                                 C.c_double(max_err), int(max_iter), int(inner_max_iter)), "cs_ba_solve_intercam_async")

    def apply_dev(self, ws, stream_ptr, pu_cams, N, d_pointFeat, nMap, d_Rcur, d_tcur, d_mapPts, d_mapCov, d_mapFlags, pixelErrVar, history=None,
                  d_numNodes=None, d_numOut=None):
        """cs_ba_intercam_apply_dev: InterCamPoseEstimator::apply's write-back (reference src/app/SL_InterCamPoseEstimator.cpp:100-136) behind
        a finished solve (ws.wait()): poses into the current ones, then poseUpdate3D's gate under them.  pu_cams: a poseupdate_cams() array."""
        vp = C.c_void_p
        check(self._L.cs_ba_intercam_apply_dev(ws._h, self._h, vp(stream_ptr), vp(history._h if history is not None else None), pu_cams, int(N),
                                               vp(d_pointFeat), int(nMap), vp(d_Rcur), vp(d_tcur), vp(d_mapPts), vp(d_mapCov), vp(d_mapFlags),
                                               C.c_double(pixelErrVar), vp(d_numNodes), vp(d_numOut)), "cs_ba_intercam_apply_dev")

    def last_problem(self):
        """(C, P, nObs, nStatic, device address of the points' map indices); call after ws.wait()"""
        c, p, o, st, pm = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_void_p()
        check(self._L.cs_ba_intercam_last_problem(self._h, C.byref(c), C.byref(p), C.byref(o), C.byref(st), C.byref(pm)),
              "cs_ba_intercam_last_problem")
        return c.value, p.value, o.value, st.value, pm.value

    def close(self):
        if self._h:
            self._L.cs_ba_intercam_destroy.argtypes = [C.c_void_p]
            self._L.cs_ba_intercam_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
