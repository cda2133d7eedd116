"""intraCamEstimate over the C-ABI (reference src/slam/SL_IntraCamPose.h:92-95): same name, same
argument meaning, same bool result; IntraCamPoseOption mirrors the reference class field by field."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class IntraCamPoseOption(C.Structure):
    """== class IntraCamPoseOption, src/slam/SL_IntraCamPose.h:19-57."""

    _fields_ = [
        ("maxIterLM", C.c_int), ("maxIterRW", C.c_int),
        ("epsErrorChangeLM", C.c_double), ("epsParamChangeLM", C.c_double), ("epsErrorChangeRW", C.c_double),
        ("verboseLM", C.c_int), ("verboseRW", C.c_int),
        ("lambda0", C.c_double), ("lambda_", C.c_double),
        ("err0", C.c_double), ("err", C.c_double), ("errRW", C.c_double),
        ("retTypeLM", C.c_int), ("npts", C.c_int), ("nIterLM", C.c_int), ("nIterRW", C.c_int),
    ]

    def __init__(self):
        super().__init__()
        self.maxIterLM, self.maxIterRW = 100, 5
        self.epsErrorChangeLM, self.epsParamChangeLM, self.epsErrorChangeRW = 1e-7, 1e-6, 1e-6
        self.lambda0 = 1e-3


assert C.sizeof(IntraCamPoseOption) == 96


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def intraCamEstimate(K, R0, t0, npts, prevErrs, Ms, ms, tau, opt=None, device=0):
    """Returns (ok, R_opt[3,3], t_opt[3], opt).  ok is the reference's bool."""
    L = lib()
    K, R0, t0, Ms, ms = _d(K).ravel(), _d(R0).ravel(), _d(t0).ravel(), _d(Ms).ravel(), _d(ms).ravel()
    assert K.size == 9 and R0.size == 9 and t0.size == 3 and Ms.size >= 3 * npts and ms.size >= 2 * npts
    pe = None if prevErrs is None else _d(prevErrs).ravel()
    R_opt, t_opt = np.zeros(9), np.zeros(3)
    opt = opt or IntraCamPoseOption()
    vp = C.c_void_p
    rc = L.cs_pose_intracam(K.ctypes.data_as(vp), R0.ctypes.data_as(vp), t0.ctypes.data_as(vp), int(npts),
                            None if pe is None else pe.ctypes.data_as(vp), Ms.ctypes.data_as(vp),
                            ms.ctypes.data_as(vp), C.c_double(tau), R_opt.ctypes.data_as(vp), t_opt.ctypes.data_as(vp),
                            C.byref(opt), int(device))
    if rc < 0:
        check(rc, "cs_pose_intracam")
    return bool(rc), R_opt.reshape(3, 3), t_opt, opt


def intraCamEstimate_batch_dev(stream_ptr, nProb, ptsStride, d_K, d_R0, d_t0, d_npts, d_prevErrs, d_Ms, d_ms, tau,
                               d_Ropt, d_topt, d_opt, d_ok, device=0):
    """Device-resident batch (one workgroup per camera); all d_* are device pointers (ints)."""
    L = lib()
    vp = C.c_void_p
    check(L.cs_pose_intracam_batch_dev(int(device), vp(stream_ptr), int(nProb), int(ptsStride), vp(d_K), vp(d_R0),
                                       vp(d_t0), vp(d_npts), vp(d_prevErrs) if d_prevErrs else None, vp(d_Ms), vp(d_ms),
                                       C.c_double(tau), vp(d_Ropt), vp(d_topt), vp(d_opt), vp(d_ok)),
          "cs_pose_intracam_batch_dev")
