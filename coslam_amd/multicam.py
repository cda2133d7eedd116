"""Inter-camera merge step for the one-camera-per-GPU layout.

The reference tracks all cameras serially inside one process and reads every camera's features and pose
directly (src/app/SL_CoSLAM.cpp:299-305, SL_InterCamPoseEstimator.cpp:24-37).  With one camera per rank the
same information is exchanged once per frame with a single all-gather of a fixed-size record:
    N x KLT_TrackedFeature (20 B each)  ||  R (9 f64)  ||  t (3 f64)
packed as int32 words.  `torch.distributed` backend "nccl" is RCCL over xGMI on MI355X; the CPU tests run the
same code over gloo.  The payload is ~40 KB per rank: latency-bound, so it is ONE collective per frame.
"""
import numpy as np
import torch
import torch.distributed as dist

from contextlib import nullcontext as _nullcontext

FEATURE_WORDS = 5   # sizeof(KLT_TrackedFeature) / 4
POSE_WORDS = 24     # 12 doubles


def feature_words_padded(n_features):
    """int32 words of one camera's feature part: N x 5, padded to an even count so that R | t behind it are 8-byte aligned"""
    return (n_features * FEATURE_WORDS + 1) & ~1


def record_words(n_features):
    """int32 words of ONE camera's record: its N features (padded to 8 bytes), R (9 doubles), t (3 doubles) -- the layout of
    both the torch path below and the native path (csrc/comm.hip)"""
    return feature_words_padded(n_features) + POSE_WORDS


class NativeComm:
    """RCCL communicators owned by libcoslam_hip (cs_comm_*): the collectives of the merge step are then issued from C++
    (one fused pack kernel + ncclAllGather per frame; the whole sliced-BA schedule with its ncclAllReduce calls enqueued
    natively).  Rank 0 creates the unique ids and ships them through torch.distributed; two communicators, because the
    per-frame exchange and the BA's all-reduces run on different streams."""

    def __init__(self, world, rank, device, group=None):
        import ctypes as C

        from ._lib import CoslamHipError, check, lib

        self._L, self._C = lib(), C
        self.world, self.rank, self.device = world, rank, device
        L = self._L
        L.cs_comm_create.restype = C.c_void_p
        L.cs_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        self.comms = []
        # no rank may enter ncclCommInitRank unless ALL can: (1) every rank can load RCCL (one all-reduce), (2) rank 0's unique id arrives
        # with a flag that says it was created -- a failure then raises on EVERY rank instead of leaving the others waiting
        if world > 1:
            backend = dist.get_backend(group)
            dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
            ok = torch.tensor([int(L.cs_comm_available())], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                raise CoslamHipError("cs_comm_available: RCCL cannot be loaded on at least one rank (" + L.cs_last_error().decode() + ")")
        for _ in range(2):
            ident = (C.c_ubyte * 128)()
            made = 1
            if rank == 0 and L.cs_comm_unique_id(ident) != 0:
                made = 0
            if world > 1:
                t = torch.tensor([made] + list(ident), dtype=torch.uint8, device=dev)
                dist.broadcast(t, src=0, group=group)
                tl = t.cpu().tolist()
                made, ident = tl[0], (C.c_ubyte * 128)(*tl[1:])
            if not made:
                raise CoslamHipError("cs_comm_unique_id failed on rank 0: " + L.cs_last_error().decode())
            h = L.cs_comm_create(ident, world, rank, device)
            if not h:
                raise CoslamHipError("cs_comm_create: " + L.cs_last_error().decode())
            self.comms.append(C.c_void_p(h))
        self.exchange_comm, self.ba_comm = self.comms

    def close(self):
        self._L.cs_comm_destroy.argtypes = [self._C.c_void_p]
        for c in self.comms:
            self._L.cs_comm_destroy(c)
        self.comms = []


class CameraExchange:
    """Preallocated send/recv buffers + the per-frame all-gather.  n_features counts the feature slots of ALL cameras of
    this rank (cameras_per_rank x N); the record of a rank is its cameras' dest[] arrays back to back, then R, t of each.
    With `native` (a NativeComm) packing and the collective are one C-ABI call (cs_exchange_allgather_dev)."""

    def __init__(self, n_features, device, group=None, native=None, cams_per_rank=1):
        self.n = n_features
        self.group = group
        self.native = native
        self.cams = cams_per_rank
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._x = None
        if native is not None:
            import ctypes as C

            from ._lib import CoslamHipError

            L = native._L
            L.cs_exchange_create.restype = C.c_void_p
            L.cs_exchange_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
            h = L.cs_exchange_create(native.exchange_comm, cams_per_rank, n_features // cams_per_rank)
            if not h:
                raise CoslamHipError("cs_exchange_create: " + L.cs_last_error().decode())
            self._x = C.c_void_p(h)
            self.world = native.world
            return
        w = record_words(n_features // cams_per_rank) * cams_per_rank   # per-camera records back to back
        self.send = torch.zeros(w, dtype=torch.int32, device=device)
        self.recv = torch.zeros(w * self.world, dtype=torch.int32, device=device)

    def pack_group(self, dest_words_list, R, t, stream=None):
        """dest_words_list: one int32[N*5] tensor per local camera; R: f64[cams, 9]; t: f64[cams, 3] (same device).
        torch path: the copies go out on the current stream.  Native path: nothing here, all_gather() packs."""
        self._pending = (dest_words_list, R, t)
        if self._x is not None:
            return
        n1 = self.n // self.cams
        nf, fp, rw = n1 * FEATURE_WORDS, feature_words_padded(n1), record_words(n1)
        Rw, tw = R.reshape(self.cams, 9).view(torch.int32), t.reshape(self.cams, 3).view(torch.int32)
        for i, d in enumerate(dest_words_list):
            self.send[i * rw: i * rw + nf].copy_(d, non_blocking=True)
            self.send[i * rw + fp: i * rw + fp + 18].copy_(Rw[i], non_blocking=True)
            self.send[i * rw + fp + 18: (i + 1) * rw].copy_(tw[i], non_blocking=True)

    def pack(self, dest_words, R, t):
        """dest_words: int32[N*5] view of the KLT_TrackedFeature array; R: f64[9]; t: f64[3] (same device)."""
        assert self.cams == 1, "pack() is the one-camera-per-rank form; use pack_group()"
        nf, fp = self.n * FEATURE_WORDS, feature_words_padded(self.n)
        self.send[:nf].copy_(dest_words, non_blocking=True)
        self.send[fp: fp + 18].copy_(R.view(torch.int32), non_blocking=True)
        self.send[fp + 18:].copy_(t.view(torch.int32), non_blocking=True)

    def all_gather(self, stream=None):
        if self._x is not None:
            import ctypes as C

            from ._lib import check

            dests, R, t = self._pending
            arr = (C.c_void_p * len(dests))(*[d.data_ptr() for d in dests])
            s = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
            check(self.native._L.cs_exchange_allgather_dev(self._x, C.c_void_p(s), arr, C.c_void_p(R.data_ptr()),
                                                           C.c_void_p(t.data_ptr())), "cs_exchange_allgather_dev")
            return None
        if self.world == 1:
            self.recv.copy_(self.send, non_blocking=True)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        return self.recv

    def native_records(self, device):
        """(torch uint8 view of the gathered records, record_bytes) of the native path: global camera g at g * record_bytes"""
        import ctypes as C

        p, nb = C.c_void_p(), C.c_size_t(0)
        self.native._L.cs_exchange_buffers(self._x, C.byref(p), C.byref(nb))
        total = nb.value * self.cams * self.world
        return torch.as_tensor(_DevArray(p.value, total, "|u1"), device=torch.device("cuda", device)), nb.value

    def record_ptr(self, cam, device=None):
        """device address of GLOBAL camera `cam`'s gathered record: its N KLT_TrackedFeature first -- what cs_klt_handback_dev takes as
        `dest` for a camera another rank tracks"""
        n1 = self.n // self.cams
        w = record_words(n1)
        if self._x is not None:
            import ctypes as C

            p, nb = C.c_void_p(), C.c_size_t(0)
            self.native._L.cs_exchange_buffers(self._x, C.byref(p), C.byref(nb))
            return p.value + cam * nb.value
        return self.recv.data_ptr() + 4 * w * cam

    def unpack_poses(self, d_R, d_t, stream=None, skip_own=True):
        """every gathered camera's R | t into d_R [world * cams_per_rank, 9] / d_t [.., 3] (torch f64 tensors on this device), on
        `stream`; skip_own: this rank's own cameras are left alone (the pose solve wrote them in place)"""
        if self._x is not None:
            import ctypes as C

            from ._lib import check

            s = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
            check(self.native._L.cs_exchange_unpack_poses_dev(self._x, C.c_void_p(s), C.c_void_p(d_R.data_ptr()), C.c_void_p(d_t.data_ptr()),
                                                              1 if skip_own else 0), "cs_exchange_unpack_poses_dev")
            return
        n1 = self.n // self.cams
        fp, w = feature_words_padded(n1), record_words(n1)
        own = range(self.rank * self.cams, (self.rank + 1) * self.cams) if skip_own else ()
        ctx = torch.cuda.stream(stream) if stream is not None else _nullcontext()
        with ctx:
            rec = self.recv.view(self.world * self.cams, w)
            keep = [g for g in range(self.world * self.cams) if g not in own]
            idx = torch.tensor(keep, device=self.recv.device)
            d_R.view(-1, 9)[idx] = rec[idx, fp: fp + 18].contiguous().view(torch.float64).view(-1, 9)
            d_t.view(-1, 3)[idx] = rec[idx, fp + 18: fp + 24].contiguous().view(torch.float64).view(-1, 3)

    def broadcast(self, ptr, nbytes, root, device, stream=None):
        """`nbytes` of device memory at `ptr` from rank `root` to every rank, in place (collective 3: a packed bundle-adjustment
        result); native: ncclBroadcast on `stream` through the exchange communicator (same stream, same order on every rank as the
        per-frame all-gather); torch path: dist.broadcast of a view of the same memory"""
        if self.world == 1:
            return
        if self._x is not None:
            import ctypes as C

            from ._lib import check

            s = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
            L = self.native._L
            L.cs_comm_broadcast_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            check(L.cs_comm_broadcast_dev(self.native.exchange_comm, C.c_void_p(s), C.c_void_p(ptr), int(nbytes), int(root)),
                  "cs_comm_broadcast_dev")
            return
        t = torch.as_tensor(_DevArray(ptr, nbytes, "|u1"), device=torch.device("cuda", device))
        ctx = torch.cuda.stream(stream) if stream is not None else _nullcontext()
        with ctx:
            dist.broadcast(t, src=root, group=self.group)

    def close(self):
        if self._x is not None:
            self.native._L.cs_exchange_destroy.argtypes = [self.native._C.c_void_p]
            self.native._L.cs_exchange_destroy(self._x)
            self._x = None

    def unpack(self, cam, device=None):
        """-> (features int32[N,5] view, R f64[9], t f64[3]) of GLOBAL camera `cam` (rank cam // cams_per_rank, local index
        cam % cams_per_rank) from the last all_gather; N = the slots of one camera.  Both paths (torch buffers / the native
        exchange's library-owned buffer: pass `device`) use the same per-camera record."""
        n1 = self.n // self.cams
        nf, fp, w = n1 * FEATURE_WORDS, feature_words_padded(n1), record_words(n1)
        if self._x is not None:
            rec8, nb = self.native_records(device if device is not None else self.native.device)
            assert nb == 4 * w
            rec = rec8.view(torch.int32)[cam * w: (cam + 1) * w]
        else:
            rec = self.recv[cam * w: (cam + 1) * w]
        return rec[:nf].view(n1, FEATURE_WORDS), rec[fp: fp + 18].view(torch.float64), rec[fp + 18:].view(torch.float64)


def features_from_words(words):
    """int32[N,5] (host tensor / array) -> numpy structured KLT_TrackedFeature[N]"""
    from .klt import KLT_TrackedFeature

    a = words.cpu().numpy() if hasattr(words, "cpu") else np.asarray(words)
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1).view(KLT_TrackedFeature)


# ---------------------------------------------------------------------------------------------------------------
# Collective 2: joint bundle adjustment with the points sliced by rank.
#
# After the per-frame all-gather every rank holds all measurements and poses, i.e. the whole bundleAdjustRobust
# problem of src/app/SL_CoSLAMRobustBA.cpp:170-180 / SL_InterCamPoseEstimator.cpp:92-95.  Rank r owns the points
# [pLo, pHi): it linearises them and forms ITS part of the reduced camera system; ONE all-reduce of S || rhs
# ((6C')^2 + 6C' doubles: 29 KB for 8 cameras x 1 key frame, 4.1 MB for cfg5) per LM step makes it whole; every rank
# then solves the same system, steps its own points and all cameras, and four scalars (cost, point step, flag change,
# outlier count) are all-reduced so that every rank takes the same LM / outlier decisions.  Points and outlier flags
# are exchanged once at the end.  The phases are device launches on one stream (coslam_amd/csrc/ba.hip,
# cs_ba_dist_*); with backend "nccl" (RCCL over xGMI) the collectives are stream-ordered too, so the host never
# synchronises.
PH_COST0, PH_CONTROL0, PH_LIN_SCHUR, PH_SOLVE_UPDATE, PH_CONTROL1, PH_FLAG, PH_OUTER_END, PH_FINAL_PREP, PH_FINISH = range(9)


def point_slice(rank, world, n_points):
    """Contiguous slice of the point list owned by `rank`."""
    per = (n_points + world - 1) // world
    lo = min(rank * per, n_points)
    return lo, min(lo + per, n_points)


def run_sliced_ba(backends, reduce_fn, max_iter, inner_max_iter):
    """The phase / collective schedule of the sliced BA.

    backends: the rank-local phase engines (ONE in a real job: HipSlicedBA; several only when ranks are emulated in one
    process).  Each exposes phase(ph) and the buffers S_rhs, scal, pts, outlier.  reduce_fn(name) sums buffer `name`
    over all ranks in place (dist.all_reduce in a real job)."""
    def ph(p):
        for b in backends:
            b.phase(p)

    for _ in range(max_iter):
        ph(PH_COST0)
        reduce_fn("scal")
        ph(PH_CONTROL0)
        for _ in range(inner_max_iter):
            ph(PH_LIN_SCHUR)
            reduce_fn("S_rhs")
            ph(PH_SOLVE_UPDATE)
            reduce_fn("scal")
            ph(PH_CONTROL1)
        ph(PH_FLAG)
        reduce_fn("scal")
        ph(PH_OUTER_END)
    ph(PH_FINAL_PREP)
    reduce_fn("pts")
    reduce_fn("outlier")
    ph(PH_FINISH)


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (CUDA array interface, also honoured by ROCm builds)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipSlicedBA:
    """Rank-local engine of the sliced BA on one MI355X: the phases of cs_ba_dist_phase and torch views of the buffers
    the collectives run on.  `ws` is a BAWorkspace holding the (replicated) problem."""

    def __init__(self, ws, stream, d_Rs0, d_Ts0, d_pts0, nCamsCon, nPtsCon, maxErr, innerMaxIter, pLo, pHi, add_lambda,
                 device):
        import ctypes as C

        from ._lib import check

        self.ws, self.stream, self._check, self._C = ws, stream, check, C
        L = ws._L
        vp = C.c_void_p
        check(L.cs_ba_dist_begin(ws._h, vp(stream.cuda_stream), ws.C, ws.P, ws.nObs, vp(d_Rs0), vp(d_Ts0), vp(d_pts0),
                                 int(nCamsCon), int(nPtsCon), C.c_double(maxErr), int(innerMaxIter), int(pLo), int(pHi),
                                 1 if add_lambda else 0), "cs_ba_dist_begin")
        pS, pscal, ppts, pout, nred = vp(), vp(), vp(), vp(), C.c_int(0)
        check(L.cs_ba_dist_buffers(ws._h, C.byref(pS), C.byref(nred), C.byref(pscal), C.byref(ppts), C.byref(pout)),
              "cs_ba_dist_buffers")
        dev = torch.device("cuda", device)
        self.S_rhs = torch.as_tensor(_DevArray(pS.value, max(nred.value, 1), "<f8"), device=dev)
        self.scal = torch.as_tensor(_DevArray(pscal.value, 4, "<f8"), device=dev)
        self.pts = torch.as_tensor(_DevArray(ppts.value, max(3 * ws.P, 1), "<f8"), device=dev)
        self.outlier = torch.as_tensor(_DevArray(pout.value, max(ws.nObs, 1), "<i4"), device=dev)

    def phase(self, ph):
        self._check(self.ws._L.cs_ba_dist_phase(self.ws._h, self._C.c_void_p(self.stream.cuda_stream), int(ph)),
                    "cs_ba_dist_phase")


def bundle_adjust_sliced(ws, stream, d_Rs0, d_Ts0, d_pts0, nCamsCon, nPtsCon, maxErr, maxIter, innerMaxIter, device,
                         group=None, native=None):
    """bundleAdjustRobust over all ranks of `group` (one process per GPU).  Every rank passes the same replicated
    problem (already in `ws`) and receives the same result (ws.download()).  With `native` (a NativeComm) the whole
    schedule -- phases and RCCL all-reduces -- is enqueued by ONE C-ABI call (cs_ba_dist_solve)."""
    if native is not None:
        import ctypes as C

        from ._lib import check

        vp = C.c_void_p
        check(ws._L.cs_ba_dist_solve(ws._h, native.ba_comm, vp(stream.cuda_stream), ws.C, ws.P, ws.nObs, vp(d_Rs0), vp(d_Ts0),
                                     vp(d_pts0), int(nCamsCon), int(nPtsCon), C.c_double(maxErr), int(maxIter),
                                     int(innerMaxIter)), "cs_ba_dist_solve")
        return None
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = point_slice(rank, world, ws.P)
    eng = HipSlicedBA(ws, stream, d_Rs0, d_Ts0, d_pts0, nCamsCon, nPtsCon, maxErr, innerMaxIter, lo, hi, rank == 0, device)

    def reduce_fn(name):
        if world > 1:
            with torch.cuda.stream(stream):
                dist.all_reduce(getattr(eng, name), group=group)

    run_sliced_ba([eng], reduce_fn, maxIter, innerMaxIter)
    return eng
