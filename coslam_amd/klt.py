"""Host-side mirror of the reference's tracker interface over the C-ABI (include/coslam_hip.h).

Names, argument meaning and error behaviour follow V3D_GPU::KLT_SequenceTracker
(reference src/tracking/CGKLT/v3d_gpuklt.h:166-263).  This is plumbing for tests / bench; the C++
drop-in header is include/v3d_gpuklt_hip.h.
"""
import ctypes as C

import numpy as np

from ._lib import CoslamHipError, check, lib

# == KLT_TrackedFeature, v3d_gpuklt.h:166-176
KLT_TrackedFeature = np.dtype([("status", "<i4"), ("pos", "<f4", (2,)), ("gain", "<f4"), ("fed", "<i4")])
assert KLT_TrackedFeature.itemsize == 20


class KLT_SequenceTrackerConfig(C.Structure):
    """== KLT_SequenceTrackerConfig, v3d_gpuklt.h:180-199 (defaults identical)."""

    _fields_ = [
        ("nIterations", C.c_int),
        ("nLevels", C.c_int),
        ("levelSkip", C.c_int),
        ("windowWidth", C.c_int),
        ("trackBorderMargin", C.c_float),
        ("convergenceThreshold", C.c_float),
        ("SSD_Threshold", C.c_float),
        ("trackWithGain", C.c_int),
        ("minDistance", C.c_int),
        ("minCornerness", C.c_float),
        ("detectBorderMargin", C.c_float),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.nIterations = 12
        self.nLevels = 3
        self.levelSkip = 2
        self.windowWidth = 5
        self.trackBorderMargin = 4.0
        self.convergenceThreshold = 0.1
        self.SSD_Threshold = 5000.0
        self.trackWithGain = 0
        self.minDistance = 8
        self.minCornerness = 1000.0
        self.detectBorderMargin = 4.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def coslam_config(**kw):
    """The parameter set CoSLAM itself runs with (src/app/SL_SingleSLAM.cpp:291-298,
    src/app/SL_GlobParam.cpp:28-34, src/gui/MyApp.cpp:210-211)."""
    base = dict(minDistance=8, minCornerness=3000.0, nLevels=6, windowWidth=6, convergenceThreshold=1.0,
                SSD_Threshold=20000.0, trackWithGain=1)
    base.update(kw)
    return KLT_SequenceTrackerConfig(**base)


def _u8(img, W, H):
    a = np.ascontiguousarray(img, dtype=np.uint8)
    if a.size != W * H:
        raise ValueError(f"image has {a.size} bytes, expected {W}x{H}")
    return a


class KLT_SequenceTracker:
    """== V3D_GPU::KLT_SequenceTracker (v3d_gpuklt.h:202-294) on one MI355X."""

    def __init__(self, config, device=0, tap_mode=0):
        self._L = lib()
        self._L.cs_klt_create.restype = C.c_void_p
        self._L.cs_klt_create.argtypes = [C.POINTER(KLT_SequenceTrackerConfig), C.c_int, C.c_int]
        self.config = config
        self._h = self._L.cs_klt_create(C.byref(config), int(device), int(tap_mode))
        if not self._h:
            raise CoslamHipError("cs_klt_create: " + self._L.cs_last_error().decode())
        self._h = C.c_void_p(self._h)
        self.N = 0

    # -- lifetime
    def allocate(self, width, height, nLevels, featuresWidth, featuresHeight, pointListWidth=0, pointListHeight=0):
        check(self._L.cs_klt_allocate(self._h, width, height, nLevels, featuresWidth, featuresHeight,
                                      pointListWidth, pointListHeight), "cs_klt_allocate")
        self.W, self.H, self.L = width, height, nLevels
        self.fw, self.fh = featuresWidth, featuresHeight
        self.N = featuresWidth * featuresHeight

    def deallocate(self):
        check(self._L.cs_klt_deallocate(self._h), "cs_klt_deallocate")

    def close(self):
        if self._h:
            self._L.cs_klt_destroy.argtypes = [C.c_void_p]
            self._L.cs_klt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setters, v3d_gpuklt.h:219-240
    def setBorderMargin(self, m):
        check(self._L.cs_klt_set_border_margin(self._h, C.c_float(m)))

    def setConvergenceThreshold(self, t):
        check(self._L.cs_klt_set_convergence_threshold(self._h, C.c_float(t)))

    def setSSD_Threshold(self, t):
        check(self._L.cs_klt_set_ssd_threshold(self._h, C.c_float(t)))

    # -- reference-shaped host calls: return (count, dest[N])
    def _dest(self):
        return np.zeros(self.N, dtype=KLT_TrackedFeature)

    def detect(self, image, present=None):
        img = _u8(image, self.W, self.H)
        dest, n = self._dest(), C.c_int(0)
        if present is None:
            check(self._L.cs_klt_detect(self._h, img.ctypes.data_as(C.c_void_p), C.byref(n),
                                        dest.ctypes.data_as(C.c_void_p)), "cs_klt_detect")
        else:
            p = np.ascontiguousarray(present, dtype=np.float32).reshape(-1, 3)
            check(self._L.cs_klt_detect_present(self._h, img.ctypes.data_as(C.c_void_p), C.byref(n),
                                                dest.ctypes.data_as(C.c_void_p), p.shape[0],
                                                p.ctypes.data_as(C.c_void_p)), "cs_klt_detect_present")
        return n.value, dest

    def redetect(self, image):
        img = _u8(image, self.W, self.H)
        dest, n = self._dest(), C.c_int(0)
        check(self._L.cs_klt_redetect(self._h, img.ctypes.data_as(C.c_void_p), C.byref(n),
                                      dest.ctypes.data_as(C.c_void_p)), "cs_klt_redetect")
        return n.value, dest

    def track(self, image):
        img = _u8(image, self.W, self.H)
        dest, n = self._dest(), C.c_int(0)
        check(self._L.cs_klt_track(self._h, img.ctypes.data_as(C.c_void_p), C.byref(n),
                                   dest.ctypes.data_as(C.c_void_p)), "cs_klt_track")
        return n.value, dest

    def feedExternFeaturePoints(self, featPts):
        p = np.ascontiguousarray(featPts, dtype=np.float32).reshape(-1, 3)
        ids = np.full(max(p.shape[0], 1), -1, dtype=np.int32)
        n = C.c_int(0)
        check(self._L.cs_klt_feed(self._h, p.shape[0], p.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                  C.byref(n)), "cs_klt_feed")
        return n.value, ids[: n.value].copy()

    def feed_dev(self, npts, d_featPts, d_trackIds, d_nFed):
        """feedExternFeaturePoints with the list, the slots and the count in device memory (asynchronous on the tracker's stream)"""
        check(self._L.cs_klt_feed_dev(self._h, int(npts), C.c_void_p(d_featPts), C.c_void_p(d_trackIds), C.c_void_p(d_nFed)), "cs_klt_feed_dev")

    def advanceFrame(self):
        check(self._L.cs_klt_advance(self._h), "cs_klt_advance")

    # -- device-resident calls (pointers are ints: torch tensor .data_ptr())
    def set_stream(self, stream_ptr):
        check(self._L.cs_klt_set_stream(self._h, C.c_void_p(stream_ptr)), "cs_klt_set_stream")

    def detect_dev(self, d_image, d_dest, d_counts):
        check(self._L.cs_klt_detect_dev(self._h, C.c_void_p(d_image), C.c_void_p(d_dest), C.c_void_p(d_counts)),
              "cs_klt_detect_dev")

    def redetect_dev(self, d_image, d_dest, d_counts):
        check(self._L.cs_klt_redetect_dev(self._h, C.c_void_p(d_image), C.c_void_p(d_dest), C.c_void_p(d_counts)),
              "cs_klt_redetect_dev")

    def track_dev(self, d_image, d_dest, d_counts):
        check(self._L.cs_klt_track_dev(self._h, C.c_void_p(d_image), C.c_void_p(d_dest), C.c_void_p(d_counts)),
              "cs_klt_track_dev")

    def prefetch_dev(self, d_image_next):
        """Call BEFORE this frame's redetect_dev/detect_dev: its detector tail also builds the next frame's pyramid +
        cornerness map (cs_klt_prefetch_dev)."""
        check(self._L.cs_klt_prefetch_dev(self._h, C.c_void_p(d_image_next)), "cs_klt_prefetch_dev")

    def enable_graphs(self, on=True):
        check(self._L.cs_klt_enable_graphs(self._h, 1 if on else 0), "cs_klt_enable_graphs")

    def set_profiling(self, on=True):
        check(self._L.cs_klt_set_profiling(self._h, 1 if on else 0), "cs_klt_set_profiling")

    def get_profile(self):
        us, n, lpf = C.c_double(0), C.c_int(0), C.c_int(0)
        check(self._L.cs_klt_get_profile(self._h, C.byref(us), C.byref(n), C.byref(lpf)), "cs_klt_get_profile")
        return {"tracker_us_total": us.value, "frames": n.value, "launches_per_frame": lpf.value}

    def set_fused(self, on=True):
        check(self._L.cs_klt_set_fused(self._h, 1 if on else 0), "cs_klt_set_fused")

    def debug_probe(self, on=True, read=False):
        """diagnostic: per-slot cycle counters of the persistent gain tracker (uint64[N, 8])"""
        out = np.zeros((self.N, 8), dtype=np.uint64) if read else None
        check(self._L.cs_klt_debug_probe(self._h, 1 if on else 0, out.ctypes.data_as(C.c_void_p) if read else None),
              "cs_klt_debug_probe")
        return out

    def redetect_async(self, image):
        """cs_klt_redetect_async_h: upload + redetect + read-back enqueued, returns at once (GPUKLT::next, first half)"""
        img = _u8(image, self.W, self.H)
        check(self._L.cs_klt_redetect_async_h(self._h, img.ctypes.data_as(C.c_void_p)), "cs_klt_redetect_async_h")

    def fetch(self):
        """cs_klt_fetch: blocks until the outstanding frame's results are on the host -> (count, dest[])"""
        dest, n = self._dest(), C.c_int(0)
        check(self._L.cs_klt_fetch(self._h, C.byref(n), dest.ctypes.data_as(C.c_void_p)), "cs_klt_fetch")
        return n.value, dest

    def set_cu_count(self, n_cus):
        check(self._L.cs_klt_set_cu_count(self._h, int(n_cus)), "cs_klt_set_cu_count")

    def set_xcd_placement(self, on):
        check(self._L.cs_klt_set_xcd_placement(self._h, int(bool(on))), "cs_klt_set_xcd_placement")

    def set_concurrent_handles(self, n):
        check(self._L.cs_klt_set_concurrent_handles(self._h, int(n)), "cs_klt_set_concurrent_handles")

    def synchronize(self):
        check(self._L.cs_klt_synchronize(self._h), "cs_klt_synchronize")

    # -- introspection for parity tests
    def pyramid_texels(self):
        self._L.cs_klt_pyramid_texels.restype = C.c_size_t
        return int(self._L.cs_klt_pyramid_texels(self._h))

    def level_view(self, pyr, level):
        off, w, h = C.c_int64(0), C.c_int(0), C.c_int(0)
        check(self._L.cs_klt_pyramid_level_offset(self._h, level, C.byref(off), C.byref(w), C.byref(h)))
        return pyr.reshape(-1, 4)[off.value: off.value + w.value * h.value].reshape(h.value, w.value, 4)

    def read_pyramid(self, which=1):
        out = np.zeros(self.pyramid_texels() * 4, dtype=np.uint16)
        check(self._L.cs_klt_read_pyramid(self._h, which, out.ctypes.data_as(C.c_void_p)), "cs_klt_read_pyramid")
        return out

    def read_cornerness(self):
        out = np.zeros((self.H, self.W), dtype=np.float32)
        check(self._L.cs_klt_read_cornerness(self._h, out.ctypes.data_as(C.c_void_p)), "cs_klt_read_cornerness")
        return out

    def read_features(self):
        out = np.zeros((self.N, 3), dtype=np.float32)
        check(self._L.cs_klt_read_features(self._h, out.ctypes.data_as(C.c_void_p)), "cs_klt_read_features")
        return out

    def build_pyramid(self, image):
        img = _u8(image, self.W, self.H)
        check(self._L.cs_klt_build_pyramid(self._h, img.ctypes.data_as(C.c_void_p)), "cs_klt_build_pyramid")


class KLT_TrackerGroup:
    """Several KLT_SequenceTracker objects (same device, size and configuration) driven together: the per-frame loop of
    CoSLAM::featureTracking() (reference src/app/SL_CoSLAM.cpp:299-305) as ONE set of launches, the camera being one more
    grid dimension of every kernel (cs_klt_group_*).  Results are bit-identical to driving the trackers one by one."""

    def __init__(self, trackers):
        self._L = lib()
        self.trackers = list(trackers)
        n = len(self.trackers)
        arr = (C.c_void_p * n)(*[t._h for t in self.trackers])
        self._L.cs_klt_group_create.restype = C.c_void_p
        self._L.cs_klt_group_create.argtypes = [C.c_void_p, C.c_int]
        h = self._L.cs_klt_group_create(arr, n)
        if not h:
            raise CoslamHipError("cs_klt_group_create: " + self._L.cs_last_error().decode())
        self._h = C.c_void_p(h)
        self.n = n

    def close(self):
        if self._h:
            self._L.cs_klt_group_destroy.argtypes = [C.c_void_p]
            self._L.cs_klt_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ptrs(self, ptrs):
        if len(ptrs) != self.n:
            raise ValueError(f"expected {self.n} pointers, got {len(ptrs)}")
        return (C.c_void_p * self.n)(*[int(p) for p in ptrs])

    def set_stream(self, stream_ptr):
        check(self._L.cs_klt_group_set_stream(self._h, C.c_void_p(stream_ptr)), "cs_klt_group_set_stream")

    def detect_dev(self, d_images, d_dests, d_counts):
        check(self._L.cs_klt_group_detect_dev(self._h, self._ptrs(d_images), self._ptrs(d_dests), self._ptrs(d_counts)),
              "cs_klt_group_detect_dev")

    def redetect_dev(self, d_images, d_dests, d_counts):
        check(self._L.cs_klt_group_redetect_dev(self._h, self._ptrs(d_images), self._ptrs(d_dests), self._ptrs(d_counts)),
              "cs_klt_group_redetect_dev")

    def track_dev(self, d_images, d_dests, d_counts):
        check(self._L.cs_klt_group_track_dev(self._h, self._ptrs(d_images), self._ptrs(d_dests), self._ptrs(d_counts)),
              "cs_klt_group_track_dev")

    def prefetch_dev(self, d_images_next):
        check(self._L.cs_klt_group_prefetch_dev(self._h, self._ptrs(d_images_next)), "cs_klt_group_prefetch_dev")

    def advanceFrame(self):
        check(self._L.cs_klt_group_advance(self._h), "cs_klt_group_advance")

    def synchronize(self):
        check(self._L.cs_klt_group_synchronize(self._h), "cs_klt_group_synchronize")

    def stage_h(self, h_images):
        """cs_klt_group_stage_h: the n HOST images (addresses; pinned memory for a truly asynchronous copy) of a future frame
        into the next slot of the group's staging ring, on the group's copy stream.  Returns the slot."""
        slot = C.c_int(-1)
        check(self._L.cs_klt_group_stage_h(self._h, self._ptrs(h_images), C.byref(slot)), "cs_klt_group_stage_h")
        return slot.value

    def staged(self, slot):
        """cs_klt_group_staged: the slot's device images (ints); the group's stream waits for the slot's copy"""
        out = (C.c_void_p * self.n)()
        check(self._L.cs_klt_group_staged(self._h, int(slot), out), "cs_klt_group_staged")
        return [int(p) for p in out]
