// hostview.hip -- SURVEY 8f-1, second half: the camera group's frames as HOST records, for an adaptor that fills the
// reference's FeaturePoints / Track2D lists lazily (include/shim/tracking/GPUKLTGroup.h).
//
// GPUKLT::next (reference src/tracking/GPUKLT.cpp:144-161) is redetect + addToFeaturePoints + advanceFrame per camera, and
// addToFeaturePoints (:36-60) heap-allocates a FeaturePoint and a Track2DNode per feature per frame
// (src/slam/SL_FeaturePoints.cpp:81-87, src/tracking/SL_Track2D.h:79-82) whether or not anybody on the host looks at them
// that frame.  A host view keeps the per-frame loop on the device -- ONE set of launches for all cameras (cs_klt_group_*),
// the on-device hand-back (cs_klt_handback_dev: undistorPoint, the out >= W | H drop rule, the Track2D span bookkeeping) --
// and streams what addToFeaturePoints WOULD have appended, per frame and slot {state, undistorted x, y}, into a ring in
// pinned host memory with stores from a kernel: nothing on the host waits.  A consumer that wants the lists calls
// cs_klt_hostview_fetch for the frames it has not seen and replays them; between two such calls the frames cost the host
// nothing but the enqueue.
#include <vector>

#include "cs_common.h"

namespace {

struct HvCam {
    const int* state;
    const double* xy;
};
struct HvArgs {
    int nCams, N;
    int* outState;      // pinned, device-visible: [nCams][N]
    double* outXY;      // [nCams][2 N]
    HvCam cam[16];
};

__global__ __launch_bounds__(256) void k_hostview_pack(HvArgs A) {
    const int c = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.N) return;
    A.outState[(size_t)c * A.N + i] = A.cam[c].state[i];
    A.outXY[(size_t)c * 2 * A.N + i] = A.cam[c].xy[i];
    A.outXY[(size_t)c * 2 * A.N + A.N + i] = A.cam[c].xy[A.N + i];
}

}  // namespace

struct cs_klt_hostview {
    int device, nCams, N, W, H, depth;
    cs_klt_group* grp;
    hipStream_t stream;
    std::vector<cs_handback_cam> hb;
    std::vector<void*> dests, counts, owned;
    int* hState;       // pinned: depth x nCams x N
    double* hXY;       // pinned: depth x nCams x 2N
    std::vector<hipEvent_t> landed;
    std::vector<int> frameOf;   // which frame a ring slot holds (-1: none)
    int newest;
    // the view's own image ring (pinned, pulled by the group's copy kernel): HV_IMG_SLOTS x nCams x W*H; a caller that decodes
    // straight into cs_klt_hostview_image() pays no host copy, and a caller that hands its own buffers may reuse them at once
    unsigned char* hImg;
    hipEvent_t pulled[4];
    bool pulledValid[4];
    int imgSlot;   // slot the NEXT frame's images go to
};
constexpr int HV_IMG_SLOTS = 4;

static void hv_free(cs_klt_hostview* v) {
    if (!v) return;
    for (void* p : v->owned) (void)hipFree(p);
    for (hipEvent_t e : v->landed) (void)hipEventDestroy(e);
    if (v->hState) (void)hipHostFree(v->hState);
    if (v->hXY) (void)hipHostFree(v->hXY);
    for (int q = 0; q < HV_IMG_SLOTS; ++q)
        if (v->pulledValid[q]) (void)hipEventDestroy(v->pulled[q]);
    if (v->hImg) cs_pinned_free(v->hImg);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    if (v->grp) cs_klt_group_destroy(v->grp);
    delete v;
}

extern "C" cs_klt_hostview* cs_klt_hostview_create(int device, cs_klt* const* handles, int nCams, int W, int H, int N, const double* K,
                                                   const double* kud, int depth) {
    if (!handles || nCams < 1 || nCams > 16 || W < 1 || H < 1 || N < 1 || !K || !kud || depth < 2) {
        cs_set_error("cs_klt_hostview_create: bad arguments (1..16 cameras, depth >= 2)");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        cs_set_error("cs_klt_hostview_create: hipSetDevice(%d) failed", device);
        return nullptr;
    }
    cs_klt_hostview* v = new cs_klt_hostview();
    v->device = device, v->nCams = nCams, v->N = N, v->W = W, v->H = H, v->depth = depth, v->newest = -1;
    v->grp = nullptr, v->stream = nullptr, v->hState = nullptr, v->hXY = nullptr, v->hImg = nullptr, v->imgSlot = 0;
    for (int q = 0; q < HV_IMG_SLOTS; ++q) v->pulledValid[q] = false;
    v->grp = cs_klt_group_create(handles, nCams);
    if (!v->grp) {
        hv_free(v);
        return nullptr;
    }
    bool ok = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) == hipSuccess &&
              cs_klt_group_set_stream(v->grp, (void*)v->stream) == CS_OK;
    auto dev = [&](size_t bytes, int fill) -> void* {
        void* p = nullptr;
        if (!ok || hipMalloc(&p, bytes) != hipSuccess) {
            ok = false;
            return nullptr;
        }
        v->owned.push_back(p);
        if (hipMemset(p, fill, bytes) != hipSuccess) ok = false;
        return p;
    };
    v->hb.resize(nCams);
    double* dZero = (double*)dev(24, 0);   // (no map behind this view: slot2map stays -1, mapPts is never indexed)
    for (int c = 0; c < nCams && ok; ++c) {
        cs_handback_cam& q = v->hb[c];
        memset(&q, 0, sizeof(q));
        void* dK = dev(72, 0);
        void* dk = dev(56, 0);
        if (!ok) break;
        ok = hipMemcpy(dK, K + 9 * c, 72, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dk, kud + 7 * c, 56, hipMemcpyHostToDevice) == hipSuccess;
        void* dest = dev(sizeof(cs_klt_feature) * (size_t)N, 0);
        v->dests.push_back(dest), v->counts.push_back(dev(16, 0));
        q.dest = (const cs_klt_feature*)dest, q.K = (const double*)dK, q.kud = (const double*)dk, q.mapPts = dZero;
        q.slot2map = (int*)dev(4 * (size_t)N, 0xff), q.trackSpan = (int*)dev(8 * (size_t)N, 0xff), q.xy = (double*)dev(16 * (size_t)N, 0);
        q.state = (int*)dev(4 * (size_t)N, 0xff);
        q.Ms = (double*)dev(192 * 24, 0), q.ms = (double*)dev(192 * 16, 0), q.sel = (int*)dev(192 * 4, 0), q.npts = (int*)dev(4, 0);
    }
    ok = ok && hipHostMalloc((void**)&v->hState, sizeof(int) * (size_t)depth * nCams * N, hipHostMallocMapped) == hipSuccess &&
         hipHostMalloc((void**)&v->hXY, sizeof(double) * (size_t)depth * nCams * 2 * N, hipHostMallocMapped) == hipSuccess;
    v->frameOf.assign(depth, -1);
    if (ok) {
        v->hImg = (unsigned char*)cs_pinned_alloc((size_t)HV_IMG_SLOTS * nCams * W * H);
        ok = v->hImg != nullptr;
    }
    for (int q = 0; q < HV_IMG_SLOTS && ok; ++q) {
        ok = hipEventCreateWithFlags(&v->pulled[q], hipEventDisableTiming) == hipSuccess;
        v->pulledValid[q] = ok;
    }
    for (int s = 0; s < depth && ok; ++s) {
        hipEvent_t e;
        ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) v->landed.push_back(e);
    }
    if (!ok) {
        cs_set_error("cs_klt_hostview_create: allocation failed (%s)", hipGetErrorString(hipGetLastError()));
        hv_free(v);
        return nullptr;
    }
    return v;
}

extern "C" void cs_klt_hostview_destroy(cs_klt_hostview* v) { hv_free(v); }

// where the NEXT frame's image of camera `cam` is to be written (W*H bytes of pinned memory).  Blocks only if the device has not yet
// pulled the frame that used this buffer HV_IMG_SLOTS frames ago -- the back-pressure that keeps the host at most that far ahead.
extern "C" unsigned char* cs_klt_hostview_image(cs_klt_hostview* v, int cam) {
    if (!v || cam < 0 || cam >= v->nCams) {
        cs_set_error("cs_klt_hostview_image: bad arguments");
        return nullptr;
    }
    (void)hipSetDevice(v->device);
    (void)hipEventSynchronize(v->pulled[v->imgSlot]);   // (an event never recorded is complete)
    return v->hImg + ((size_t)v->imgSlot * v->nCams + cam) * v->W * v->H;
}

// one frame: the images (h_images == NULL: already written into cs_klt_hostview_image(); else copied there, so the caller's buffers are
// free on return) pulled by the device, detect (first) or redetect, advanceFrame, hand-back, the frame's records into the ring.
// Returns as soon as everything is enqueued.
extern "C" int cs_klt_hostview_frame(cs_klt_hostview* v, const unsigned char* const* h_images, int frame, int first) {
    if (!v || frame < 0 || frame <= v->newest) {
        cs_set_error("cs_klt_hostview_frame: bad arguments (frames must increase)");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(v->device));
    const int s = frame % v->depth;
    // the slot's previous tenant must have landed before its storage is rewritten (it has, unless the device is `depth` frames behind)
    if (v->frameOf[s] >= 0) CS_HIP(hipEventSynchronize(v->landed[s]));
    const size_t bytes = (size_t)v->W * v->H;
    const unsigned char* own[16];
    for (int c = 0; c < v->nCams; ++c) own[c] = v->hImg + ((size_t)v->imgSlot * v->nCams + c) * bytes;
    if (h_images) {
        CS_HIP(hipEventSynchronize(v->pulled[v->imgSlot]));
        for (int c = 0; c < v->nCams; ++c) {
            if (!h_images[c]) {
                cs_set_error("cs_klt_hostview_frame: null image of camera %d", c);
                return CS_ERR_INVALID;
            }
            memcpy(const_cast<unsigned char*>(own[c]), h_images[c], bytes);
        }
    }
    int slot = -1;
    int rc = cs_klt_group_stage_h(v->grp, own, &slot);
    if (rc != CS_OK) return rc;
    CS_HIP(hipEventRecord(v->pulled[v->imgSlot], v->stream));
    v->imgSlot = (v->imgSlot + 1) % HV_IMG_SLOTS;
    const void* d_images[16];
    rc = cs_klt_group_staged(v->grp, slot, d_images);
    if (rc != CS_OK) return rc;
    rc = first ? cs_klt_group_detect_dev(v->grp, d_images, v->dests.data(), v->counts.data())
               : cs_klt_group_redetect_dev(v->grp, d_images, v->dests.data(), v->counts.data());
    if (rc != CS_OK) return rc;
    rc = cs_klt_group_advance(v->grp);
    if (rc != CS_OK) return rc;
    rc = cs_klt_handback_dev(v->device, (void*)v->stream, v->nCams, v->hb.data(), v->N, v->W, v->H, 16, 12, 192, frame);
    if (rc != CS_OK) return rc;
    HvArgs A;
    A.nCams = v->nCams, A.N = v->N;
    CS_HIP(hipHostGetDevicePointer((void**)&A.outState, v->hState + (size_t)s * v->nCams * v->N, 0));
    CS_HIP(hipHostGetDevicePointer((void**)&A.outXY, v->hXY + (size_t)s * v->nCams * 2 * v->N, 0));
    for (int c = 0; c < v->nCams; ++c) A.cam[c].state = v->hb[c].state, A.cam[c].xy = v->hb[c].xy;
    hipLaunchKernelGGL(k_hostview_pack, dim3((v->N + 255) / 256, v->nCams), dim3(256), 0, v->stream, A);
    CS_CHECK_LAUNCH();
    CS_HIP(hipEventRecord(v->landed[s], v->stream));
    v->frameOf[s] = frame, v->newest = frame;
    return CS_OK;
}

// the oldest frame still in the ring (frames older than that were overwritten: fetch them earlier or make the ring deeper)
extern "C" int cs_klt_hostview_oldest(const cs_klt_hostview* v) {
    if (!v || v->newest < 0) return -1;
    int o = v->newest;
    for (int s = 0; s < v->depth; ++s)
        if (v->frameOf[s] >= 0 && v->frameOf[s] < o) o = v->frameOf[s];
    return o;
}
extern "C" int cs_klt_hostview_newest(const cs_klt_hostview* v) { return v ? v->newest : -1; }

extern "C" int cs_klt_hostview_fetch(cs_klt_hostview* v, int frame, const int** state, const double** xy) {
    if (!v || frame < 0 || !state || !xy) {
        cs_set_error("cs_klt_hostview_fetch: bad arguments");
        return CS_ERR_INVALID;
    }
    const int s = frame % v->depth;
    if (v->frameOf[s] != frame) {
        cs_set_error("cs_klt_hostview_fetch: frame %d is not in the ring (newest %d, depth %d)", frame, v->newest, v->depth);
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(v->device));
    CS_HIP(hipEventSynchronize(v->landed[s]));
    *state = v->hState + (size_t)s * v->nCams * v->N;
    *xy = v->hXY + (size_t)s * v->nCams * 2 * v->N;
    return CS_OK;
}

extern "C" int cs_klt_hostview_synchronize(cs_klt_hostview* v) {
    if (!v) return CS_ERR_INVALID;
    return cs_klt_group_synchronize(v->grp);
}
