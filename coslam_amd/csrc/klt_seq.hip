// klt_seq.hip -- C-ABI of the KLT sequence tracker (include/coslam_hip.h) and its frame schedule.
//
// Replaces V3D_GPU::KLT_SequenceTracker (src/tracking/CGKLT/v3d_gpuklt.h:202-294,
// v3d_gpuklt.cpp:592-889).  The feature-buffer and pyramid "pointer swaps" of the reference
// (_featuresBuffer0/1/2, _pyrCreator0/1) are modelled one to one, so every call sequence -- including
// the odd ones (track without redetect, feed + advance) -- evolves the same state as the reference.
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#include "klt_internal.h"

// ---- error string ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void cs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- test / diagnostic switches: set through the C-ABI by whoever drives a test or a profile, never read from the environment ----
#include <atomic>
static std::atomic<int> g_dbg[CS_DBG_COUNT] = {{-1}, {-1}, {-1}, {-1}};
int cs_debug_get(int which) { return which >= 0 && which < CS_DBG_COUNT ? g_dbg[which].load(std::memory_order_relaxed) : -1; }

extern "C" {

int cs_version(void) { return 100; }
const char* cs_last_error(void) { return g_err; }
int cs_debug_set(const char* key, int value) {
    static const char* names[CS_DBG_COUNT] = {"ba_syrk", "ba_packed", "ba_graphs", "merge_print"};
    for (int k = 0; k < CS_DBG_COUNT; ++k)
        if (key && !strcmp(key, names[k])) {
            g_dbg[k].store(value, std::memory_order_relaxed);
            return CS_OK;
        }
    cs_set_error("cs_debug_set: unknown switch (ba_syrk, ba_packed, ba_graphs, merge_print)");
    return CS_ERR_INVALID;
}


int cs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// A HIP stream restricted to the CU-mask bits [first_cu, first_cu + n_cus) of `device` (hipExtStreamCreateWithCUMask;
// bit i = XCC i % 8, shader engine (i / 8) % 4, CU (i / 8) / 4: a range is the same few CUs of every XCD).
// The persistent tracker runs every wave in lock-step with its neighbours, so one foreign wave on one of its SIMDs
// slows the whole mesh; pose / BA streams confined to a few CUs, and the tracker stream to the others, keeps them
// apart.  Returns the stream handle (a hipStream_t) or null.
void* cs_stream_create_cu_range(int device, int first_cu, int n_cus) {
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        cs_set_error("cs_stream_create_cu_range: bad device %d", device);
        return nullptr;
    }
    const int total = prop.multiProcessorCount;
    if (first_cu < 0 || n_cus <= 0 || first_cu + n_cus > total) {
        cs_set_error("cs_stream_create_cu_range: range [%d, %d) outside the %d CUs", first_cu, first_cu + n_cus, total);
        return nullptr;
    }
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) {
        cs_set_error("hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    return (void*)s;
}

// Mask bits whose index modulo `period` is (complement == 0) / is not (complement != 0) in [0, take).  Mask bit i is a CU
// of XCC (i % 8) (tools/micro/cu_map.hip), and an XCC whose share of the mask is empty gets ALL its CUs: with period 8, 4
// or 2 this is therefore no partition at all (what an earlier round took for "an interleaved partition that keeps the
// tracker at 91 us" was the whole chip).  Kept for experiments with other periods.
void* cs_stream_create_cu_interleaved(int device, int period, int take, int complement) {
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        cs_set_error("cs_stream_create_cu_interleaved: bad device %d", device);
        return nullptr;
    }
    const int total = prop.multiProcessorCount;
    if (period <= 1 || take <= 0 || take >= period) {
        cs_set_error("cs_stream_create_cu_interleaved: need 0 < take < period");
        return nullptr;
    }
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    int n = 0;
    for (int c = 0; c < total; ++c) {
        const bool in = (c % period) < take;
        if (in != (complement != 0)) {
            mask[c >> 5] |= 1u << (c & 31);
            ++n;
        }
    }
    hipStream_t s = nullptr;
    hipError_t e = (n > 0) ? hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) : hipErrorInvalidValue;
    if (e != hipSuccess) {
        cs_set_error("hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
        return nullptr;
    }
    return (void*)s;
}

int cs_stream_destroy(void* stream) {
    if (stream) CS_HIP(hipStreamDestroy((hipStream_t)stream));
    return CS_OK;
}

void cs_klt_config_default(cs_klt_config* c) {  // v3d_gpuklt.h:181-191
    c->nIterations = 12;
    c->nLevels = 3;
    c->levelSkip = 2;
    c->windowWidth = 5;
    c->trackBorderMargin = 4.0f;
    c->convergenceThreshold = 0.1f;
    c->SSD_Threshold = 5000.0f;
    c->trackWithGain = 0;
    c->minDistance = 8;
    c->minCornerness = 1000.0f;
    c->detectBorderMargin = 4.0f;
}

}  // extern "C"

struct cs_klt {
    cs_klt_config cfg;
    int device, tap_mode;
    bool allocated;
    int W, H, L, fw, fh, plw, plh, N;
    float margin, convThr, ssdThr, detMargin;
    CsPyrLayout lay;
    hipStream_t own_stream, stream;
    uint8_t* d_img;
    cs_texel* d_pyr[3];  // [2]: target of cs_klt_prefetch_dev, allocated on first use
    int p0, p1, p2;
    float* d_fb[3];
    int b0, b1, b2;
    float *d_corner_raw, *d_corner;
    float* d_corner_raw_spare;  // prefetch target, swapped with d_corner_raw when the prefetched frame is consumed
    // frame-front prefetch (cs_klt_prefetch_dev): the next frame's pyramid + cornerness map are built by the SAME two
    // launches as this frame's detector tail (horizontal fusion, klt_detect.hip)
    const void* pf_next_img;  // request: image the next redetect's tail should build the front of
    const void* pf_img;       // image whose front sits in d_pyr[p2] / d_corner_raw_spare
    bool pf_valid;
    CsCand *d_cand, *d_sel;
    int maxCand;
    int* d_rank;
    int* d_ctr;
    cs_klt_feature* d_dest;
    int* d_counts;
    float* d_present;
    int presentCap;
    int* d_feed_ids;         // [presentCap + 1] cs_klt_feed's slots + count
    cs_klt_feature* h_dest;  // pinned
    int* h_counts;           // pinned
    float* h_feat;           // pinned
    uint8_t* h_img;          // pinned staging for the host-pointer entry points (a pageable source is staged by the runtime, ~3x slower)
    // HIP-event timing of the tracker stage (bench.py roofline leg): eager launches only
    bool profiling;
    bool async_pending;           // cs_klt_redetect_async_h enqueued a frame that cs_klt_fetch has not collected yet
    hipEvent_t async_ev;
    int last_tracker_launches;    // persistent launches the last frame's tracker took (a span that is not co-resident is split)
    hipEvent_t ev0, ev1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* ev_pairs;
    // persistent (single-launch) gain tracker
    bool use_fused;
    unsigned long long* d_gran;  // [granRows][N] hand-off granules of the persistent tracker (one row per pass + 1)
    int granRows;
    int* d_err;
    int xcd_placement;            // persistent tracker: 1 = a camera's workgroups on its own XCD(s) (cs_klt_set_xcd_placement)
    int cu_count;                 // compute units the handle's stream may use (cs_klt_set_cu_count; default: all)
    int concurrent;               // handles whose persistent kernels may overlap (cs_klt_set_concurrent_handles; 0: all live ones)
    unsigned long long* d_probe;  // diagnostic cycle counters of the persistent tracker (cs_klt_debug_probe)
    // hipGraph cache for the *_dev entry points: one executable graph per (call, buffer rotation state)
    bool use_graphs;
    struct GraphEntry {
        int mode, b0, b1, b2, p0, p1;
        int live_gen;  // generation of the device's live-handle count the residency decision was captured under
        const void* img;
        void *dest, *counts;
        int post_b0, post_b1, post_b2, post_p0, post_p1;
        hipGraphExec_t exec;
    };
    std::vector<GraphEntry>* graphs;
};

// live handles per device: the persistent tracker needs every one of its waves co-resident, so the budget of
// resident workgroups (8 x 256-thread blocks on each of the 256 CUs, minus a margin) is shared between handles
static std::mutex g_reg_mutex;
static int g_live_handles[64];
static int g_live_gen[64];  // bumped whenever g_live_handles changes: cached graphs captured under another count are stale

// cameras whose frame schedule is issued together (one more grid dimension of every kernel)
struct Span {
    cs_klt* const* k;
    int n;
    hipStream_t stream;
};

#define CS_REQUIRE(cond, msg)      \
    do {                           \
        if (!(cond)) {             \
            cs_set_error(msg);     \
            return CS_ERR_INVALID; \
        }                          \
    } while (0)

static int bind_device(cs_klt* k) {
    CS_HIP(hipSetDevice(k->device));
    return CS_OK;
}

static void drop_graphs(cs_klt* k) {
    if (!k->graphs) return;
    for (auto& g : *k->graphs) (void)hipGraphExecDestroy(g.exec);
    k->graphs->clear();
}

static float* read_buffer(cs_klt* k) {  // readFeatures / readFeaturesAndGain, v3d_gpuklt.cpp:94-97,199-203
    return k->cfg.trackWithGain ? k->d_fb[k->b2] : k->d_fb[k->b1];
}

// ---- frame schedules (all asynchronous on the span's stream) ------------------------------------

static void gain_neighbour_offsets(int fw, int fh, int* n1x, int* n1y) {
    // st0 +- ds0.x / ds0.y are scalar broadcasts (klt_tracker_with_gain.cg:64-67)
    const double rxy = (double)fh / (double)fw, ryx = (double)fw / (double)fh;
    n1x[0] = 1;
    n1x[1] = -1;
    n1x[2] = (int)floor(0.5 + ryx);
    n1x[3] = (int)floor(0.5 - ryx);
    n1y[0] = (int)floor(0.5 + rxy);
    n1y[1] = (int)floor(0.5 - rxy);
    n1y[2] = 1;
    n1y[3] = -1;
}

// How many cameras of this span may share ONE persistent launch of the rows tracker (0: none -- use the per-pass
// schedule)?  Every wave of a launch must be co-resident: the budget is the occupancy the runtime reports for the
// instantiation actually launched (with its dynamic LDS size), scaled to the CUs the stream may use, shared between the
// independent launches that may overlap on the device.  A span with more cameras than fit is issued as several
// persistent launches one after the other on its stream.  (Every spin in the kernel is bounded: a grid that is not
// resident after all raises the handle's error word instead of hanging.)
static int rows_cams_per_persistent_launch(const Span& S, int hw, int T, int* perCuOut = nullptr) {
    cs_klt* k0 = S.k[0];
    if (!k0->use_fused || T < 1 || T + 1 > k0->granRows) return 0;
    int perCu = 0;  // resident waves per CU the runtime reports for this instantiation (VGPR- or LDS-bound here)
    if (cs_rows_max_resident_blocks(hw, k0->device, &perCu) <= 0) return 0;
    if (perCuOut) *perCuOut = perCu;
    const long capacity = (long)perCu * k0->cu_count - k0->cu_count / 8;  // a little slack below the reported occupancy
    int live = 1;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        live = g_live_handles[k0->device & 63] > 0 ? g_live_handles[k0->device & 63] : 1;
    }
    if (k0->concurrent > 0 && k0->concurrent < live) live = k0->concurrent;
    int launches = live - S.n + 1;  // the span's cameras share ONE stream; every other live handle may overlap with it
    if (launches < 1) launches = 1;
    const long perCam = (long)cs_rows_waves(hw, k0->N) * launches;
    long m = capacity / (perCam > 0 ? perCam : 1);
    if (m > S.n) m = S.n;
    return (int)m;
}

// postDest != null: the persistent tracker folds k_post_track into its epilogue; *postFused says whether it did
static int enqueue_tracker(const Span& S, cs_klt_feature* const* postDest, int doSuppress, bool* postFused) {
    if (postFused) *postFused = false;
    cs_klt* k0 = S.k[0];
    const cs_klt_config& c = k0->cfg;
    const int hw = c.windowWidth / 2;
    if (!c.trackWithGain) {
        // the host passes -DNITERATIONS but the shader reads N_ITERATIONS: always 5
        // (v3d_gpuklt.cpp:108 vs klt_tracker.cg:16-18)
        for (int i = 0; i < S.n; ++i) {
            cs_klt* k = S.k[i];
            int rc = cs_launch_track_nogain(k->d_pyr[k->p0], k->d_pyr[k->p1], k->lay, c.levelSkip, hw, 5, k->margin,
                                            k->convThr, k->ssdThr, k->N, k->d_fb[k->b0], k->d_fb[k->b1], S.stream);
            if (rc) return rc;
        }
        return CS_OK;
    }
    int levelSkip = c.levelSkip > 0 ? c.levelSkip : (c.nLevels - 1);  // v3d_gpuklt.h:14
    if (levelSkip <= 0) levelSkip = 1;
    int nLevelsVisited = 0;
    for (int level = k0->L - 1; level >= 0; level -= levelSkip) ++nLevelsVisited;
    const int T = nLevelsVisited * c.nIterations;
    const float delta = 200.0f, lambda = 1.0f;  // v3d_gpuklt.cpp:243-250 (tau = 1: delta stays)

    if (cs_rows_supported(hw)) {
        CsRowsArgs A;
        memset(&A, 0, sizeof(A));
        A.lv.L = k0->L;
        for (int l = 0; l < k0->L; ++l) {
            A.lv.w[l] = k0->lay.w[l];
            A.lv.h[l] = k0->lay.h[l];
            A.lv.off[l] = k0->lay.off[l];
        }
        A.W = k0->W;
        A.H = k0->H;
        A.fw = k0->fw;
        A.fh = k0->fh;
        A.N = k0->N;
        A.nIter = c.nIterations;
        A.levelSkip = levelSkip;
        A.doSuppress = doSuppress;
        A.lambda = lambda;
        A.delta = delta;
        gain_neighbour_offsets(k0->fw, k0->fh, A.n1x, A.n1y);
        A.nCams = S.n;
        const float realConv = k0->convThr * k0->convThr, realSsd = k0->ssdThr;
        const float realVr[4] = {k0->margin / (float)k0->W, k0->margin / (float)k0->H, 1.0f - k0->margin / (float)k0->W,
                                 1.0f - k0->margin / (float)k0->H};
        int perCu = 0;   // resident waves per CU of the instantiation launched
        int perLaunch = rows_cams_per_persistent_launch(S, hw, T, &perCu);
        if (perLaunch >= 1) {
            // a span that needs several launches is cut EVENLY (4 cameras, 3 fit: 2 + 2, not 3 + 1): the launches follow each other on
            // one stream and each lasts about as long as its slowest camera, so nothing is lost -- and 1, 2, 4 or 8 cameras per launch
            // divide the eight XCDs, so the camera-per-XCD placement below applies (cfg5: four XCDs = four L2s per camera instead of
            // eight for a launch of three)
            const int nLaunches = (S.n + perLaunch - 1) / perLaunch;
            perLaunch = (S.n + nLaunches - 1) / nLaunches;
            A.sqrConvThr = realConv;
            A.ssdThr = realSsd;
            for (int q = 0; q < 4; ++q) A.vr[q] = realVr[q];
            k0->last_tracker_launches = (S.n + perLaunch - 1) / perLaunch;
            for (int first = 0; first < S.n; first += perLaunch) {
                const int m = (S.n - first < perLaunch) ? S.n - first : perLaunch;
                A.nCams = m;
                // camera per XCD when the launch's cameras divide the eight XCDs (1, 2, 4, 8 cameras) and a camera's workgroups fit
                // the XCDs they are sent to
                A.xcdsPerCam = 0;
                if (k0->xcd_placement && m <= 8 && 8 % m == 0) {
                    const int q = 8 / m, wgPerCam = (cs_rows_waves(hw, k0->N) + 3) / 4;
                    const long room = (long)(perCu / 4) * (k0->cu_count / 8);
                    if ((long)((wgPerCam + q - 1) / q) <= room) A.xcdsPerCam = q;   // (every camera's last workgroup sits at the end of the
                    // placed dispatch order: ALL of an XCD's share has to be resident at once -- 65 on a 64-slot XCD never finish)
                }
                for (int i = 0; i < m; ++i) {
                    cs_klt* k = S.k[first + i];
                    CsRowsCam& C = A.cam[i];
                    C.pyr0 = k->d_pyr[k->p0];
                    C.pyr1 = k->d_pyr[k->p1];
                    C.feat0 = k->d_fb[k->b2];
                    C.featStart = k->d_fb[k->b0];
                    // where the ping-pong schedule of the reference leaves its last two results (v3d_gpuklt.cpp:281-285)
                    C.outLast = k->d_fb[(T & 1) ? k->b1 : k->b0];
                    C.outPrev = k->d_fb[(T & 1) ? k->b0 : k->b1];
                    C.gran = k->d_gran;
                    C.tagWord = (const unsigned*)(k->d_counts + 5);
                    C.err = k->d_err;
                    C.dest = postDest ? postDest[first + i] : nullptr;
                    C.corner = k->d_corner_raw;
                    C.probe = k->d_probe;
                }
                int rc = cs_launch_track_rows_fused(A, hw, S.stream);
                if (rc) return rc;
            }
            if (postFused) *postFused = (postDest != nullptr);
            for (int i = 0; i < S.n; ++i) {
                cs_klt* k = S.k[i];
                if (T & 1) std::swap(k->b0, k->b1);  // T swaps of (buffer0, buffer1)
                std::swap(k->b0, k->b2);             // v3d_gpuklt.cpp:304
            }
            return CS_OK;
        }
        A.nCams = S.n;
        // one launch per Gauss-Newton pass (v3d_gpuklt.cpp:254-287), all cameras per launch
        for (int i = 0; i < S.n; ++i) {
            int rc = cs_launch_reset_beta(S.k[i]->d_fb[S.k[i]->b0], S.k[i]->N, S.stream);  // v3d_gpuklt.cpp:223-227
            if (rc) return rc;
        }
        for (int level = k0->L - 1; level >= 0; level -= levelSkip) {  // :254
            A.level = level;
            for (int iter = 1; iter <= c.nIterations; ++iter) {  // :268
                if (iter == 1) {  // :271-279
                    A.sqrConvThr = 1000000.0f;
                    A.ssdThr = 1000000.0f;
                    A.vr[0] = A.vr[1] = -1.0f;
                    A.vr[2] = A.vr[3] = 2.0f;
                } else if (iter == c.nIterations) {
                    A.sqrConvThr = realConv;
                    A.ssdThr = realSsd;
                    for (int q = 0; q < 4; ++q) A.vr[q] = realVr[q];
                }
                for (int i = 0; i < S.n; ++i) {
                    cs_klt* k = S.k[i];
                    CsRowsCam& C = A.cam[i];
                    C.pyr0 = k->d_pyr[k->p0];
                    C.pyr1 = k->d_pyr[k->p1];
                    C.feat0 = k->d_fb[k->b2];
                    C.featStart = k->d_fb[k->b0];
                    C.outLast = k->d_fb[k->b1];
                }
                int rc = cs_launch_track_rows_pass(A, hw, S.stream);
                if (rc) return rc;
                for (int i = 0; i < S.n; ++i) std::swap(S.k[i]->b0, S.k[i]->b1);  // :285
            }
        }
        for (int i = 0; i < S.n; ++i) std::swap(S.k[i]->b0, S.k[i]->b2);  // :304
        return CS_OK;
    }

    // window sizes the rows design does not cover: one wave per feature, one launch per pass, camera by camera
    for (int i = 0; i < S.n; ++i) {
        cs_klt* k = S.k[i];
        const cs_texel *P0 = k->d_pyr[k->p0], *P1 = k->d_pyr[k->p1];
        int rc = cs_launch_reset_beta(k->d_fb[k->b0], k->N, S.stream);  // v3d_gpuklt.cpp:223-227
        if (rc) return rc;
        CsGainPassArgs a;
        memset(&a, 0, sizeof(a));
        a.whx = (float)k->W;
        a.why = (float)k->H;
        a.fw = k->fw;
        a.fh = k->fh;
        a.N = k->N;
        a.hw = hw;
        a.lambda = lambda;  // :250
        a.delta = delta;
        gain_neighbour_offsets(k->fw, k->fh, a.n1x, a.n1y);
        a.sqrConvThr = 1000000.0f;
        a.ssdThr = 1000000.0f;
        a.vr[0] = a.vr[1] = -1.0f;
        a.vr[2] = a.vr[3] = 2.0f;
        for (int level = k->L - 1; level >= 0; level -= levelSkip) {  // :254
            a.lvl0 = P0 + k->lay.off[level];
            a.lvl1 = P1 + k->lay.off[level];
            a.Wl = k->lay.w[level];
            a.Hl = k->lay.h[level];
            for (int iter = 1; iter <= c.nIterations; ++iter) {  // :268
                if (iter == 1) {  // :271-279
                    a.sqrConvThr = 1000000.0f;
                    a.ssdThr = 1000000.0f;
                    a.vr[0] = a.vr[1] = -1.0f;
                    a.vr[2] = a.vr[3] = 2.0f;
                } else if (iter == c.nIterations) {
                    a.sqrConvThr = k->convThr * k->convThr;
                    a.ssdThr = k->ssdThr;
                    a.vr[0] = k->margin / (float)k->W;
                    a.vr[1] = k->margin / (float)k->H;
                    a.vr[2] = 1.0f - k->margin / (float)k->W;
                    a.vr[3] = 1.0f - k->margin / (float)k->H;
                }
                a.feat0 = k->d_fb[k->b2];
                a.featIn = k->d_fb[k->b0];
                a.featOut = k->d_fb[k->b1];
                rc = cs_launch_track_gain_pass(a, S.stream);
                if (rc) return rc;
                std::swap(k->b0, k->b1);  // :285
            }
        }
        std::swap(k->b0, k->b2);  // :304
    }
    return CS_OK;
}

static int enqueue_detect_tail(const Span& S, int mode, const int* nPresentGiven, const int* maxKeepFixed,
                               cs_klt_feature* const* d_dest, int* const* d_counts) {
    cs_klt* k0 = S.k[0];
    // pending cs_klt_prefetch_dev requests ride in the tail's two launches (all cameras or none); inside a graph
    // capture they are dropped
    bool withNext = true;
    for (int i = 0; i < S.n; ++i) withNext = withNext && (S.k[i]->pf_next_img != nullptr);
    if (withNext) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(S.stream, &cap);
        if (cap != hipStreamCaptureStatusNone || cs_nonmax_lds_bytes(k0->cfg.minDistance) > 64 * 1024) withNext = false;
    }
    CsNonmaxCam nm[CS_MAX_CAMS];
    CsSelectCam sel[CS_MAX_CAMS];
    CsFrontCam nxt[CS_MAX_CAMS];
    for (int i = 0; i < S.n; ++i) {
        cs_klt* k = S.k[i];
        nm[i].in = k->d_corner_raw;
        nm[i].out = k->d_corner;
        nm[i].cand = k->d_cand;
        nm[i].ctr = k->d_ctr;
        sel[i].cand = k->d_cand;
        sel[i].rankM = k->d_rank;
        sel[i].sel = k->d_sel;
        sel[i].maxKeepFixed = maxKeepFixed ? maxKeepFixed[i] : -1;
        CsFillArgs& f = sel[i].fill;
        f.mode = mode;
        f.N = k->N;
        f.withGain = k->cfg.trackWithGain;
        f.nPresentGiven = nPresentGiven ? nPresentGiven[i] : 0;
        f.present3 = k->d_present;
        f.sel = k->d_sel;
        f.ctr = k->d_ctr;
        f.dest = d_dest[i];
        // provideFeatures / provideFeaturesAndGain, v3d_gpuklt.cpp:86-92,188-197
        f.list_a = k->d_fb[k->b1];
        f.list_b = k->cfg.trackWithGain ? k->d_fb[k->b2] : nullptr;
        f.counts = d_counts[i];
        f.tagWord = (unsigned*)(k->d_counts + 5);
        // spare buffers: last read by the tracker / non-max of the frame BEFORE this one -- older than this point of
        // the stream
        nxt[i].img = (const uint8_t*)k->pf_next_img;
        nxt[i].pyr = k->d_pyr[k->p2];
        nxt[i].corner = k->d_corner_raw_spare;
        nxt[i].ctr = nullptr;
        k->pf_next_img = nullptr;
    }
    if (withNext) {
        int rc = cs_launch_tail_with_next_front(nm, sel, nxt, S.n, k0->W, k0->H, k0->cfg.minDistance, k0->maxCand,
                                                k0->plw * k0->plh, k0->lay, k0->tap_mode, k0->cfg.minCornerness,
                                                k0->detMargin, S.stream);
        if (rc) return rc;
        for (int i = 0; i < S.n; ++i) {
            S.k[i]->pf_img = nxt[i].img;
            S.k[i]->pf_valid = true;
        }
        return CS_OK;
    }
    int rc = cs_launch_nonmax_compact(nm, S.n, k0->W, k0->H, k0->cfg.minDistance, k0->maxCand, S.stream);
    if (rc) return rc;
    return cs_launch_select_fill(sel, S.n, k0->maxCand, k0->plw * k0->plh, S.stream);
}

static int enqueue_front(const Span& S, const uint8_t* const* d_img, bool withCorner) {
    cs_klt* k0 = S.k[0];
    CsFrontCam fc[CS_MAX_CAMS];
    for (int i = 0; i < S.n; ++i) {
        cs_klt* k = S.k[i];
        fc[i].img = d_img[i];
        fc[i].pyr = k->d_pyr[k->p1];
        fc[i].corner = withCorner ? k->d_corner_raw : nullptr;
        fc[i].ctr = k->d_ctr;
    }
    return cs_launch_frame_front(fc, S.n, k0->lay, k0->tap_mode, withCorner, k0->cfg.minCornerness, k0->detMargin, S.stream);
}

static int enqueue_track(const Span& S, const uint8_t* const* d_img, cs_klt_feature* const* d_dest, int* const* d_counts,
                         bool forRedetect) {
    cs_klt* k0 = S.k[0];
    // pyramid (:858), the cornerness map the detector will need, and the zeroing of this frame's counters: two launches
    // (klt_pyramid.hip) -- unless the previous frame's detector tail has already built them (cs_klt_prefetch_dev)
    int rc = CS_OK;
    bool prefetched = true;
    for (int i = 0; i < S.n; ++i) prefetched = prefetched && S.k[i]->pf_valid && S.k[i]->pf_img == (const void*)d_img[i];
    if (prefetched) {
        // the spare pyramid / cornerness buffers become this frame's, the ones they replace (last read two frames ago)
        // become the next prefetch's targets.  The candidate counter was zeroed by the previous frame's tail.
        for (int i = 0; i < S.n; ++i) {
            std::swap(S.k[i]->p1, S.k[i]->p2);
            std::swap(S.k[i]->d_corner_raw, S.k[i]->d_corner_raw_spare);
        }
    } else {
        rc = enqueue_front(S, d_img, forRedetect);
        if (rc) return rc;
    }
    for (int i = 0; i < S.n; ++i) S.k[i]->pf_valid = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (k0->profiling) {
        CS_HIP(hipEventCreate(&e0));
        CS_HIP(hipEventCreate(&e1));
        CS_HIP(hipEventRecord(e0, S.stream));
    }
    bool postFused = false;
    rc = enqueue_tracker(S, d_dest, forRedetect ? 1 : 0, &postFused);
    if (rc) return rc;
    if (k0->profiling) {
        CS_HIP(hipEventRecord(e1, S.stream));
        k0->ev_pairs->push_back(std::make_pair(e0, e1));
    }
    for (int i = 0; i < S.n; ++i) {
        cs_klt* k = S.k[i];
        if (!postFused) {
            rc = cs_launch_post_track(k->cfg.trackWithGain ? k->d_fb[k->b2] : k->d_fb[k->b1], k->N, d_dest[i], k->d_ctr,
                                      k->d_corner_raw, k->W, k->H, forRedetect ? 1 : 0, S.stream);
            if (rc) return rc;
        }
        if (!forRedetect) {
            k->pf_next_img = nullptr;  // no detector tail to carry the next front: the request lapses
            rc = cs_launch_counts_track(d_dest[i], k->N, d_counts[i], k->d_ctr, (unsigned*)(k->d_counts + 5), S.stream);
            if (rc) return rc;
        }
    }
    return rc;
}

static int enqueue_redetect(const Span& S, const uint8_t* const* d_img, cs_klt_feature* const* d_dest, int* const* d_counts) {
    int rc = enqueue_track(S, d_img, d_dest, d_counts, true);
    if (rc) return rc;
    return enqueue_detect_tail(S, 2, nullptr, nullptr, d_dest, d_counts);
}

static int enqueue_detect(const Span& S, const uint8_t* const* d_img, cs_klt_feature* const* d_dest, int* const* d_counts,
                          const int* nPresent) {
    for (int i = 0; i < S.n; ++i) S.k[i]->pf_valid = false;
    int rc = enqueue_front(S, d_img, true);
    if (rc) return rc;
    int maxKeep[CS_MAX_CAMS], nPres[CS_MAX_CAMS];
    bool anyPresent = false;
    for (int i = 0; i < S.n; ++i) {
        cs_klt* k = S.k[i];
        nPres[i] = nPresent ? nPresent[i] : 0;
        anyPresent = anyPresent || nPres[i] > 0;
        if (nPres[i] > 0) {
            rc = cs_launch_suppress_list(k->d_corner_raw, k->W, k->H, nPres[i], k->d_present, S.stream);
            if (rc) return rc;
        }
        rc = cs_launch_clear_dest(d_dest[i], k->N, S.stream);
        if (rc) return rc;
        maxKeep[i] = k->N - nPres[i];
        if (maxKeep[i] < 0) maxKeep[i] = 0;
    }
    return enqueue_detect_tail(S, anyPresent ? 1 : 0, nPres, maxKeep, d_dest, d_counts);
}

// single-camera forms used by the per-handle entry points
static int enqueue_detect1(cs_klt* k, const uint8_t* img, cs_klt_feature* dest, int* counts, int nPresent) {
    Span S = {&k, 1, k->stream};
    return enqueue_detect(S, &img, &dest, &counts, &nPresent);
}
static int enqueue_redetect1(cs_klt* k, const uint8_t* img, cs_klt_feature* dest, int* counts) {
    Span S = {&k, 1, k->stream};
    return enqueue_redetect(S, &img, &dest, &counts);
}
static int enqueue_track1(cs_klt* k, const uint8_t* img, cs_klt_feature* dest, int* counts) {
    Span S = {&k, 1, k->stream};
    return enqueue_track(S, &img, &dest, &counts, false);
}

// the persistent tracker raises *d_err when a bounded spin ran out (waves not co-resident): results are invalid
static int check_device_error(cs_klt* k) {
    int e = 0;
    CS_HIP(hipMemcpy(&e, k->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (e) {
        (void)hipMemset(k->d_err, 0, sizeof(int));
        cs_set_error("persistent KLT tracker timed out waiting for a neighbour (grid not co-resident); "
                     "call cs_klt_set_fused(k, 0) to use the one-launch-per-pass schedule");
        return CS_ERR_HIP;
    }
    return CS_OK;
}

static int fetch_results(cs_klt* k, int* count, cs_klt_feature* dest) {
    // dest[], the counts and the persistent tracker's error word come back in ONE copy behind one synchronisation
    CS_HIP(hipMemcpyAsync(k->h_dest, k->d_dest, sizeof(cs_klt_feature) * k->N + 8 * sizeof(int), hipMemcpyDeviceToHost,
                          k->stream));
    CS_HIP(hipStreamSynchronize(k->stream));
    memcpy(dest, k->h_dest, sizeof(cs_klt_feature) * k->N);
    const int* tail = (const int*)(k->h_dest + k->N);
    *count = tail[0];
    if (tail[4]) return check_device_error(k);
    return CS_OK;
}

// device-side pull of a host image (pinned, device-visible) into the staging ring: 16-byte loads over PCIe, grid-stride
__global__ __launch_bounds__(256) void k_stage_pull(uint8_t* dst, const uint8_t* src, size_t nbytes) {
    const size_t n16 = nbytes / 16, t0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v* s4 = (const u4v*)src;
    u4v* d4 = (u4v*)dst;
    __builtin_amdgcn_s_setprio(0);  // (a background copy: never ahead of the tracker's or the solves' waves)
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        size_t q = t0;
        for (; q + 3 * stride < n16; q += 4 * stride) {  // four 16-byte PCIe reads in flight per lane
            const u4v a = __builtin_nontemporal_load(s4 + q), b = __builtin_nontemporal_load(s4 + q + stride),
                      c = __builtin_nontemporal_load(s4 + q + 2 * stride), d = __builtin_nontemporal_load(s4 + q + 3 * stride);
            d4[q] = a, d4[q + stride] = b, d4[q + 2 * stride] = c, d4[q + 3 * stride] = d;
        }
        for (; q < n16; q += stride) d4[q] = __builtin_nontemporal_load(s4 + q);
        for (size_t q = n16 * 16 + t0; q < nbytes; q += stride) dst[q] = src[q];
    } else {
        for (size_t q = t0; q < nbytes; q += stride) dst[q] = src[q];
    }
}

static int upload_image(cs_klt* k, const uint8_t* image) {
    memcpy(k->h_img, image, (size_t)k->W * k->H);
    CS_HIP(hipMemcpyAsync(k->d_img, k->h_img, (size_t)k->W * k->H, hipMemcpyHostToDevice, k->stream));
    return CS_OK;
}

extern "C" {

cs_klt* cs_klt_create(const cs_klt_config* cfg, int device, int tap_mode) {
    if (!cfg) {
        cs_set_error("cs_klt_create: null config");
        return nullptr;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device < 0 || device >= n) {
        cs_set_error("cs_klt_create: no usable HIP device %d (count %d, %s); there is no CPU fallback", device, n,
                     e == hipSuccess ? "ok" : hipGetErrorString(e));
        return nullptr;
    }
    cs_klt* k = new (std::nothrow) cs_klt();
    if (!k) return nullptr;
    memset(k, 0, sizeof(*k));
    k->cfg = *cfg;
    k->device = device;
    k->tap_mode = tap_mode;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&k->own_stream, hipStreamNonBlocking) != hipSuccess) {
        cs_set_error("cs_klt_create: cannot create a stream on device %d", device);
        delete k;
        return nullptr;
    }
    k->stream = k->own_stream;
    {
        hipDeviceProp_t prop;
        k->cu_count = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    k->graphs = new std::vector<cs_klt::GraphEntry>();
    k->ev_pairs = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
    k->use_fused = true;   // (cs_klt_set_fused selects the per-pass schedule)
    k->xcd_placement = 1;  // (cs_klt_set_xcd_placement)
    return k;
}

int cs_klt_deallocate(cs_klt* k) {
    CS_REQUIRE(k, "null handle");
    if (!k->allocated) return CS_OK;
    int rc = bind_device(k);
    if (rc) return rc;
    hipStreamSynchronize(k->stream);
    drop_graphs(k);
    hipFree(k->d_img);
    hipFree(k->d_pyr[0]);
    hipFree(k->d_pyr[1]);
    if (k->d_pyr[2]) hipFree(k->d_pyr[2]);
    k->d_pyr[2] = nullptr;
    if (k->d_corner_raw_spare) hipFree(k->d_corner_raw_spare);
    k->d_corner_raw_spare = nullptr;
    for (int i = 0; i < 3; ++i) hipFree(k->d_fb[i]);
    hipFree(k->d_corner_raw);
    hipFree(k->d_corner);
    hipFree(k->d_cand);
    hipFree(k->d_sel);
    hipFree(k->d_rank);
    hipFree(k->d_ctr);
    hipFree(k->d_dest);
    hipFree(k->d_present);
    hipFree(k->d_feed_ids);
    hipFree(k->d_gran);
    if (k->d_probe) hipFree(k->d_probe);
    k->d_probe = nullptr;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        g_live_handles[k->device & 63]--;
        g_live_gen[k->device & 63]++;
    }
    hipHostFree(k->h_dest);
    hipHostFree(k->h_counts);
    hipHostFree(k->h_feat);
    hipHostFree(k->h_img);
    k->allocated = false;
    return CS_OK;
}

void cs_klt_destroy(cs_klt* k) {
    if (!k) return;
    cs_klt_deallocate(k);
    delete k->graphs;
    k->graphs = nullptr;
    for (auto& p : *k->ev_pairs) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    delete k->ev_pairs;
    k->ev_pairs = nullptr;
    hipSetDevice(k->device);
    hipStreamDestroy(k->own_stream);
    delete k;
}

int cs_klt_allocate(cs_klt* k, int W, int H, int nLevels, int fw, int fh, int plw, int plh) {
    CS_REQUIRE(k, "null handle");
    CS_REQUIRE(!k->allocated, "cs_klt_allocate: already allocated (the reference warns on double allocate)");
    CS_REQUIRE(W >= 16 && H >= 16 && nLevels >= 1 && nLevels <= CS_MAX_LEVELS, "cs_klt_allocate: bad image size / level count");
    CS_REQUIRE((W >> (nLevels - 1)) >= 2 && (H >> (nLevels - 1)) >= 2, "cs_klt_allocate: too many levels for this size");
    CS_REQUIRE(fw >= 1 && fh >= 1, "cs_klt_allocate: bad feature grid");
    CS_REQUIRE(k->cfg.minDistance >= 0 && k->cfg.windowWidth >= 1, "cs_klt_allocate: bad minDistance/windowWidth");
    if (plw <= 0 || plh <= 0) {  // v3d_gpuklt.h:213-215
        plw = 2 * fw;
        plh = 2 * fh;
    }
    int rc = bind_device(k);
    if (rc) return rc;
    k->W = W;
    k->H = H;
    k->L = nLevels;
    k->fw = fw;
    k->fh = fh;
    k->plw = plw;
    k->plh = plh;
    k->N = fw * fh;
    // v3d_gpuklt.cpp:603-619; the detector margin stays at its constructor value 10 (v3d_gpuklt.h:114)
    k->margin = k->cfg.trackBorderMargin;
    k->convThr = k->cfg.convergenceThreshold;
    k->ssdThr = k->cfg.SSD_Threshold;
    k->detMargin = 10.0f;
    k->lay = cs_make_layout(W, H, nLevels);
    k->p0 = 0;
    k->p1 = 1;
    k->p2 = 2;
    k->d_pyr[2] = nullptr;
    k->d_corner_raw_spare = nullptr;
    k->pf_valid = false;
    k->pf_img = k->pf_next_img = nullptr;
    k->b0 = 0;
    k->b1 = 1;
    k->b2 = 2;
    const int d = k->cfg.minDistance;
    k->maxCand = ((W + d) / (d + 1)) * ((H + d) / (d + 1));  // one strict maximum per (d+1)^2 block at most
    if (k->maxCand < 256) k->maxCand = 256;
    k->presentCap = k->N > 4096 ? k->N : 4096;
    rc = cs_nonmax_prepare(d);
    if (rc) return rc;

    CS_HIP(hipMalloc((void**)&k->d_img, (size_t)W * H));
    for (int i = 0; i < 2; ++i) {
        CS_HIP(hipMalloc((void**)&k->d_pyr[i], k->lay.texels * sizeof(cs_texel)));
        CS_HIP(hipMemsetAsync(k->d_pyr[i], 0, k->lay.texels * sizeof(cs_texel), k->stream));
    }
    for (int i = 0; i < 3; ++i) {
        CS_HIP(hipMalloc((void**)&k->d_fb[i], sizeof(float) * 3 * k->N));
    }
    CS_HIP(hipMalloc((void**)&k->d_corner_raw, sizeof(float) * (size_t)W * H));
    CS_HIP(hipMalloc((void**)&k->d_corner, sizeof(float) * (size_t)W * H));
    CS_HIP(hipMemsetAsync(k->d_corner, 0, sizeof(float) * (size_t)W * H, k->stream));
    CS_HIP(hipMalloc((void**)&k->d_cand, sizeof(CsCand) * k->maxCand));
    CS_HIP(hipMalloc((void**)&k->d_sel, sizeof(CsCand) * k->maxCand));
    CS_HIP(hipMalloc((void**)&k->d_rank, sizeof(int) * k->maxCand));
    CS_HIP(hipMalloc((void**)&k->d_ctr, sizeof(int) * 8));
    // dest[] || counts[4] || error word in ONE allocation: the host-pointer entry points read all of it back with one copy
    CS_HIP(hipMalloc((void**)&k->d_dest, sizeof(cs_klt_feature) * k->N + 8 * sizeof(int)));
    k->d_counts = (int*)(k->d_dest + k->N);
    k->d_err = k->d_counts + 4;
    CS_HIP(hipMalloc((void**)&k->d_present, sizeof(float) * 3 * k->presentCap));
    CS_HIP(hipMalloc((void**)&k->d_feed_ids, sizeof(int) * ((size_t)k->presentCap + 1)));
    {
        // one granule row per Gauss-Newton pass of the with-gain schedule (+ the initial row): levels visited x iterations
        int visited = 0;
        int skip = k->cfg.levelSkip > 0 ? k->cfg.levelSkip : (k->cfg.nLevels - 1);  // as enqueue_tracker
        if (skip <= 0) skip = 1;
        for (int l = nLevels - 1; l >= 0; l -= skip) ++visited;
        k->granRows = visited * (k->cfg.nIterations > 0 ? k->cfg.nIterations : 0) + 1;
        CS_HIP(hipMalloc((void**)&k->d_gran, sizeof(unsigned long long) * (size_t)k->granRows * k->N));
        CS_HIP(hipMemsetAsync(k->d_gran, 0, sizeof(unsigned long long) * (size_t)k->granRows * k->N, k->stream));
    }
    CS_HIP(hipMemsetAsync(k->d_err, 0, 4 * sizeof(int), k->stream));  // error word, frame tag of the hand-off granules, spare
    CS_HIP(hipHostMalloc((void**)&k->h_dest, sizeof(cs_klt_feature) * k->N + 8 * sizeof(int), hipHostMallocDefault));
    CS_HIP(hipHostMalloc((void**)&k->h_counts, sizeof(int) * 8, hipHostMallocDefault));
    CS_HIP(hipHostMalloc((void**)&k->h_feat, sizeof(float) * (4 * (size_t)k->presentCap + 4), hipHostMallocDefault));   // (+ cs_klt_feed's slots)
    CS_HIP(hipHostMalloc((void**)&k->h_img, (size_t)W * H, hipHostMallocDefault));
    // RTT buffers start undefined in the reference; we define every slot dead
    for (int i = 0; i < 3 * k->N; ++i) k->h_feat[i] = -1.0f;
    for (int i = 0; i < 3; ++i) {
        CS_HIP(hipMemcpyAsync(k->d_fb[i], k->h_feat, sizeof(float) * 3 * k->N, hipMemcpyHostToDevice, k->stream));
    }
    CS_HIP(hipMemsetAsync(k->d_dest, 0xff, sizeof(cs_klt_feature) * k->N, k->stream));  // status = fed = -1
    CS_HIP(hipMemsetAsync(k->d_counts, 0, sizeof(int) * 4, k->stream));
    CS_HIP(hipStreamSynchronize(k->stream));
    k->allocated = true;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        g_live_handles[k->device & 63]++;
        g_live_gen[k->device & 63]++;
    }
    return CS_OK;
}

int cs_klt_set_border_margin(cs_klt* k, float m) {  // v3d_gpuklt.h:219-226
    CS_REQUIRE(k, "null handle");
    if (k->allocated && k->graphs && !k->graphs->empty()) {  // the thresholds are baked into captured kernel arguments
        if (bind_device(k) == CS_OK) (void)hipStreamSynchronize(k->stream);
        drop_graphs(k);
    }
    k->margin = m;
    k->detMargin = m;
    return CS_OK;
}
int cs_klt_set_convergence_threshold(cs_klt* k, float t) {
    CS_REQUIRE(k, "null handle");
    if (k->allocated && k->graphs && !k->graphs->empty()) {  // the thresholds are baked into captured kernel arguments
        if (bind_device(k) == CS_OK) (void)hipStreamSynchronize(k->stream);
        drop_graphs(k);
    }
    k->convThr = t;
    return CS_OK;
}
int cs_klt_set_ssd_threshold(cs_klt* k, float t) {
    CS_REQUIRE(k, "null handle");
    if (k->allocated && k->graphs && !k->graphs->empty()) {  // the thresholds are baked into captured kernel arguments
        if (bind_device(k) == CS_OK) (void)hipStreamSynchronize(k->stream);
        drop_graphs(k);
    }
    k->ssdThr = t;
    return CS_OK;
}

int cs_klt_set_stream(cs_klt* k, void* s) {
    CS_REQUIRE(k, "null handle");
    k->stream = s ? (hipStream_t)s : k->own_stream;
    return CS_OK;
}

// The persistent tracker needs every wave co-resident: tell the handle how many compute units its stream may use
// when that stream carries a CU mask (cs_stream_create_cu_range); the default is the whole device.
int cs_klt_set_cu_count(cs_klt* k, int n_cus) {
    CS_REQUIRE(k && n_cus > 0, "cs_klt_set_cu_count: bad arguments");
    k->cu_count = n_cus;
    if (k->allocated) {
        int rc = bind_device(k);
        if (rc) return rc;
        CS_HIP(hipStreamSynchronize(k->stream));
        drop_graphs(k);
    }
    return CS_OK;
}

// Placement of the persistent tracker's workgroups (speed only; results are bit-identical either way).
int cs_klt_set_xcd_placement(cs_klt* k, int on) {
    CS_REQUIRE(k, "cs_klt_set_xcd_placement: null handle");
    k->xcd_placement = on ? 1 : 0;
    if (k->allocated) {
        CS_HIP(hipStreamSynchronize(k->stream));
        drop_graphs(k);
    }
    return CS_OK;
}

// Many cameras on one GPU: the co-residency budget of the persistent tracker is shared between the handles whose
// launches can overlap.  By default that is every live handle of the device; a caller that serialises cameras (e.g.
// eight cameras on three streams) states the real concurrency here.
int cs_klt_set_concurrent_handles(cs_klt* k, int n) {
    CS_REQUIRE(k && n >= 0, "cs_klt_set_concurrent_handles: bad arguments");
    k->concurrent = n;
    if (k->allocated) {
        int rc = bind_device(k);
        if (rc) return rc;
        CS_HIP(hipStreamSynchronize(k->stream));
        drop_graphs(k);
    }
    return CS_OK;
}

int cs_klt_synchronize(cs_klt* k) {
    CS_REQUIRE(k && k->allocated, "not allocated");
    int rc = bind_device(k);
    if (rc) return rc;
    CS_HIP(hipStreamSynchronize(k->stream));
    return check_device_error(k);
}

// HIP-event timing of the tracker stage on the handle's stream.  While on, the *_dev calls run eagerly (no graph
// replay) and every track/redetect brackets the tracker launch(es) with an event pair.
int cs_klt_set_profiling(cs_klt* k, int on) {
    CS_REQUIRE(k, "null handle");
    k->profiling = on != 0;
    return CS_OK;
}

// sum of the bracketed tracker times (microseconds), number of frames and kernel launches per frame; clears the log
int cs_klt_get_profile(cs_klt* k, double* tracker_us, int* n_frames, int* launches_per_frame) {
    CS_REQUIRE(k && k->allocated && tracker_us && n_frames, "cs_klt_get_profile: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    CS_HIP(hipStreamSynchronize(k->stream));
    double total = 0;
    for (auto& p : *k->ev_pairs) {
        float ms = 0;
        CS_HIP(hipEventElapsedTime(&ms, p.first, p.second));
        total += (double)ms * 1e3;
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    *tracker_us = total;
    *n_frames = (int)k->ev_pairs->size();
    k->ev_pairs->clear();
    if (launches_per_frame) {
        const cs_klt_config& c = k->cfg;
        int skip = c.levelSkip > 0 ? c.levelSkip : (c.nLevels - 1);
        if (skip <= 0) skip = 1;
        int lv = 0;
        for (int level = k->L - 1; level >= 0; level -= skip) ++lv;
        *launches_per_frame = !c.trackWithGain ? 1 : (k->use_fused ? (k->last_tracker_launches > 0 ? k->last_tracker_launches : 1)
                                                                     : lv * c.nIterations + 1);
    }
    return CS_OK;
}

int cs_klt_set_fused(cs_klt* k, int on) {
    CS_REQUIRE(k, "null handle");
    if (k->allocated) {
        int rc = bind_device(k);
        if (rc) return rc;
        CS_HIP(hipStreamSynchronize(k->stream));
        drop_graphs(k);
    }
    k->use_fused = on != 0;
    return CS_OK;
}

// diagnostic: per-slot cycle counters of the persistent gain tracker {texel wait, arithmetic, hand-off wait, solve +
// publish, polls, total, start, XCC id}; eager launches only (drops the graph cache)
int cs_klt_debug_probe(cs_klt* k, int on, unsigned long long* host_out8) {
    CS_REQUIRE(k && k->allocated, "not allocated");
    int rc = bind_device(k);
    if (rc) return rc;
    CS_HIP(hipStreamSynchronize(k->stream));
    drop_graphs(k);
    if (host_out8 && k->d_probe)
        CS_HIP(hipMemcpy(host_out8, k->d_probe, sizeof(unsigned long long) * 8 * k->N, hipMemcpyDeviceToHost));
    if (on && !k->d_probe) {
        CS_HIP(hipMalloc((void**)&k->d_probe, sizeof(unsigned long long) * 8 * k->N));
        CS_HIP(hipMemset(k->d_probe, 0, sizeof(unsigned long long) * 8 * k->N));
    } else if (!on && k->d_probe) {
        CS_HIP(hipFree(k->d_probe));
        k->d_probe = nullptr;
    }
    return CS_OK;
}

int cs_klt_advance(cs_klt* k) {  // v3d_gpuklt.h:252-259
    CS_REQUIRE(k && k->allocated, "not allocated");
    std::swap(k->b0, k->b1);
    std::swap(k->p0, k->p1);
    return CS_OK;
}

// ---- device-resident entry points ----------------------------------------------------------------
// The frame schedule is five launches (klt_pyramid.hip x2, klt_track.hip, klt_detect.hip x2); it can be issued
// eagerly (default) or replayed from a hipGraph (cs_klt_enable_graphs; one host launch per frame).
// The three feature buffers and two pyramids rotate from call to call (period <= 6), so the cache is keyed by
// the rotation state; the graph reads the image from the handle's own staging buffer.
static int run_dev(cs_klt* k, int mode, const void* d_image, void* d_dest, void* d_counts) {
    auto enqueue = [&](const uint8_t* img) -> int {
        if (mode == 0) return enqueue_detect1(k, img, (cs_klt_feature*)d_dest, (int*)d_counts, 0);
        if (mode == 1) return enqueue_redetect1(k, img, (cs_klt_feature*)d_dest, (int*)d_counts);
        return enqueue_track1(k, img, (cs_klt_feature*)d_dest, (int*)d_counts);
    };
    if (!k->use_graphs || k->profiling) return enqueue((const uint8_t*)d_image);
    CS_HIP(hipMemcpyAsync(k->d_img, d_image, (size_t)k->W * k->H, hipMemcpyDeviceToDevice, k->stream));
    int liveGen = 0;
    {
        std::lock_guard<std::mutex> lg(g_reg_mutex);
        liveGen = g_live_gen[k->device & 63];
    }
    for (auto& g : *k->graphs) {
        if (g.live_gen != liveGen) continue;  // captured under another live-handle count: its residency decision is stale
        if (g.mode == mode && g.b0 == k->b0 && g.b1 == k->b1 && g.b2 == k->b2 && g.p0 == k->p0 && g.p1 == k->p1 &&
            g.dest == d_dest && g.counts == d_counts) {
            CS_HIP(hipGraphLaunch(g.exec, k->stream));
            k->b0 = g.post_b0;
            k->b1 = g.post_b1;
            k->b2 = g.post_b2;
            k->p0 = g.post_p0;
            k->p1 = g.post_p1;
            return CS_OK;
        }
    }
    cs_klt::GraphEntry g;
    g.mode = mode;
    g.b0 = k->b0;
    g.b1 = k->b1;
    g.b2 = k->b2;
    g.p0 = k->p0;
    g.p1 = k->p1;
    g.live_gen = liveGen;
    g.img = k->d_img;
    g.dest = d_dest;
    g.counts = d_counts;
    CS_HIP(hipStreamBeginCapture(k->stream, hipStreamCaptureModeRelaxed));
    int rc = enqueue(k->d_img);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(k->stream, &graph);
    if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    if (e != hipSuccess) {
        cs_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return CS_ERR_HIP;
    }
    e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        cs_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return CS_ERR_HIP;
    }
    g.post_b0 = k->b0;
    g.post_b1 = k->b1;
    g.post_b2 = k->b2;
    g.post_p0 = k->p0;
    g.post_p1 = k->p1;
    k->graphs->push_back(g);
    CS_HIP(hipGraphLaunch(g.exec, k->stream));
    return CS_OK;
}

int cs_klt_enable_graphs(cs_klt* k, int on) {
    CS_REQUIRE(k, "null handle");
    k->use_graphs = on != 0;
    if (!on && k->allocated) {
        int rc = bind_device(k);
        if (rc) return rc;
        CS_HIP(hipStreamSynchronize(k->stream));
        drop_graphs(k);
    }
    return CS_OK;
}

// Frame-front prefetch.  The pyramid and cornerness map of the NEXT frame depend only on its image, so the detector
// tail of the CURRENT frame (two small launches that leave the chip mostly idle) can build them on the side: call
// cs_klt_prefetch_dev(next image) BEFORE the cs_klt_redetect_dev / cs_klt_detect_dev of the current frame; that call's
// tail then carries the next front in the same two launches (third pyramid buffer, second cornerness map), and the
// cs_klt_redetect_dev / cs_klt_track_dev that follows with the same image pointer starts at the tracker.
// Results are identical with and without; an unconsumed or mismatched prefetch is ignored.
int cs_klt_prefetch_dev(cs_klt* k, const void* d_image_next) {
    CS_REQUIRE(k && k->allocated, "cs_klt_prefetch_dev: not allocated");
    if (k->use_graphs || !d_image_next) {  // the captured schedule has its front inside
        k->pf_next_img = nullptr;
        return CS_OK;
    }
    int rc = bind_device(k);
    if (rc) return rc;
    if (!k->d_pyr[2]) {
        CS_HIP(hipMalloc((void**)&k->d_pyr[2], k->lay.texels * sizeof(cs_texel)));
        CS_HIP(hipMalloc((void**)&k->d_corner_raw_spare, sizeof(float) * (size_t)k->W * k->H));
        CS_HIP(hipMemsetAsync(k->d_pyr[2], 0, k->lay.texels * sizeof(cs_texel), k->stream));
    }
    k->pf_next_img = d_image_next;
    return CS_OK;
}

int cs_klt_detect_dev(cs_klt* k, const void* d_image, void* d_dest, void* d_counts) {
    CS_REQUIRE(k && k->allocated && d_image && d_dest && d_counts, "cs_klt_detect_dev: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    return run_dev(k, 0, d_image, d_dest, d_counts);
}
int cs_klt_redetect_dev(cs_klt* k, const void* d_image, void* d_dest, void* d_counts) {
    CS_REQUIRE(k && k->allocated && d_image && d_dest && d_counts, "cs_klt_redetect_dev: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    return run_dev(k, 1, d_image, d_dest, d_counts);
}
int cs_klt_track_dev(cs_klt* k, const void* d_image, void* d_dest, void* d_counts) {
    CS_REQUIRE(k && k->allocated && d_image && d_dest && d_counts, "cs_klt_track_dev: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    return run_dev(k, 2, d_image, d_dest, d_counts);
}

// ---- reference-shaped host entry points ---------------------------------------------------------
int cs_klt_detect(cs_klt* k, const uint8_t* image, int* nDetected, cs_klt_feature* dest) {
    CS_REQUIRE(k && k->allocated && image && nDetected && dest, "cs_klt_detect: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    if ((rc = upload_image(k, image))) return rc;
    if ((rc = enqueue_detect1(k, k->d_img, k->d_dest, k->d_counts, 0))) return rc;
    return fetch_results(k, nDetected, dest);
}

int cs_klt_detect_present(cs_klt* k, const uint8_t* image, int* nDetected, cs_klt_feature* dest, int nPresent,
                          const float* present) {
    CS_REQUIRE(k && k->allocated && image && nDetected && dest, "cs_klt_detect_present: bad arguments");
    CS_REQUIRE(nPresent >= 0 && nPresent <= k->presentCap && (nPresent == 0 || present), "cs_klt_detect_present: bad present list");
    int rc = bind_device(k);
    if (rc) return rc;
    if ((rc = upload_image(k, image))) return rc;
    if (nPresent > 0) {
        memcpy(k->h_feat, present, sizeof(float) * 3 * nPresent);
        CS_HIP(hipMemcpyAsync(k->d_present, k->h_feat, sizeof(float) * 3 * nPresent, hipMemcpyHostToDevice, k->stream));
    }
    if ((rc = enqueue_detect1(k, k->d_img, k->d_dest, k->d_counts, nPresent))) return rc;
    return fetch_results(k, nDetected, dest);
}

int cs_klt_redetect(cs_klt* k, const uint8_t* image, int* nNew, cs_klt_feature* dest) {
    CS_REQUIRE(k && k->allocated && image && nNew && dest, "cs_klt_redetect: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    if ((rc = upload_image(k, image))) return rc;
    if ((rc = enqueue_redetect1(k, k->d_img, k->d_dest, k->d_counts))) return rc;
    return fetch_results(k, nNew, dest);
}

// GPUKLT::next in two halves: everything of the frame -- upload, redetect, the copy of dest[] back -- is enqueued and the call
// returns; cs_klt_fetch blocks until that frame's results are on the host.  Several cameras' frames are then in flight
// together (one handle each), where the synchronous form serialises upload -> kernels -> read-back per camera (155-178 us per
// camera-frame, profiles/r02_dropin_cxx_latency.txt).  One frame per handle may be outstanding.
int cs_klt_redetect_async_h(cs_klt* k, const uint8_t* image) {
    CS_REQUIRE(k && k->allocated && image, "cs_klt_redetect_async_h: bad arguments");
    CS_REQUIRE(!k->async_pending, "cs_klt_redetect_async_h: the previous frame has not been fetched (cs_klt_fetch)");
    int rc = bind_device(k);
    if (rc) return rc;
    if ((rc = upload_image(k, image))) return rc;
    if ((rc = enqueue_redetect1(k, k->d_img, k->d_dest, k->d_counts))) return rc;
    CS_HIP(hipMemcpyAsync(k->h_dest, k->d_dest, sizeof(cs_klt_feature) * k->N + 8 * sizeof(int), hipMemcpyDeviceToHost, k->stream));
    if (!k->async_ev) CS_HIP(hipEventCreateWithFlags(&k->async_ev, hipEventDisableTiming));
    CS_HIP(hipEventRecord(k->async_ev, k->stream));
    k->async_pending = true;
    return CS_OK;
}

int cs_klt_fetch(cs_klt* k, int* nNew, cs_klt_feature* dest) {
    CS_REQUIRE(k && k->allocated && nNew && dest, "cs_klt_fetch: bad arguments");
    CS_REQUIRE(k->async_pending, "cs_klt_fetch: no frame outstanding (cs_klt_redetect_async_h)");
    int rc = bind_device(k);
    if (rc) return rc;
    CS_HIP(hipEventSynchronize(k->async_ev));
    k->async_pending = false;
    memcpy(dest, k->h_dest, sizeof(cs_klt_feature) * k->N);
    const int* tail = (const int*)(k->h_dest + k->N);
    *nNew = tail[0];
    if (tail[4]) return check_device_error(k);
    return CS_OK;
}

int cs_klt_track(cs_klt* k, const uint8_t* image, int* nPresent, cs_klt_feature* dest) {
    CS_REQUIRE(k && k->allocated && image && nPresent && dest, "cs_klt_track: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    if ((rc = upload_image(k, image))) return rc;
    if ((rc = enqueue_track1(k, k->d_img, k->d_dest, k->d_counts))) return rc;
    return fetch_results(k, nPresent, dest);
}

// KLT_SequenceTracker::feedExternFeaturePoints (v3d_gpuklt.cpp:808-855) as one workgroup: every slot of the feature list against
// the fed points (a slot within sqrt(1e-4) of one is freed -- the distance loop reads the stride-3 point list with stride 2, as
// shipped, :826-827), then the free slots in slot order take the points in their order (an ordered rank over the list: ballots
// per wave, the waves' totals through LDS) -- position from the stride-3 list, gain 1 -- and trackIds[k] = the slot point k took.
// The list is read where readFeatures(AndGain) reads it and written where provideFeatures(AndGain) writes it.
__global__ __launch_bounds__(1024) void k_feed_extern(int N, int npts, const float* __restrict__ pts, const float* in, float* outA, float* outB,
                                                      int* __restrict__ trackIds, int* __restrict__ nFed) {
    __shared__ int wTot[16];
    __shared__ int sBase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) sBase = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += 1024) {
        const int i = i0 + tid;
        float x = -1.0f, y = 0.0f, g = 0.0f;
        bool isFree = false;
        if (i < N) {
            x = in[3 * i], y = in[3 * i + 1], g = in[3 * i + 2];
            if (!(x < 0)) {
                const double radius2 = 1e-4;
                for (int q = 0; q < npts; ++q) {
                    const double dx = (double)(pts[2 * q] - x), dy = (double)(pts[2 * q + 1] - y);
                    if (dx * dx + dy * dy < radius2) {
                        x = -1.0f;
                        break;   // (a freed slot is skipped by every later point, :823-824)
                    }
                }
            }
            isFree = x < 0;
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(isFree);
        if (lane == 0) wTot[wv] = __popcll(bal);
        __syncthreads();
        int r = sBase;
        for (int w = 0; w < wv; ++w) r += wTot[w];
        r += __popcll(bal & ((1ull << lane) - 1ull));
        if (i < N) {
            if (isFree && r < npts) {
                x = pts[3 * r], y = pts[3 * r + 1], g = 1.0f;
                trackIds[r] = i;
            }
            outA[3 * i] = x, outA[3 * i + 1] = y, outA[3 * i + 2] = g;
            if (outB) outB[3 * i] = x, outB[3 * i + 1] = y, outB[3 * i + 2] = g;
        }
        __syncthreads();
        if (tid == 0) {
            int t = sBase;
            for (int w = 0; w < 16; ++w) t += wTot[w];
            sBase = t;
        }
        __syncthreads();
    }
    if (tid == 0) *nFed = sBase < npts ? sBase : npts;
}

int cs_klt_feed_dev(cs_klt* k, int npts, const float* d_featPts, int* d_trackIds, int* d_nFed) {
    CS_REQUIRE(k && k->allocated && d_nFed && npts >= 0 && (npts == 0 || (d_featPts && d_trackIds)), "cs_klt_feed_dev: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    hipLaunchKernelGGL(k_feed_extern, dim3(1), dim3(1024), 0, k->stream, k->N, npts, d_featPts, read_buffer(k), k->d_fb[k->b1],
                       k->cfg.trackWithGain ? k->d_fb[k->b2] : nullptr, d_trackIds, d_nFed);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int cs_klt_feed(cs_klt* k, int npts, const float* featPts, int* trackIds, int* nFed) {  // v3d_gpuklt.cpp:808-855
    CS_REQUIRE(k && k->allocated && nFed && npts >= 0 && (npts == 0 || (featPts && trackIds)), "cs_klt_feed: bad arguments");
    CS_REQUIRE(npts <= k->presentCap, "cs_klt_feed: more points than the tracker's staging holds");
    int rc = bind_device(k);
    if (rc) return rc;
    // the points up, the kernel, the slots and the count back: the host form of cs_klt_feed_dev
    if (npts > 0) {
        memcpy(k->h_feat, featPts, sizeof(float) * 3 * npts);
        CS_HIP(hipMemcpyAsync(k->d_present, k->h_feat, sizeof(float) * 3 * npts, hipMemcpyHostToDevice, k->stream));
    }
    if ((rc = cs_klt_feed_dev(k, npts, k->d_present, k->d_feed_ids, k->d_feed_ids + k->presentCap))) return rc;
    int* h = (int*)(k->h_feat + 3 * (size_t)k->presentCap);   // (the staging's tail: count, then the slots)
    CS_HIP(hipMemcpyAsync(h, k->d_feed_ids + k->presentCap, sizeof(int), hipMemcpyDeviceToHost, k->stream));
    if (npts > 0) CS_HIP(hipMemcpyAsync(h + 1, k->d_feed_ids, sizeof(int) * npts, hipMemcpyDeviceToHost, k->stream));
    CS_HIP(hipStreamSynchronize(k->stream));
    *nFed = h[0];
    for (int q = 0; q < h[0]; ++q) trackIds[q] = h[1 + q];
    return CS_OK;
}

// ---- introspection ------------------------------------------------------------------------------
size_t cs_klt_pyramid_texels(const cs_klt* k) { return (k && k->allocated) ? k->lay.texels : 0; }

int cs_klt_pyramid_level_offset(const cs_klt* k, int level, int64_t* off, int* w, int* h) {
    CS_REQUIRE(k && k->allocated && level >= 0 && level < k->L, "bad level");
    if (off) *off = k->lay.off[level];
    if (w) *w = k->lay.w[level];
    if (h) *h = k->lay.h[level];
    return CS_OK;
}

int cs_klt_read_pyramid(cs_klt* k, int which, uint16_t* out) {
    CS_REQUIRE(k && k->allocated && out && (which == 0 || which == 1), "cs_klt_read_pyramid: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    CS_HIP(hipStreamSynchronize(k->stream));
    CS_HIP(hipMemcpy(out, k->d_pyr[which ? k->p1 : k->p0], k->lay.texels * sizeof(cs_texel), hipMemcpyDeviceToHost));
    return CS_OK;
}

int cs_klt_read_cornerness(cs_klt* k, float* out) {
    CS_REQUIRE(k && k->allocated && out, "cs_klt_read_cornerness: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    CS_HIP(hipStreamSynchronize(k->stream));
    CS_HIP(hipMemcpy(out, k->d_corner, sizeof(float) * (size_t)k->W * k->H, hipMemcpyDeviceToHost));
    return CS_OK;
}

int cs_klt_read_features(cs_klt* k, float* out) {
    CS_REQUIRE(k && k->allocated && out, "cs_klt_read_features: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    CS_HIP(hipStreamSynchronize(k->stream));
    CS_HIP(hipMemcpy(out, read_buffer(k), sizeof(float) * 3 * k->N, hipMemcpyDeviceToHost));
    return CS_OK;
}

int cs_klt_build_pyramid(cs_klt* k, const uint8_t* image) {
    CS_REQUIRE(k && k->allocated && image, "cs_klt_build_pyramid: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    if ((rc = upload_image(k, image))) return rc;
    CsFrontCam fc = {k->d_img, k->d_pyr[k->p1], nullptr, nullptr};
    if ((rc = cs_launch_frame_front(&fc, 1, k->lay, k->tap_mode, false, 0.0f, 0.0f, k->stream))) return rc;
    CS_HIP(hipStreamSynchronize(k->stream));
    return CS_OK;
}

// ---- camera groups: the frame schedule of several cameras in one set of launches -----------------------------------
// CoSLAM::featureTracking() (src/app/SL_CoSLAM.cpp:299-305) calls GPUKLT::next camera by camera; a group issues the
// same per-camera work with the camera as one more grid dimension of every kernel: 3-5 launches per frame for ALL
// cameras, and one persistent tracker launch whose waves (8 features each) are all co-resident.
constexpr int CS_STAGE_SLOTS = 3;
struct cs_klt_group {
    std::vector<cs_klt*> ks;
    hipStream_t stream;
    // host images into the device off the frame's critical path (cs_klt_group_stage_h): a ring of CS_STAGE_SLOTS x n images,
    // filled by a pull kernel on the group's own stream, in order with the frames (see cs_klt_group_stage_h)
    uint8_t* d_stage = nullptr;
    int next_slot = 0;
};

static bool same_setup(const cs_klt* a, const cs_klt* b) {
    return a->device == b->device && a->tap_mode == b->tap_mode && a->W == b->W && a->H == b->H && a->L == b->L &&
           a->fw == b->fw && a->fh == b->fh && a->plw == b->plw && a->plh == b->plh && a->margin == b->margin &&
           a->convThr == b->convThr && a->ssdThr == b->ssdThr && a->detMargin == b->detMargin &&
           memcmp(&a->cfg, &b->cfg, sizeof(cs_klt_config)) == 0;
}

cs_klt_group* cs_klt_group_create(cs_klt* const* handles, int n) {
    if (!handles || n < 1 || n > CS_MAX_CAMS) {
        cs_set_error("cs_klt_group_create: need 1..%d handles", CS_MAX_CAMS);
        return nullptr;
    }
    for (int i = 0; i < n; ++i) {
        if (!handles[i] || !handles[i]->allocated) {
            cs_set_error("cs_klt_group_create: handle %d is null or not allocated", i);
            return nullptr;
        }
        for (int j = 0; j < i; ++j)
            if (handles[j] == handles[i]) {
                cs_set_error("cs_klt_group_create: handle %d appears twice", i);
                return nullptr;
            }
        if (!same_setup(handles[0], handles[i])) {
            cs_set_error("cs_klt_group_create: handle %d differs from handle 0 in device, size or configuration", i);
            return nullptr;
        }
    }
    cs_klt_group* g = new (std::nothrow) cs_klt_group();
    if (!g) return nullptr;
    g->ks.assign(handles, handles + n);
    g->stream = handles[0]->stream;
    return g;
}

void cs_klt_group_destroy(cs_klt_group* g) {
    if (!g) return;
    if (g->d_stage) {
        (void)hipSetDevice(g->ks[0]->device);
        (void)hipStreamSynchronize(g->stream);
        (void)hipFree(g->d_stage);
    }
    delete g;
}

// ---- host images into the device, asynchronously -----------------------------------------------------------------------
// GPUKLT::next(const unsigned char*) (reference src/tracking/GPUKLT.cpp:144-161) uploads the frame and then tracks it: the
// copy sits on the frame's critical path.  Here the n host images of a FUTURE frame go into the next slot of a small ring,
// enqueued on the group's stream ahead of the frames that use them; the slot's device pointers are then handed to
// cs_klt_group_prefetch_dev / _redetect_dev like any device image.  Pinned host memory (cs_pinned_alloc) makes the copies
// truly asynchronous; pageable memory works, the runtime then stages it (and blocks the caller for the duration).
int cs_klt_group_stage_h(cs_klt_group* g, const unsigned char* const* h_images, int* slot) {
    CS_REQUIRE(g && h_images && slot, "cs_klt_group_stage_h: bad arguments");
    cs_klt* k0 = g->ks[0];
    int rc = bind_device(k0);
    if (rc) return rc;
    const size_t bytes = (size_t)k0->W * k0->H, n = g->ks.size();
    if (!g->d_stage) CS_HIP(hipMalloc((void**)&g->d_stage, bytes * n * CS_STAGE_SLOTS));
    const int q = g->next_slot;
    g->next_slot = (q + 1) % CS_STAGE_SLOTS;
    // The pull runs on the group's OWN stream, in order with the frames: 57 us in front of the next tracker launch, on a stream
    // that has slack -- measured 0.88-0.91 of the resident-image frame rate, against 0.84 with the pull on a second stream beside
    // the tracker (two more cross-stream events per frame cost more than the serialisation; profiles/r03_upload.txt).  Whatever
    // read this slot's previous images was enqueued on the same stream before this call.
    hipStream_t cs = g->stream;
    bool contiguous = true;  // the capture side wrote the n images back to back (one pinned ring entry per frame): ONE copy
    for (size_t i = 0; i < n; ++i) {
        if (!h_images[i]) {
            cs_set_error("cs_klt_group_stage_h: null image of camera %d", (int)i);
            return CS_ERR_INVALID;
        }
        if (i > 0 && h_images[i] != h_images[i - 1] + bytes) contiguous = false;
    }
    // hipMemcpyAsync from pinned memory is NOT asynchronous for the caller here: the runtime writes through the PCIe aperture
    // on the calling thread (measured: 160-280 us of host time per 2.4 MB call, tools/stage_time.py) -- a frame loop that
    // enqueues 0.4 ms frames cannot afford that.  Pinned (device-visible) memory is therefore PULLED by a small copy kernel on
    // the copy stream (a kernel launch for the host, PCIe reads for the device); pageable memory takes hipMemcpyAsync.
    auto one = [&](uint8_t* dst, const unsigned char* src, size_t nbytes) -> int {
        void* dsrc = nullptr;
        if (hipHostGetDevicePointer(&dsrc, (void*)src, 0) == hipSuccess && dsrc) {
            const int blocks = 16;  // a few CUs
            // are plenty for a PCIe-bound pull (~90 KB in flight saturate the link); the tracker and the solves keep the rest
            hipLaunchKernelGGL(k_stage_pull, dim3(blocks), dim3(256), 0, cs, dst, (const uint8_t*)dsrc, nbytes);
            CS_CHECK_LAUNCH();
        } else {
            (void)hipGetLastError();
            CS_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, cs));
        }
        return CS_OK;
    };
    if (contiguous) {
        if ((rc = one(g->d_stage + (size_t)q * n * bytes, h_images[0], bytes * n))) return rc;
    } else {
        for (size_t i = 0; i < n; ++i)
            if ((rc = one(g->d_stage + ((size_t)q * n + i) * bytes, h_images[i], bytes))) return rc;
    }
    *slot = q;
    return CS_OK;
}

// the device images of a staged slot (the pull is ahead of every later use on the group's stream: nothing to wait for)
int cs_klt_group_staged(cs_klt_group* g, int slot, const void** d_images) {
    CS_REQUIRE(g && d_images && slot >= 0 && slot < CS_STAGE_SLOTS && g->d_stage, "cs_klt_group_staged: bad arguments (stage first)");
    cs_klt* k0 = g->ks[0];
    int rc = bind_device(k0);
    if (rc) return rc;
    const size_t bytes = (size_t)k0->W * k0->H, n = g->ks.size();
    for (size_t i = 0; i < n; ++i) d_images[i] = g->d_stage + ((size_t)slot * n + i) * bytes;
    return CS_OK;
}

void* cs_pinned_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        cs_set_error("cs_pinned_alloc: hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}
void cs_pinned_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int cs_klt_group_size(const cs_klt_group* g) { return g ? (int)g->ks.size() : 0; }

int cs_klt_group_set_stream(cs_klt_group* g, void* hip_stream) {
    CS_REQUIRE(g, "null group");
    g->stream = hip_stream ? (hipStream_t)hip_stream : g->ks[0]->own_stream;
    for (cs_klt* k : g->ks) k->stream = g->stream;
    return CS_OK;
}

static int group_check(cs_klt_group* g, const void* const* a, void* const* b, void* const* c, const char* what) {
    if (!g || !a || !b || !c) {
        cs_set_error("%s: bad arguments", what);
        return CS_ERR_INVALID;
    }
    const int n = (int)g->ks.size();
    for (int i = 0; i < n; ++i) {
        if (!a[i] || !b[i] || !c[i] || !g->ks[i]->allocated) {
            cs_set_error("%s: null pointer or deallocated handle for camera %d", what, i);
            return CS_ERR_INVALID;
        }
        if (!same_setup(g->ks[0], g->ks[i])) {
            cs_set_error("%s: camera %d no longer matches camera 0 (thresholds changed on one handle only?)", what, i);
            return CS_ERR_INVALID;
        }
    }
    return bind_device(g->ks[0]);
}

static int group_run(cs_klt_group* g, int mode, const void* const* d_images, void* const* d_dests, void* const* d_counts,
                     const char* what) {
    int rc = group_check(g, d_images, d_dests, d_counts, what);
    if (rc) return rc;
    const int n = (int)g->ks.size();
    Span S = {g->ks.data(), n, g->stream};
    const uint8_t* img[CS_MAX_CAMS];
    cs_klt_feature* dest[CS_MAX_CAMS];
    int* counts[CS_MAX_CAMS];
    for (int i = 0; i < n; ++i) {
        img[i] = (const uint8_t*)d_images[i];
        dest[i] = (cs_klt_feature*)d_dests[i];
        counts[i] = (int*)d_counts[i];
    }
    if (mode == 0) return enqueue_detect(S, img, dest, counts, nullptr);
    if (mode == 1) return enqueue_redetect(S, img, dest, counts);
    return enqueue_track(S, img, dest, counts, false);
}

int cs_klt_group_detect_dev(cs_klt_group* g, const void* const* d_images, void* const* d_dests, void* const* d_counts) {
    return group_run(g, 0, d_images, d_dests, d_counts, "cs_klt_group_detect_dev");
}
int cs_klt_group_redetect_dev(cs_klt_group* g, const void* const* d_images, void* const* d_dests, void* const* d_counts) {
    return group_run(g, 1, d_images, d_dests, d_counts, "cs_klt_group_redetect_dev");
}
int cs_klt_group_track_dev(cs_klt_group* g, const void* const* d_images, void* const* d_dests, void* const* d_counts) {
    return group_run(g, 2, d_images, d_dests, d_counts, "cs_klt_group_track_dev");
}

int cs_klt_group_prefetch_dev(cs_klt_group* g, const void* const* d_images_next) {
    CS_REQUIRE(g && d_images_next, "cs_klt_group_prefetch_dev: bad arguments");
    for (size_t i = 0; i < g->ks.size(); ++i) {
        int rc = cs_klt_prefetch_dev(g->ks[i], d_images_next[i]);
        if (rc) return rc;
    }
    return CS_OK;
}

int cs_klt_group_advance(cs_klt_group* g) {
    CS_REQUIRE(g, "null group");
    for (cs_klt* k : g->ks) {
        int rc = cs_klt_advance(k);
        if (rc) return rc;
    }
    return CS_OK;
}

int cs_klt_group_synchronize(cs_klt_group* g) {
    CS_REQUIRE(g, "null group");
    int rc = bind_device(g->ks[0]);
    if (rc) return rc;
    CS_HIP(hipStreamSynchronize(g->stream));
    for (cs_klt* k : g->ks) {
        rc = check_device_error(k);
        if (rc) return rc;
    }
    return CS_OK;
}

}  // extern "C"
