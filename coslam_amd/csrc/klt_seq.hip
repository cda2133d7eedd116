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

extern "C" {

int cs_version(void) { return 100; }
const char* cs_last_error(void) { return g_err; }

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
    cs_klt_feature* h_dest;  // pinned
    int* h_counts;           // pinned
    float* h_feat;           // pinned
    uint8_t* h_img;          // pinned staging for the host-pointer entry points (a pageable source is staged by the runtime, ~3x slower)
    // HIP-event timing of the tracker stage (bench.py roofline leg): eager launches only
    bool profiling;
    hipEvent_t ev0, ev1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* ev_pairs;
    // persistent (single-launch) gain tracker
    bool use_fused;
    unsigned long long* d_gran;  // [granRows][N] hand-off granules of the persistent tracker (one row per pass + 1)
    int granRows;
    int* d_err;
    int cu_count;                 // compute units the handle's stream may use (cs_klt_set_cu_count; default: all)
    int concurrent;               // handles whose persistent kernels may overlap (cs_klt_set_concurrent_handles; 0: all live ones)
    unsigned long long* d_probe;  // diagnostic cycle counters of the persistent tracker (cs_klt_debug_probe)
    // hipGraph cache for the *_dev entry points: one executable graph per (call, buffer rotation state)
    bool use_graphs;
    struct GraphEntry {
        int mode, b0, b1, b2, p0, p1;
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
constexpr int CS_RESIDENT_BLOCKS_PER_CU = 6;  // of the 8 x 256-thread blocks a CU admits, minus a margin

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

// ---- frame schedules (all asynchronous on k->stream) -------------------------------------------

// postDest != null: the persistent tracker may fold k_post_track into its epilogue; *postFused says whether it did
static int enqueue_tracker(cs_klt* k, cs_klt_feature* postDest, int doSuppress, bool* postFused) {
    if (postFused) *postFused = false;
    const cs_klt_config& c = k->cfg;
    const int hw = c.windowWidth / 2;
    const cs_texel *P0 = k->d_pyr[k->p0], *P1 = k->d_pyr[k->p1];
    if (!c.trackWithGain) {
        // the host passes -DNITERATIONS but the shader reads N_ITERATIONS: always 5
        // (v3d_gpuklt.cpp:108 vs klt_tracker.cg:16-18)
        return cs_launch_track_nogain(P0, P1, k->lay, c.levelSkip, hw, 5, k->margin, k->convThr, k->ssdThr, k->N,
                                      k->d_fb[k->b0], k->d_fb[k->b1], k->stream);
    }
    int levelSkipF = c.levelSkip > 0 ? c.levelSkip : (c.nLevels - 1);
    if (levelSkipF <= 0) levelSkipF = 1;
    int nLevelsVisited = 0;
    for (int level = k->L - 1; level >= 0; level -= levelSkipF) ++nLevelsVisited;
    const int T = nLevelsVisited * c.nIterations;
    int live = 1;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        live = g_live_handles[k->device & 63] > 0 ? g_live_handles[k->device & 63] : 1;
    }
    if (k->concurrent > 0 && k->concurrent < live) live = k->concurrent;
    const int blocks = (k->N + 3) / 4;
    if (k->use_fused && T >= 1 && T + 1 <= k->granRows && blocks * live <= CS_RESIDENT_BLOCKS_PER_CU * k->cu_count &&
        (2 * hw + 1) * (2 * hw + 1) <= 256) {
        CsGainFusedArgs f;
        memset(&f, 0, sizeof(f));
        f.pyr0 = P0;
        f.pyr1 = P1;
        f.lv.L = k->L;
        for (int l = 0; l < k->L; ++l) {
            f.lv.w[l] = k->lay.w[l];
            f.lv.h[l] = k->lay.h[l];
            f.lv.off[l] = k->lay.off[l];
        }
        f.W = k->W;
        f.H = k->H;
        f.fw = k->fw;
        f.fh = k->fh;
        f.N = k->N;
        f.hw = hw;
        f.nIter = c.nIterations;
        f.levelSkip = levelSkipF;
        f.feat0 = k->d_fb[k->b2];
        f.featStart = k->d_fb[k->b0];
        // where the ping-pong schedule of the reference leaves its last two results (v3d_gpuklt.cpp:281-285)
        f.outLast = k->d_fb[(T & 1) ? k->b1 : k->b0];
        f.outPrev = k->d_fb[(T & 1) ? k->b0 : k->b1];
        f.gran = k->d_gran;
        f.tagWord = (const unsigned*)(k->d_counts + 5);
        f.sqrConvThr = k->convThr * k->convThr;
        f.ssdThr = k->ssdThr;
        f.vr[0] = k->margin / (float)k->W;
        f.vr[1] = k->margin / (float)k->H;
        f.vr[2] = 1.0f - k->margin / (float)k->W;
        f.vr[3] = 1.0f - k->margin / (float)k->H;
        f.lambda = 1.0f;
        f.delta = 200.0f;
        {
            const double rxy = (double)k->fh / (double)k->fw, ryx = (double)k->fw / (double)k->fh;
            f.n1x[0] = 1;
            f.n1x[1] = -1;
            f.n1x[2] = (int)floor(0.5 + ryx);
            f.n1x[3] = (int)floor(0.5 - ryx);
            f.n1y[0] = (int)floor(0.5 + rxy);
            f.n1y[1] = (int)floor(0.5 - rxy);
            f.n1y[2] = 1;
            f.n1y[3] = -1;
        }
        f.err = k->d_err;
        f.probe = k->d_probe;
        f.dest = postDest;
        f.ctr = k->d_ctr;
        f.corner = k->d_corner_raw;
        f.doSuppress = doSuppress;
        if (postFused) *postFused = (postDest != nullptr);
        f.pollGap = 0;
        int rcf = cs_launch_track_gain_fused(f, k->stream);
        if (rcf) return rcf;
        if (T & 1) std::swap(k->b0, k->b1);  // T swaps of (buffer0, buffer1)
        std::swap(k->b0, k->b2);             // v3d_gpuklt.cpp:304
        return CS_OK;
    }
    int rc = cs_launch_reset_beta(k->d_fb[k->b0], k->N, k->stream);  // v3d_gpuklt.cpp:223-227
    if (rc) return rc;
    CsGainPassArgs a;
    memset(&a, 0, sizeof(a));
    a.whx = (float)k->W;
    a.why = (float)k->H;
    a.fw = k->fw;
    a.fh = k->fh;
    a.N = k->N;
    a.hw = hw;
    a.lambda = 1.0f;  // :250
    {
        // st0 +- ds0.x / ds0.y are scalar broadcasts (klt_tracker_with_gain.cg:64-67)
        const double rxy = (double)k->fh / (double)k->fw, ryx = (double)k->fw / (double)k->fh;
        a.n1x[0] = 1;
        a.n1x[1] = -1;
        a.n1x[2] = (int)floor(0.5 + ryx);
        a.n1x[3] = (int)floor(0.5 - ryx);
        a.n1y[0] = (int)floor(0.5 + rxy);
        a.n1y[1] = (int)floor(0.5 - rxy);
        a.n1y[2] = 1;
        a.n1y[3] = -1;
    }
    float delta = 200.0f;
    const float tau = 1.0f;
    int levelSkip = c.levelSkip > 0 ? c.levelSkip : (c.nLevels - 1);  // v3d_gpuklt.h:14
    if (levelSkip <= 0) levelSkip = 1;
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
            a.delta = delta;
            delta *= tau;
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
            rc = cs_launch_track_gain_pass(a, k->stream);
            if (rc) return rc;
            std::swap(k->b0, k->b1);  // :285
        }
    }
    std::swap(k->b0, k->b2);  // :304
    return CS_OK;
}

static int enqueue_detect_tail(cs_klt* k, int mode, int nPresentGiven, int maxKeepFixed, cs_klt_feature* d_dest,
                               int* d_counts) {
    // a pending cs_klt_prefetch_dev request rides in the tail's two launches; inside a graph capture it is dropped
    const uint8_t* next = (const uint8_t*)k->pf_next_img;
    k->pf_next_img = nullptr;
    if (next) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(k->stream, &cap);
        if (cap != hipStreamCaptureStatusNone || cs_nonmax_lds_bytes(k->cfg.minDistance) > 64 * 1024) next = nullptr;
    }
    int rc = CS_OK;
    if (!next) {
        rc = cs_launch_nonmax_compact(k->d_corner_raw, k->W, k->H, k->cfg.minDistance, k->d_corner, k->d_cand, k->maxCand,
                                      k->d_ctr, k->stream);
        if (rc) return rc;
    }
    CsFillArgs f;
    f.mode = mode;
    f.N = k->N;
    f.withGain = k->cfg.trackWithGain;
    f.nPresentGiven = nPresentGiven;
    f.present3 = k->d_present;
    f.sel = k->d_sel;
    f.ctr = k->d_ctr;
    f.dest = d_dest;
    // provideFeatures / provideFeaturesAndGain, v3d_gpuklt.cpp:86-92,188-197
    f.list_a = k->d_fb[k->b1];
    f.list_b = k->cfg.trackWithGain ? k->d_fb[k->b2] : nullptr;
    f.counts = d_counts;
    f.tagWord = (unsigned*)(k->d_counts + 5);
    if (next) {
        // spare buffers: last read by the tracker / non-max of the frame BEFORE this one -- older than this point of
        // the stream
        rc = cs_launch_tail_with_next_front(k->d_corner_raw, k->W, k->H, k->cfg.minDistance, k->d_corner, k->d_cand,
                                            k->maxCand, k->plw * k->plh, maxKeepFixed, k->d_rank, k->d_sel, f, next, k->lay,
                                            k->d_pyr[k->p2], k->tap_mode, k->d_corner_raw_spare, k->cfg.minCornerness,
                                            k->detMargin, k->stream);
        if (rc) return rc;
        k->pf_img = next;
        k->pf_valid = true;
        return CS_OK;
    }
    return cs_launch_select_fill(k->d_cand, k->maxCand, k->plw * k->plh, maxKeepFixed, k->d_rank, k->d_sel, f, k->stream);
}

static int enqueue_track(cs_klt* k, const uint8_t* d_img, cs_klt_feature* d_dest, int* d_counts, bool forRedetect) {
    // pyramid (:858), the cornerness map the detector will need, and the zeroing of this frame's counters and
    // hand-off granules: two launches (klt_pyramid.hip)
    int rc = CS_OK;
    if (k->pf_valid && k->pf_img == (const void*)d_img) {
        // prefetched: the spare pyramid / cornerness buffers become this frame's, the ones they replace (last read two
        // frames ago) become the next prefetch's targets.  The candidate counter was zeroed by the previous frame's tail.
        std::swap(k->p1, k->p2);
        std::swap(k->d_corner_raw, k->d_corner_raw_spare);
    } else {
        rc = cs_launch_frame_front(d_img, k->lay, k->d_pyr[k->p1], k->tap_mode, forRedetect ? k->d_corner_raw : nullptr,
                                   k->cfg.minCornerness, k->detMargin, k->d_ctr, nullptr, 0, k->stream);
        if (rc) return rc;
    }
    k->pf_valid = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (k->profiling) {
        CS_HIP(hipEventCreate(&e0));
        CS_HIP(hipEventCreate(&e1));
        CS_HIP(hipEventRecord(e0, k->stream));
    }
    bool postFused = false;
    rc = enqueue_tracker(k, d_dest, forRedetect ? 1 : 0, &postFused);
    if (rc) return rc;
    if (k->profiling) {
        CS_HIP(hipEventRecord(e1, k->stream));
        k->ev_pairs->push_back(std::make_pair(e0, e1));
    }
    if (!postFused) {
        rc = cs_launch_post_track(read_buffer(k), k->N, d_dest, k->d_ctr, k->d_corner_raw, k->W, k->H,
                                  forRedetect ? 1 : 0, k->stream);
        if (rc) return rc;
    }
    if (!forRedetect) {
        k->pf_next_img = nullptr;  // no detector tail to carry the next front: the request lapses
        rc = cs_launch_counts_track(d_dest, k->N, d_counts, k->d_ctr, (unsigned*)(k->d_counts + 5), k->stream);
    }
    return rc;
}

static int enqueue_redetect(cs_klt* k, const uint8_t* d_img, cs_klt_feature* d_dest, int* d_counts) {
    int rc = enqueue_track(k, d_img, d_dest, d_counts, true);
    if (rc) return rc;
    return enqueue_detect_tail(k, 2, 0, -1, d_dest, d_counts);
}

static int enqueue_detect(cs_klt* k, const uint8_t* d_img, cs_klt_feature* d_dest, int* d_counts, int nPresent) {
    k->pf_valid = false;
    int rc = cs_launch_frame_front(d_img, k->lay, k->d_pyr[k->p1], k->tap_mode, k->d_corner_raw, k->cfg.minCornerness,
                                   k->detMargin, k->d_ctr, nullptr, 0, k->stream);
    if (rc) return rc;
    if (nPresent > 0) {
        rc = cs_launch_suppress_list(k->d_corner_raw, k->W, k->H, nPresent, k->d_present, k->stream);
        if (rc) return rc;
    }
    rc = cs_launch_clear_dest(d_dest, k->N, k->stream);
    if (rc) return rc;
    int maxKeep = k->N - nPresent;
    if (maxKeep < 0) maxKeep = 0;
    return enqueue_detect_tail(k, nPresent > 0 ? 1 : 0, nPresent, maxKeep, d_dest, d_counts);
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
    const char* env = getenv("COSLAM_KLT_FUSED");
    k->use_fused = !(env && env[0] == '0');
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
    hipFree(k->d_gran);
    if (k->d_probe) hipFree(k->d_probe);
    k->d_probe = nullptr;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        g_live_handles[k->device & 63]--;
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
    CS_HIP(hipHostMalloc((void**)&k->h_feat, sizeof(float) * 3 * k->presentCap, hipHostMallocDefault));
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
    }
    return CS_OK;
}

int cs_klt_set_border_margin(cs_klt* k, float m) {  // v3d_gpuklt.h:219-226
    CS_REQUIRE(k, "null handle");
    k->margin = m;
    k->detMargin = m;
    return CS_OK;
}
int cs_klt_set_convergence_threshold(cs_klt* k, float t) {
    CS_REQUIRE(k, "null handle");
    k->convThr = t;
    return CS_OK;
}
int cs_klt_set_ssd_threshold(cs_klt* k, float t) {
    CS_REQUIRE(k, "null handle");
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
        *launches_per_frame = !c.trackWithGain ? 1 : (k->use_fused ? 1 : lv * c.nIterations + 1);
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
        if (mode == 0) return enqueue_detect(k, img, (cs_klt_feature*)d_dest, (int*)d_counts, 0);
        if (mode == 1) return enqueue_redetect(k, img, (cs_klt_feature*)d_dest, (int*)d_counts);
        return enqueue_track(k, img, (cs_klt_feature*)d_dest, (int*)d_counts, false);
    };
    if (!k->use_graphs || k->profiling) return enqueue((const uint8_t*)d_image);
    CS_HIP(hipMemcpyAsync(k->d_img, d_image, (size_t)k->W * k->H, hipMemcpyDeviceToDevice, k->stream));
    for (auto& g : *k->graphs) {
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
    if ((rc = enqueue_detect(k, k->d_img, k->d_dest, k->d_counts, 0))) return rc;
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
    if ((rc = enqueue_detect(k, k->d_img, k->d_dest, k->d_counts, nPresent))) return rc;
    return fetch_results(k, nDetected, dest);
}

int cs_klt_redetect(cs_klt* k, const uint8_t* image, int* nNew, cs_klt_feature* dest) {
    CS_REQUIRE(k && k->allocated && image && nNew && dest, "cs_klt_redetect: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    if ((rc = upload_image(k, image))) return rc;
    if ((rc = enqueue_redetect(k, k->d_img, k->d_dest, k->d_counts))) return rc;
    return fetch_results(k, nNew, dest);
}

int cs_klt_track(cs_klt* k, const uint8_t* image, int* nPresent, cs_klt_feature* dest) {
    CS_REQUIRE(k && k->allocated && image && nPresent && dest, "cs_klt_track: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    if ((rc = upload_image(k, image))) return rc;
    if ((rc = enqueue_track(k, k->d_img, k->d_dest, k->d_counts, false))) return rc;
    return fetch_results(k, nPresent, dest);
}

int cs_klt_feed(cs_klt* k, int npts, const float* featPts, int* trackIds, int* nFed) {  // v3d_gpuklt.cpp:808-855
    CS_REQUIRE(k && k->allocated && nFed && npts >= 0 && (npts == 0 || (featPts && trackIds)), "cs_klt_feed: bad arguments");
    int rc = bind_device(k);
    if (rc) return rc;
    float* c = k->h_feat;
    CS_HIP(hipMemcpyAsync(c, read_buffer(k), sizeof(float) * 3 * k->N, hipMemcpyDeviceToHost, k->stream));
    CS_HIP(hipStreamSynchronize(k->stream));
    const double radius2 = 1e-4;
    for (int q = 0; q < npts; ++q) {
        for (int i = 0; i < k->N; ++i) {
            if (c[3 * i] < 0) continue;
            // stride-2 read of the stride-3 list: reproduces v3d_gpuklt.cpp:826-827 as shipped
            double dx = (double)(featPts[2 * q] - c[3 * i]);
            double dy = (double)(featPts[2 * q + 1] - c[3 * i + 1]);
            if (dx * dx + dy * dy < radius2) c[3 * i] = -1.0f;
        }
    }
    int q = 0;
    for (int i = 0; i < k->N && q < npts; ++i) {
        if (c[3 * i] < 0) {
            c[3 * i] = featPts[3 * q];
            c[3 * i + 1] = featPts[3 * q + 1];
            c[3 * i + 2] = 1.0f;
            trackIds[q] = i;
            ++q;
        }
    }
    *nFed = q;
    CS_HIP(hipMemcpyAsync(k->d_fb[k->b1], c, sizeof(float) * 3 * k->N, hipMemcpyHostToDevice, k->stream));
    if (k->cfg.trackWithGain) {
        CS_HIP(hipMemcpyAsync(k->d_fb[k->b2], c, sizeof(float) * 3 * k->N, hipMemcpyHostToDevice, k->stream));
    }
    CS_HIP(hipStreamSynchronize(k->stream));
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
    if ((rc = cs_launch_frame_front(k->d_img, k->lay, k->d_pyr[k->p1], k->tap_mode, nullptr, 0.0f, 0.0f, nullptr, nullptr, 0,
                                    k->stream)))
        return rc;
    CS_HIP(hipStreamSynchronize(k->stream));
    return CS_OK;
}

}  // extern "C"
