// ref_gpuklt_dropin_test.cpp -- the reference's OWN tracker facade driven on the MI355X.
//
// oracle/Makefile compiles /root/reference/src/tracking/GPUKLT.cpp, SL_Track2D.cpp and src/slam/SL_FeaturePoint(s).cpp,
// SL_MapPoint.cpp, SL_Camera.cpp IN PLACE (never copied) against include/shim/CGKLT/v3d_gpuklt.h -- the header-compatible
// V3D_GPU::KLT_SequenceTracker over libcoslam_hip.so -- plus stand-ins for the un-vendored LibVisualSLAM headers
// (oracle/ref_shim/).  This driver is the reference's caller: GPUKLT::init / setIntrinsicParam / first / next /
// feedExternFeatPoints / detectCorners / getCurrentCorners, exactly as SingleSLAM uses them
// (src/app/SL_SingleSLAM.cpp:291-304,329-331).  It checks
//   (1) the drop-in itself: the unchanged reference source runs over the shim and tracks a synthetic sequence;
//   (2) the on-device hand-back (cs_klt_handback_dev) against the reference's addToFeaturePoints loop:
//       the same frames go through a second tracker driven through the C-ABI, dest[] is handed back on the device, and
//       per slot the undistorted pixel, the Track2D state (empty / frame span) and the feature points added per frame
//       must equal what GPUKLT::addToFeaturePoints built with its FeaturePoints / Track2D lists -- bit for bit.
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists, run by tests/test_cxx_dropin_gpu.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tracking/GPUKLT.h"

#include "coslam_hip.h"

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                    \
        }                                                                \
    } while (0)

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static double urand() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return (double)(g_rng >> 11) / 9007199254740992.0;
}

// blobs drifting by (dx, dy) per frame over a flat background
static void render(int W, int H, const std::vector<double>& bx, const std::vector<double>& by, const std::vector<double>& amp,
                   double dx, double dy, std::vector<unsigned char>& img) {
    std::vector<double> acc((size_t)W * H, 110.0);
    for (size_t k = 0; k < bx.size(); ++k) {
        const double u = bx[k] + dx, v = by[k] + dy;
        const int ci = (int)floor(u), cj = (int)floor(v);
        for (int j = cj - 5; j <= cj + 5; ++j)
            for (int i = ci - 5; i <= ci + 5; ++i) {
                if (i < 0 || j < 0 || i >= W || j >= H) continue;
                const double d2 = (i + 0.5 - u) * (i + 0.5 - u) + (j + 0.5 - v) * (j + 0.5 - v);
                acc[(size_t)j * W + i] += amp[k] * exp(-d2 / (2 * 1.3 * 1.3));
            }
    }
    img.resize((size_t)W * H);
    for (size_t p = 0; p < img.size(); ++p) {
        double v = floor(acc[p] + 0.5);
        img[p] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

int main() {
    const int W = 640, H = 480, NF = 6;
    std::vector<double> bx, by, amp;
    for (int k = 0; k < 1500; ++k) {
        bx.push_back(urand() * W);
        by.push_back(urand() * H);
        amp.push_back((60 + 100 * urand()) * (urand() < 0.5 ? -1 : 1));
    }
    std::vector<std::vector<unsigned char> > frames(NF);
    for (int f = 0; f < NF; ++f) render(W, H, bx, by, amp, 1.3 * f, -0.7 * f, frames[f]);

    // CoSLAM's own configuration (src/app/SL_SingleSLAM.cpp:291-298, src/app/SL_GlobParam.cpp:28-34)
    V3D_GPU::KLT_SequenceTrackerConfig cfg;
    cfg.minDistance = 8;
    cfg.minCornerness = 1500.0f;
    cfg.nLevels = 6;
    cfg.windowWidth = 6;
    cfg.convergenceThreshold = 1.0f;
    cfg.SSD_Threshold = 20000.0f;
    cfg.trackWithGain = true;
    const double K[9] = {0.82 * W, 0, W / 2.0, 0, 0.82 * W, H / 2.0, 0, 0, 1};
    const double iK[9] = {1 / K[0], 0, -K[2] / K[0], 0, 1 / K[4], -K[5] / K[4], 0, 0, 1};
    const double kud[7] = {0.2, 0.02, 0, 0, 0, 0, 0};  // pushes border points outwards: the out >= W | H rule fires

    // ---- the reference's facade over the shim
    GPUKLT klt;
    klt.init(0, W, H, &cfg);
    klt.setIntrinsicParam(K, iK, kud);
    FeaturePoints ips;
    const int N = klt.m_nMaxCorners;
    CHECK(N == SLAM_FEATURE_WIDTH * SLAM_FEATURE_HEIGHT);

    // ---- the same frames through the C-ABI + the device hand-back
    cs_klt_config c;
    cs_klt_config_default(&c);
    c.minDistance = cfg.minDistance;
    c.minCornerness = cfg.minCornerness;
    c.nLevels = cfg.nLevels;
    c.windowWidth = cfg.windowWidth;
    c.convergenceThreshold = cfg.convergenceThreshold;
    c.SSD_Threshold = cfg.SSD_Threshold;
    c.trackWithGain = 1;
    cs_klt* k2 = cs_klt_create(&c, 0, 0);
    CHECK(k2 && cs_klt_allocate(k2, W, H, cfg.nLevels, SLAM_FEATURE_WIDTH, SLAM_FEATURE_HEIGHT, 0, 0) == CS_OK);
    std::vector<cs_klt_feature> dest(N);
    void *d_dest, *d_K, *d_kud, *d_map, *d_s2m, *d_span, *d_xy, *d_state, *d_Ms, *d_ms, *d_sel, *d_npts;
    CHECK(hipMalloc(&d_dest, sizeof(cs_klt_feature) * N) == hipSuccess && hipMalloc(&d_K, 72) == hipSuccess &&
          hipMalloc(&d_kud, 56) == hipSuccess && hipMalloc(&d_map, 24) == hipSuccess && hipMalloc(&d_s2m, 4 * N) == hipSuccess &&
          hipMalloc(&d_span, 8 * N) == hipSuccess && hipMalloc(&d_xy, 16 * N) == hipSuccess && hipMalloc(&d_state, 4 * N) == hipSuccess &&
          hipMalloc(&d_Ms, 192 * 24) == hipSuccess && hipMalloc(&d_ms, 192 * 16) == hipSuccess && hipMalloc(&d_sel, 192 * 4) == hipSuccess &&
          hipMalloc(&d_npts, 4) == hipSuccess);
    hipMemcpy(d_K, K, 72, hipMemcpyHostToDevice);
    hipMemcpy(d_kud, kud, 56, hipMemcpyHostToDevice);
    hipMemset(d_map, 0, 24);
    hipMemset(d_s2m, 0xff, 4 * N);
    hipMemset(d_span, 0xff, 8 * N);
    hipMemset(d_xy, 0, 16 * N);
    cs_handback_cam hc;
    memset(&hc, 0, sizeof(hc));
    hc.dest = (const cs_klt_feature*)d_dest;
    hc.K = (const double*)d_K;
    hc.kud = (const double*)d_kud;
    hc.mapPts = (const double*)d_map;
    hc.slot2map = (int*)d_s2m;
    hc.trackSpan = (int*)d_span;
    hc.xy = (double*)d_xy;
    hc.state = (int*)d_state;
    hc.Ms = (double*)d_Ms;
    hc.ms = (double*)d_ms;
    hc.sel = (int*)d_sel;
    hc.npts = (int*)d_npts;
    std::vector<int> span(2 * N), state(N);
    std::vector<double> xy(2 * N);

    int dropped = 0, longest = 0;
    for (int f = 0; f < NF; ++f) {
        int nRef = 0, nAbi = 0;
        if (f == 0) {
            nRef = klt.first(0, frames[0].data(), ips);  // detect + advanceFrame + addToFeaturePoints
            CHECK(cs_klt_detect(k2, frames[0].data(), &nAbi, dest.data()) == CS_OK);
        } else {
            nRef = klt.next(frames[f].data(), ips);  // redetect + addToFeaturePoints + advanceFrame
            CHECK(cs_klt_redetect(k2, frames[f].data(), &nAbi, dest.data()) == CS_OK);
        }
        CHECK(cs_klt_advance(k2) == CS_OK);
        CHECK(nRef == nAbi && klt.currentFrame() == f);
        hipMemcpy(d_dest, dest.data(), sizeof(cs_klt_feature) * N, hipMemcpyHostToDevice);
        CHECK(cs_klt_handback_dev(0, 0, 1, &hc, N, W, H, 16, 12, 192, f) == CS_OK);
        CHECK(hipDeviceSynchronize() == hipSuccess);
        hipMemcpy(span.data(), d_span, 8 * N, hipMemcpyDeviceToHost);
        hipMemcpy(state.data(), d_state, 4 * N, hipMemcpyDeviceToHost);
        hipMemcpy(xy.data(), d_xy, 16 * N, hipMemcpyDeviceToHost);
        int added = 0;
        for (int i = 0; i < N; ++i) {
            const Track2D& tk = klt.m_tks[i];
            CHECK(tk.empty() == (span[i] < 0));
            if (state[i] >= 0) ++added;
            if (state[i] == -2) ++dropped;
            if (tk.empty()) continue;
            CHECK(tk.f1 == span[i] && tk.f2 == span[N + i] && tk.length() == span[N + i] - span[i] + 1);
            CHECK(tk.tail->x == xy[i] && tk.tail->y == xy[N + i]);  // undistorPoint + the float -> double widening, bit for bit
            if (state[i] >= 0) CHECK(tk.tail->f == f && tk.tail->pt && tk.tail->pt->x == xy[i]);
            if (tk.length() > longest) longest = tk.length();
        }
        CHECK(ips.totalFrameNum(f) == added);  // one FeaturePoint per surviving feature per frame (SL_FeaturePoints.cpp:81-87)
        printf("frame %d: %d detected/new, %d feature points added, %d tracks alive\n", f, nRef, added, added);
    }
    CHECK(longest == NF);    // features tracked through the whole sequence by the unchanged reference source
    CHECK(dropped > 0);      // the out >= W | H rule was exercised on both sides

    // GPUKLT::getCurrentCorners and feedExternFeatPoints (src/tracking/GPUKLT.cpp:163-190,232-243)
    Mat_d cur;
    klt.getCurrentCorners(cur);
    CHECK(cur.rows > 100);
    std::vector<FeaturePoint*> ext;
    FeaturePoints extPts;
    for (int q = 0; q < 5; ++q) ext.push_back(extPts.add(NF - 1, 0, 50.0 + 40 * q, 60.0 + 30 * q));
    const int nFed = klt.feedExternFeatPoints(ext);
    CHECK(nFed >= 0 && nFed <= 5);
    // GPUKLT::detectCorners on a fresh object (src/tracking/GPUKLT.cpp:191-231; used by SL_InitMap.cpp:95-124)
    GPUKLT det;
    Mat_f corners;
    det.detectCorners(W, H, frames[0].data(), corners, 1500.0f, 5);
    CHECK(corners.rows > 100 && corners.cols == 2);
    cs_klt_destroy(k2);
    printf("ref GPUKLT drop-in ok: %d slots, longest track %d frames, %d drops by the >= W|H rule, %d corners\n", N, longest,
           dropped, corners.rows);
    return 0;
}
