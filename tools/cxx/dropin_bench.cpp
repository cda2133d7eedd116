// dropin_bench.cpp -- what a CoSLAM maintainer sees after swapping the headers: latency of the reference-shaped,
// synchronous C++ calls (host arrays in, host arrays out) through include/shim over libcoslam_hip.so.
//   g++ -O2 -std=c++11 -Iinclude -Iinclude/shim tools/cxx/dropin_bench.cpp -Lcoslam_amd/lib -lcoslam_hip \
//       -Wl,-rpath,$PWD/coslam_amd/lib -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -o tools/cxx/dropin_bench.bin
// Synthetic inputs: a band-limited noise image that shifts by a pixel per frame (640x480, 50x40 slots, CoSLAM's KLT
// parameters with 4 levels), 192 exact 3D-2D correspondences with noise, a 5 key frame x 500 point local BA.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "CGKLT/v3d_gpuklt.h"
#include "geometry/SL_BundleAdjust.h"
#include "slam/SL_IntraCamPose.h"
#include "slam/coslam_posegraph.h"

struct Mat_d {
    int rows, cols;
    std::vector<double> store;
    double* data;
    Mat_d(int r, int c, const double* d) : rows(r), cols(c), store(d, d + r * c), data(0) { data = store.data(); }
    Mat_d(const Mat_d& o) : rows(o.rows), cols(o.cols), store(o.store), data(0) { data = store.data(); }
};
struct Point3d {
    double M[3];
    Point3d(double x, double y, double z) { M[0] = x, M[1] = y, M[2] = z; }
};
struct Meas2D {
    int viewId;
    double x, y;
    int outlier;
    Meas2D(int v, double x_, double y_) : viewId(v), x(x_), y(y_), outlier(0) {}
};

// the members of GlobalPoseGraph / CamPoseNode / CamPoseEdge that relaxPoseGraphs binds to (src/slam/SL_GlobalPoseEstimation.h)
struct PgNode {
    bool fixed;
    double R[9], t[3], newR[9], newt[3];
};
struct PgEdge {
    int id1, id2;
    bool uncertainScale;
    double R[9], t[3];
};
struct PgGraph {
    int nNodes, nEdges;
    PgNode* poseNodes;
    PgEdge* poseEdges;
};

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static unsigned rng_state = 12345;
static double urand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (rng_state >> 8) / 16777216.0;
}
static double nrand() { return std::sqrt(-2.0 * std::log(urand() + 1e-12)) * std::cos(6.283185307179586 * urand()); }

int main() {
    if (cs_device_count() < 1) {
        printf("no HIP device: nothing to measure (there is no CPU fallback)\n");
        return 0;
    }
    const int W = 640, H = 480, L = 4, fw = 50, fh = 40;
    // texture: white noise, three 5-tap binomial passes, stretched
    std::vector<float> tex((W + 64) * H);
    for (auto& v : tex) v = (float)urand();
    for (int pass = 0; pass < 3; ++pass) {
        std::vector<float> t2(tex);
        const int TW = W + 64;
        for (int y = 2; y < H - 2; ++y)
            for (int x = 2; x < TW - 2; ++x) {
                float s = 0;
                for (int dy = -2; dy <= 2; ++dy)
                    for (int dx = -2; dx <= 2; ++dx) {
                        static const float k[5] = {1, 4, 6, 4, 1};
                        s += k[dy + 2] * k[dx + 2] * tex[(y + dy) * TW + x + dx];
                    }
                t2[y * TW + x] = s / 256.0f;
            }
        tex.swap(t2);
    }
    float lo = 1e9f, hi = -1e9f;
    for (float v : tex) lo = v < lo ? v : lo, hi = v > hi ? v : hi;
    for (auto& v : tex) v = 0.25f + 0.3f * (v - lo) / (hi - lo);
    {  // Gaussian blobs (sigma 1.2 px): the corners the detector is after
        const int TW = W + 64;
        for (int b = 0; b < 9000; ++b) {
            const double cx = 4 + (TW - 8) * urand(), cy = 4 + (H - 8) * urand(), amp = 0.25 + 0.35 * urand();
            for (int y = (int)cy - 4; y <= (int)cy + 4; ++y)
                for (int x = (int)cx - 4; x <= (int)cx + 4; ++x) {
                    const double d2 = (x - cx) * (x - cx) + (y - cy) * (y - cy);
                    tex[y * TW + x] += (float)(amp * std::exp(-d2 / (2 * 1.2 * 1.2)));
                }
        }
    }
    lo = 1e9f, hi = -1e9f;
    for (float v : tex) lo = v < lo ? v : lo, hi = v > hi ? v : hi;
    std::vector<std::vector<unsigned char> > frames(32, std::vector<unsigned char>(W * H));
    for (int f = 0; f < 32; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                frames[f][y * W + x] = (unsigned char)(24 + 208 * (tex[y * (W + 64) + x + f] - lo) / (hi - lo));

    V3D_GPU::KLT_SequenceTrackerConfig cfg;
    cfg.nIterations = 10, cfg.nLevels = L, cfg.levelSkip = 1, cfg.windowWidth = 7, cfg.trackWithGain = true;
    cfg.minCornerness = 3000.0f, cfg.convergenceThreshold = 1.0f, cfg.SSD_Threshold = 20000.0f, cfg.minDistance = 4;
    std::vector<V3D_GPU::KLT_TrackedFeature> feats(fw * fh);
    V3D_GPU::KLT_SequenceTracker trk(cfg);
    trk.allocate(W, H, L, fw, fh);
    int n = 0;
    trk.detect(frames[0].data(), n, feats.data());
    trk.advanceFrame();
    for (int i = 1; i < 20; ++i) {  // warm-up
        trk.redetect(frames[i % 32].data(), n, feats.data());
        trk.advanceFrame();
    }
    const int NF = 300;
    double t = now_us();
    for (int i = 0; i < NF; ++i) {
        trk.redetect(frames[(20 + i) % 32].data(), n, feats.data());
        trk.advanceFrame();
    }
    printf("KLT_SequenceTracker::redetect + advanceFrame (640x480, 2000 slots, host image in / dest[] out): %.1f us/frame, %d live\n",
           (now_us() - t) / NF, n);

    {   // the same through GPUKLT::next's two halves (cs_klt_redetect_async_h / cs_klt_fetch): 8 cameras, one handle each, every camera's
        // frame in flight before the first result is waited for -- what a host loop over the cameras costs when it does not serialise them
        const int NC8 = 8;
        cs_klt_config c8;
        cs_klt_config_default(&c8);
        c8.nIterations = 10, c8.nLevels = L, c8.levelSkip = 1, c8.windowWidth = 7, c8.trackWithGain = 1;
        c8.minCornerness = 3000.0f, c8.convergenceThreshold = 1.0f, c8.SSD_Threshold = 20000.0f, c8.minDistance = 4;
        cs_klt* k8[NC8];
        std::vector<std::vector<cs_klt_feature> > d8(NC8, std::vector<cs_klt_feature>(fw * fh));
        int n8 = 0;
        bool ok8 = true;
        for (int c = 0; c < NC8; ++c) {
            k8[c] = cs_klt_create(&c8, 0, 0);
            ok8 = ok8 && k8[c] && cs_klt_allocate(k8[c], W, H, L, fw, fh, 0, 0) == 0 && cs_klt_detect(k8[c], frames[c % 32].data(), &n8, d8[c].data()) == 0 &&
                  cs_klt_advance(k8[c]) == 0;
        }
        auto frame8 = [&](int i) {
            for (int c = 0; c < NC8 && ok8; ++c) ok8 = cs_klt_redetect_async_h(k8[c], frames[(i + c) % 32].data()) == 0;
            for (int c = 0; c < NC8 && ok8; ++c) ok8 = cs_klt_fetch(k8[c], &n8, d8[c].data()) == 0 && cs_klt_advance(k8[c]) == 0;
        };
        for (int i = 1; i < 20 && ok8; ++i) frame8(i);
        double t8 = now_us();
        for (int i = 0; i < NF && ok8; ++i) frame8(20 + i);
        if (ok8)
            printf("8 cameras through cs_klt_redetect_async_h + cs_klt_fetch (host images in / dest[] out, all cameras in flight): %.1f us per 8-camera frame "
                   "= %.0f frames/s, %d live in the last camera\n", (now_us() - t8) / NF, 1e6 * NF / (now_us() - t8), n8);
        else
            printf("8-camera async host form: FAILED (%s)\n", cs_last_error());
        for (int c = 0; c < NC8; ++c)
            if (k8[c]) cs_klt_destroy(k8[c]);
    }

    // intraCamEstimate
    double K[9] = {524.8, 0, 320, 0, 524.8, 240, 0, 0, 1}, R0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t0[3] = {0.02, -0.01, 0.03};
    const int NP = 192;
    std::vector<double> Ms(3 * NP), ms(2 * NP);
    for (int i = 0; i < NP; ++i) {
        Ms[3 * i] = -5 + 10 * urand(), Ms[3 * i + 1] = -3 + 6 * urand(), Ms[3 * i + 2] = 6 + 8 * urand();
        ms[2 * i] = K[0] * Ms[3 * i] / Ms[3 * i + 2] + K[2] + 0.5 * nrand();
        ms[2 * i + 1] = K[4] * Ms[3 * i + 1] / Ms[3 * i + 2] + K[5] + 0.5 * nrand();
    }
    double Ro[9], to[3];
    IntraCamPoseOption opt;
    for (int i = 0; i < 10; ++i) intraCamEstimate(K, R0, t0, NP, 0, Ms.data(), ms.data(), 10.0, Ro, to, &opt);
    t = now_us();
    for (int i = 0; i < 200; ++i) intraCamEstimate(K, R0, t0, NP, 0, Ms.data(), ms.data(), 10.0, Ro, to, &opt);
    printf("intraCamEstimate (192 points): %.1f us/call\n", (now_us() - t) / 200);

    // bundleAdjustRobust: 5 key frames x 500 points, all visible, the call queued at SL_CoSLAM.cpp:1345 (2 fixed, 2/10)
    const int C = 5, P = 500;
    std::vector<Mat_d> Ks, Rs, Ts;
    std::vector<Point3d> pts;
    std::vector<std::vector<Meas2D> > meas(P);
    std::vector<double> camx(C);
    for (int c = 0; c < C; ++c) {
        double tc[3] = {-0.3 * c, 0.0, 0.0};
        camx[c] = tc[0];
        Ks.push_back(Mat_d(3, 3, K));
        Rs.push_back(Mat_d(3, 3, R0));
        Ts.push_back(Mat_d(3, 1, tc));
    }
    for (int i = 0; i < P; ++i) {
        double X = -5 + 10 * urand(), Y = -3 + 6 * urand(), Z = 6 + 8 * urand();
        pts.push_back(Point3d(X + 0.05 * nrand(), Y + 0.05 * nrand(), Z + 0.05 * nrand()));
        for (int c = 0; c < C; ++c)
            meas[i].push_back(Meas2D(c, K[0] * (X + camx[c]) / Z + K[2] + 0.5 * nrand(), K[4] * Y / Z + K[5] + 0.5 * nrand()));
    }
    std::vector<Mat_d> Rs0(Rs), Ts0(Ts);
    std::vector<Point3d> pts0(pts);
    for (int i = 0; i < 5; ++i) {
        Rs = Rs0, Ts = Ts0, pts = pts0;
        bundleAdjustRobust(2, Ks, Rs, Ts, 2, pts, meas, 6.0, 2, 10);
    }
    t = now_us();
    for (int i = 0; i < 50; ++i) {
        Rs = Rs0, Ts = Ts0, pts = pts0;
        bundleAdjustRobust(2, Ks, Rs, Ts, 2, pts, meas, 6.0, 2, 10);
    }
    printf("bundleAdjustRobust (5 key frames x 500 points x 2500 measurements, maxIter 2 / inner 10): %.1f us/call\n",
           (now_us() - t) / 50);
    // the non-key frames behind a BA: 8 camera chains of 21 frames, key frame every 5th (RobustBundleRTS::updateNonKeyCameraPoses)
    {
        const int NC = 8, NN = 21;
        std::vector<std::vector<PgNode> > nodes(NC, std::vector<PgNode>(NN));
        std::vector<std::vector<PgEdge> > edges(NC, std::vector<PgEdge>(NN - 1));
        std::vector<PgGraph> graphs(NC);
        for (int c = 0; c < NC; ++c) {
            for (int i = 0; i < NN; ++i) {
                PgNode& nd = nodes[c][i];
                nd.fixed = (i % 5 == 0);
                const double a = 0.01 * i + 0.1 * c, ca = std::cos(a), sa = std::sin(a);
                const double R[9] = {ca, 0, sa, 0, 1, 0, -sa, 0, ca};
                for (int q = 0; q < 9; ++q) nd.R[q] = R[q];
                nd.t[0] = 0.05 * i, nd.t[1] = 0.01 * c, nd.t[2] = 4 + 0.02 * i;
            }
            for (int i = 0; i + 1 < NN; ++i) {  // getRigidTransFromTo(node i, node i + 1)
                PgEdge& e = edges[c][i];
                const PgNode &n1 = nodes[c][i], &n2 = nodes[c][i + 1];
                e.id1 = i, e.id2 = i + 1, e.uncertainScale = false;
                for (int r = 0; r < 3; ++r)
                    for (int q = 0; q < 3; ++q) e.R[3 * r + q] = n2.R[3 * r] * n1.R[3 * q] + n2.R[3 * r + 1] * n1.R[3 * q + 1] + n2.R[3 * r + 2] * n1.R[3 * q + 2];
                for (int r = 0; r < 3; ++r) e.t[r] = n2.t[r] - (e.R[3 * r] * n1.t[0] + e.R[3 * r + 1] * n1.t[1] + e.R[3 * r + 2] * n1.t[2]);
            }
            for (int i = 0; i < NN; i += 5) nodes[c][i].t[0] += 0.03;  // what the BA did to the key frames
            graphs[c].nNodes = NN, graphs[c].nEdges = NN - 1, graphs[c].poseNodes = nodes[c].data(), graphs[c].poseEdges = edges[c].data();
        }
        for (int i = 0; i < 10; ++i) relaxPoseGraphs(graphs.data(), NC);
        t = now_us();
        for (int i = 0; i < 200; ++i) relaxPoseGraphs(graphs.data(), NC);
        printf("relaxPoseGraphs (8 camera chains x 21 frames, key frame every 5th; topology + values up, poses back): %.1f us/call, "
               "node 3 of camera 0 moved by %.4f\n", (now_us() - t) / 200, nodes[0][3].newt[0] - nodes[0][3].t[0]);
    }
    trk.deallocate();
    return 0;
}
