// ref_register_test.cpp -- the reference's OWN searchMahaNearestFeatPt against the registration search kernel.
//
// oracle/Makefile compiles /root/reference/src/app/SL_SingleSLAM.cpp (searchMahaNearestFeatPt, :1141-1164) and the data
// model (src/slam/SL_FeaturePoints.cpp, SL_FeaturePoint.cpp, ...) IN PLACE against oracle/ref_shim/ (mat22Inv, matScale,
// mahaDist2 are un-vendored LibVisualSLAM: stand-ins).  This driver builds FeaturePoints lists with the reference's
// classes and asks the reference's function.
//   ref_register_test golden <out.bin>   CPU only: writes the feature list, the queries and the reference's answers --
//                                         tests/golden/make_golden.py turns them into tests/golden/register_golden.npz,
//                                         the fixture that pins oracle/register_oracle.c (and, on the GPU box, the kernel).
//   ref_register_test                     MI355X: cs_register_search (libcoslam_hip.so) on the same lists; every candidate
//                                         the kernel returns must be the feature point the reference's function returns
//                                         for the kernel's own (m, var), skipped pairs must be the ones the loops skip.
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists, run by tests/test_cxx_dropin_gpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "app/SL_SingleSLAM.h"
#include "slam/SL_FeaturePoints.h"

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

static const int W = 640, H = 480, FRAME = 7;

// one camera's current frame: N slots, some dead (state -1) or dropped (-2), a few exact duplicates (ties)
struct Cam {
    std::vector<double> xy;  // x[N] then y[N]
    std::vector<int> state, slot2map;
    FeaturePoints list;                    // the reference's container: frame FRAME - 1 (decoys) and frame FRAME
    std::map<FeaturePoint*, int> slotOf;   // feature point of frame FRAME -> slot
};

static void build_cam(Cam& c, int N, int camId) {
    c.xy.assign(2 * N, 0.0);
    c.state.assign(N, 0);
    c.slot2map.assign(N, -1);
    for (int i = 0; i < N / 3; ++i) c.list.add(FRAME - 1, camId, urand() * W, urand() * H);  // older frame: never a candidate
    for (int i = 0; i < N; ++i) {
        const double r = urand();
        c.state[i] = r < 0.08 ? -1 : (r < 0.10 ? -2 : (r < 0.3 ? 1 : 0));
        double x = urand() * W, y = urand() * H;
        if (i >= 8 && (i % 97) == 0) {  // exact duplicate of an earlier slot: the first one must win
            x = c.xy[i - 8];
            y = c.xy[N + i - 8];
        }
        c.xy[i] = x;
        c.xy[N + i] = y;
        c.slot2map[i] = urand() < 0.4 ? (int)(urand() * 1000) : -1;
        if (c.state[i] == 0 || c.state[i] == 1) c.slotOf[c.list.add(FRAME, camId, x, y)] = i;  // GPUKLT.cpp:48
    }
}

static int run_golden(const char* path) {
    const int N = 700, Q = 400;
    Cam c;
    build_cam(c, N, 0);
    std::vector<double> q(7 * Q);
    std::vector<int> ans(Q);
    for (int k = 0; k < Q; ++k) {
        double* e = &q[7 * k];
        e[0] = urand() * W;
        e[1] = urand() * H;
        if (k % 10 == 0) {  // on top of a duplicated pair
            const int s = 97 * (1 + k / 10 % 6);
            e[0] = c.xy[s] + 1e-3;
            e[1] = c.xy[N + s];
        }
        const double a = 1 + 40 * urand(), b = 1 + 40 * urand(), r = (2 * urand() - 1) * 0.9 * sqrt(a * b);
        e[2] = a;
        e[3] = r;
        e[4] = r;
        e[5] = b;
        e[6] = k % 2 ? 30.0 : 40.0;
        double m[2] = {e[0], e[1]}, var[4] = {e[2], e[3], e[4], e[5]};
        FeaturePoint* fp = searchMahaNearestFeatPt(c.list, FRAME, m, var, e[6]);
        ans[k] = fp ? c.slotOf.at(fp) : -1;
    }
    // a frame without features: the function returns 0
    double m[2] = {10, 10}, var[4] = {1, 0, 0, 1};
    const int none = searchMahaNearestFeatPt(c.list, FRAME + 5, m, var, 30.0) ? 1 : 0;
    FILE* f = fopen(path, "wb");
    if (!f) return 2;
    const int hdr[4] = {N, Q, none, 0};
    fwrite(hdr, sizeof(int), 4, f);
    fwrite(c.xy.data(), sizeof(double), 2 * N, f);
    fwrite(c.state.data(), sizeof(int), N, f);
    fwrite(q.data(), sizeof(double), 7 * Q, f);
    fwrite(ans.data(), sizeof(int), Q, f);
    fclose(f);
    printf("ref_register_test: wrote %d queries over %d slots (%zu in the frame list)\n", Q, N, c.slotOf.size());
    return 0;
}

static int run_gpu() {
    const int nCams = 3, N = 2000, P = 600;
    std::vector<Cam> cams(nCams);
    // cameras: K = I, R = I, t = 0 for camera 0 (project(M) = (X/Z, Y/Z)), small offsets for the others
    std::vector<double> Ks(9 * nCams, 0.0), Rs(9 * nCams, 0.0), ts(3 * nCams, 0.0);
    for (int c = 0; c < nCams; ++c) {
        build_cam(cams[c], N, c);
        for (int d = 0; d < 3; ++d) Ks[9 * c + 4 * d] = Rs[9 * c + 4 * d] = 1.0;
        ts[3 * c] = 3.0 * c;
        ts[3 * c + 1] = -2.0 * c;
    }
    std::vector<double> M(3 * P), cov(9 * P);
    std::vector<int> pf(P * nCams, -1);
    for (int p = 0; p < P; ++p) {
        const double z = (p % 50 == 7) ? -1.0 : 1.0;  // some behind the camera
        M[3 * p] = (urand() * 1.2 - 0.1) * W * z;    // some outside the image
        M[3 * p + 1] = (urand() * 1.2 - 0.1) * H * z;
        M[3 * p + 2] = z;
        double A[9];
        for (int k = 0; k < 9; ++k) A[k] = urand() - 0.5;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                cov[9 * p + 3 * i + j] = 4e-4 * (A[3 * i] * A[3 * j] + A[3 * i + 1] * A[3 * j + 1] + A[3 * i + 2] * A[3 * j + 2]) + (i == j ? 1e-5 : 0);
        for (int c = 0; c < nCams; ++c)
            if (urand() < 0.2) pf[p * nCams + c] = (int)(urand() * N);
    }
    std::vector<cs_register_cam> rc(nCams);
    for (int c = 0; c < nCams; ++c) {
        rc[c].K = &Ks[9 * c];
        rc[c].R = &Rs[9 * c];
        rc[c].t = &ts[3 * c];
        rc[c].xy = cams[c].xy.data();
        rc[c].state = cams[c].state.data();
        rc[c].slot2map = cams[c].slot2map.data();
        rc[c].isDynamic = nullptr;
    }
    const double pixelErrVar = 10.0;  // Const::PIXEL_ERR_VAR, src/app/SL_GlobParam.cpp:37
    std::vector<int> slot(P * nCams), flags(P * nCams);
    std::vector<double> m(2 * P * nCams), var(4 * P * nCams), dist(P * nCams);
    int rcode = cs_register_search(0, nCams, rc.data(), N, W, H, P, M.data(), cov.data(), pf.data(), pixelErrVar, 3 * pixelErrVar,
                                   pixelErrVar, slot.data(), m.data(), var.data(), dist.data(), flags.data());
    if (rcode != CS_OK) {
        fprintf(stderr, "cs_register_search: %s\n", cs_last_error());
        return 1;
    }
    int nCand = 0, nSkip[5] = {0, 0, 0, 0, 0};
    for (int p = 0; p < P; ++p)
        for (int c = 0; c < nCams; ++c) {
            const int o = p * nCams + c;
            if (pf[o] >= 0) {  // SL_CoSLAM.cpp:737-738
                CHECK(slot[o] == -1);
                nSkip[1]++;
                continue;
            }
            const double Z = M[3 * p + 2] + ts[3 * c + 2];
            if (Z < 0) {  // :740-742
                CHECK(slot[o] == -2);
                nSkip[2]++;
                continue;
            }
            const double mx = (M[3 * p] + ts[3 * c]) / Z, my = (M[3 * p + 1] + ts[3 * c + 1]) / Z;
            if (mx < 0 || mx >= W || my < 0 || my >= H) {  // :746-748
                CHECK(slot[o] == -3);
                nSkip[3]++;
                continue;
            }
            CHECK(slot[o] >= 0);
            CHECK(fabs(m[2 * o] - mx) < 1e-9 && fabs(m[2 * o + 1] - my) < 1e-9);
            // the reference's own search on the kernel's projection and covariance
            double mm[2] = {m[2 * o], m[2 * o + 1]}, vv[4] = {var[4 * o], var[4 * o + 1], var[4 * o + 2], var[4 * o + 3]};
            FeaturePoint* fp = searchMahaNearestFeatPt(cams[c].list, FRAME, mm, vv, 3 * pixelErrVar);  // :755-756
            CHECK(fp != 0);
            CHECK(cams[c].slotOf.at(fp) == slot[o]);
            CHECK(((flags[o] & 1) != 0) == (cams[c].slot2map[slot[o]] < 0));
            nCand++;
        }
    CHECK(nCand > 500 && nSkip[1] > 100 && nSkip[2] > 5 && nSkip[3] > 50);
    printf("ref_register_test: %d candidates agree with the reference's searchMahaNearestFeatPt; skipped %d / %d / %d "
           "(has feature / behind / outside)\n", nCand, nSkip[1], nSkip[2], nSkip[3]);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "golden")) return run_golden(argv[2]);
    return run_gpu();
}
