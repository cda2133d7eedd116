// ref_ncc_test.cpp -- the reference's OWN NCC code against the NCC kernels.
//
// oracle/Makefile compiles /root/reference/src/slam/SL_NCCBlock.cpp (NCCBlock::compute / computeScaled :15-54, matchNCCBlock
// :258-264) and src/slam/SL_FeatureMatching.cpp (getEpiNccMat :3-46) IN PLACE against oracle/ref_shim/ (epipolarError is
// un-vendored LibVisualSLAM: stand-in; the OpenCV names of the functions that are not under test are declarations only and
// dropped by --gc-sections).  This driver cuts blocks with the reference's NCCBlock, builds the reference's containers
// (PtrVec<NCCBlock>, Mat_d) and calls the reference's getEpiNccMat.
//   ref_ncc_test golden <out.bin>   CPU only: image, features, F and the reference's blocks / A B C / matrices --
//                                    tests/golden/make_golden.py turns them into tests/golden/ncc_golden.npz.
//   ref_ncc_test                     MI355X: cs_ncc_match_between (libcoslam_hip.so) on the same inputs; blocks, A / B / C and
//                                    both matrices must equal the reference's bit for bit.
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists, run by tests/test_cxx_dropin_gpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "slam/SL_FeatureMatching.h"
#include "slam/SL_NCCBlock.h"

#include "coslam_hip.h"

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                    \
        }                                                                \
    } while (0)

static unsigned long long g_rng = 0xD1B54A32D192ED03ull;
static double urand() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return (double)(g_rng >> 11) / 9007199254740992.0;
}

struct Side {
    ImgG img;                        // the camera's small image
    std::vector<double> x, y;        // feature positions in the full image
    std::vector<unsigned char> blk;  // n x 128
    std::vector<double> abc;         // n x 4
    std::vector<int> valid;
    PtrVec<NCCBlock> ref;            // the reference's blocks
    Mat_d pts;                       // n x 2, as getEpiNccMat reads them
};

static const double SCALE = 0.3;  // SingleSLAM::m_smallScale, src/app/SL_SingleSLAM.cpp:30

static void build(Side& s, int W, int H, int n, int salt) {
    s.img.resize(W, H);
    for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {  // textured, with flat patches (C = inf there) and saturated ones
            double v = 128 + 90 * sin(0.37 * xx + 0.11 * salt) * cos(0.23 * yy) + 30 * (urand() - 0.5);
            if ((xx / 24 + yy / 24) % 7 == 0) v = 200;
            s.img.data[yy * W + xx] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    s.x.resize(n), s.y.resize(n), s.blk.assign((size_t)n * 128, 0x80), s.abc.assign((size_t)n * 4, 0.0), s.valid.assign(n, 0);
    s.ref.reserve(n);
    s.pts.resize(n, 2);
    for (int i = 0; i < n; ++i) {
        s.x[i] = urand() * (W / SCALE + 20) - 10;  // some outside: compute() refuses blocks that leave the image
        s.y[i] = urand() * (H / SCALE + 20) - 10;
        s.pts.data[2 * i] = s.x[i];
        s.pts.data[2 * i + 1] = s.y[i];
        NCCBlock* b = new NCCBlock(0);
        const bool ok = b->computeScaled(s.img, SCALE, s.x[i], s.y[i]);  // the reference's own block
        s.valid[i] = ok ? 1 : 0;
        if (ok) {
            memcpy(&s.blk[(size_t)i * 128], b->I, SL_NCCBLK_LEN);
            s.abc[4 * i] = b->A, s.abc[4 * i + 1] = b->B, s.abc[4 * i + 2] = b->C, s.abc[4 * i + 3] = b->avgI;
        } else {  // give the reference's matcher something defined: a flat block never matches (C = inf -> NaN score)
            memset(b->I, 0, SL_NCCBLK_LEN);
            b->A = b->B = 0;
            b->C = 1 / sqrt(0.0);
        }
        s.ref.push_back(b);
    }
}

static void fundamental(double* F) {  // a plausible F: skew(e) * (I + small)
    const double e[3] = {400.0, -80.0, 1.0};
    const double A[9] = {1.0, 0.02, -12.0, -0.015, 0.98, 7.0, 1e-5, -2e-5, 1.0};
    const double S[9] = {0, -e[2], e[1], e[2], 0, -e[0], -e[1], e[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) F[3 * i + j] = (S[3 * i] * A[j] + S[3 * i + 1] * A[3 + j] + S[3 * i + 2] * A[6 + j]) * 1e-3;
}

static int run(bool golden, const char* path) {
    const int W = 192, H = 144, M = golden ? 150 : 700, N = golden ? 170 : 900;
    Side a, b;
    build(a, W, H, M, 1);
    build(b, W, H, N, 2);
    double F[9];
    fundamental(F);
    const double epiMax = 50, nccMin = 0.3;  // NewMapPtsNCCParam (:60-70) has 50 / 0.8; 0.3 keeps more pairs in the test
    Mat_d epiRef, nccRef;
    getEpiNccMat(F, a.pts, b.pts, a.ref, b.ref, epiMax, nccMin, epiRef, nccRef);  // the reference (wNone = -1)
    // pairs with a missing block: the kernel reports wNone; make the reference side agree (its blocks there are stand-ins)
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j)
            if (!a.valid[i] || !b.valid[j]) epiRef.data[(size_t)i * N + j] = nccRef.data[(size_t)i * N + j] = -1;
    size_t kept = 0;
    for (size_t q = 0; q < (size_t)M * N; ++q) kept += nccRef.data[q] != -1;
    if (golden) {
        FILE* f = fopen(path, "wb");
        if (!f) return 2;
        const int hdr[4] = {W, H, M, N};
        fwrite(hdr, sizeof(int), 4, f);
        fwrite(F, sizeof(double), 9, f);
        const double par[3] = {SCALE, epiMax, nccMin};
        fwrite(par, sizeof(double), 3, f);
        for (Side* s : {&a, &b}) {
            const int n = (int)s->x.size();
            fwrite(s->img.data, 1, (size_t)W * H, f);
            fwrite(s->x.data(), sizeof(double), n, f);
            fwrite(s->y.data(), sizeof(double), n, f);
            fwrite(s->blk.data(), 1, (size_t)n * 128, f);
            fwrite(s->abc.data(), sizeof(double), (size_t)n * 4, f);
            fwrite(s->valid.data(), sizeof(int), n, f);
        }
        fwrite(epiRef.data, sizeof(double), (size_t)M * N, f);
        fwrite(nccRef.data, sizeof(double), (size_t)M * N, f);
        fclose(f);
        printf("ref_ncc_test: wrote %d x %d pairs, %zu kept by the reference\n", M, N, kept);
        return 0;
    }
    std::vector<double> epi((size_t)M * N), ncc((size_t)M * N), c1((size_t)M * 4), c2((size_t)N * 4);
    std::vector<unsigned char> b1((size_t)M * 128), b2((size_t)N * 128);
    std::vector<int> v1(M), v2(N);
    int rc = cs_ncc_match_between(0, a.img.data, W, H, M, a.x.data(), a.y.data(), b.img.data, W, H, N, b.x.data(), b.y.data(), SCALE, F,
                                  epiMax, nccMin, -1.0, epi.data(), ncc.data(), b1.data(), c1.data(), v1.data(), b2.data(), c2.data(),
                                  v2.data());
    if (rc != CS_OK) {
        fprintf(stderr, "cs_ncc_match_between: %s\n", cs_last_error());
        return 1;
    }
    CHECK(v1 == a.valid && v2 == b.valid);
    CHECK(b1 == a.blk && b2 == b.blk);
    CHECK(memcmp(c1.data(), a.abc.data(), c1.size() * sizeof(double)) == 0);
    CHECK(memcmp(c2.data(), b.abc.data(), c2.size() * sizeof(double)) == 0);
    CHECK(memcmp(epi.data(), epiRef.data, epi.size() * sizeof(double)) == 0);
    CHECK(memcmp(ncc.data(), nccRef.data, ncc.size() * sizeof(double)) == 0);
    int nInvalid = 0;
    for (int v : v1) nInvalid += !v;
    CHECK(kept > 1000 && nInvalid > 5);
    printf("ref_ncc_test: blocks, A / B / C and the %d x %d epipolar / NCC matrices equal the reference's bit for bit (%zu pairs kept, "
           "%d blocks refused)\n", M, N, kept, nInvalid);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "golden")) return run(true, argv[2]);
    return run(false, nullptr);
}
