// ref_mergability_test.cpp -- the reference's OWN CoSLAM::staticCheckMergability (src/app/SL_CoSLAM.cpp:714-729) on feature tracks
// built with the reference's classes; writes the cases and its verdicts for tests/golden/make_golden.py (CPU only).
//
// oracle/Makefile compiles /root/reference/src/app/SL_CoSLAM.cpp IN PLACE (through the pipe that rewrites its three `pointer > 0`
// comparisons to `!= 0`) against oracle/ref_shim/.  A case = one map point (M, cov), one camera's intrinsics, a track of L
// feature points linked through FeaturePoint::preFrame, each with the pose of its frame (FeaturePoint::cam) -- clean tracks,
// tracks with one bad frame at the head / in the middle / at the tail, points with a large or a tiny covariance.
//   ref_mergability_test golden <out.bin>
// Layout of out.bin: int32 nCases; per case: int32 L; double sigma; K[9]; M[3]; cov[9]; L x (R[9], t[3], m[2]) newest first;
// int32 verdict.   TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "app/SL_CoSLAM.h"

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static double urand() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return (double)(g_rng >> 11) / 9007199254740992.0;
}
static double nrand() { return sqrt(-2 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

static void rodrigues(const double w[3], double R[9]) {
    const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double k[3] = {th > 0 ? w[0] / th : 0, th > 0 ? w[1] / th : 0, th > 0 ? w[2] / th : 0};
    const double c = cos(th), s = sin(th), v = 1 - c;
    const double M[9] = {c + k[0] * k[0] * v,        k[0] * k[1] * v - k[2] * s, k[0] * k[2] * v + k[1] * s,
                         k[1] * k[0] * v + k[2] * s, c + k[1] * k[1] * v,        k[1] * k[2] * v - k[0] * s,
                         k[2] * k[0] * v - k[1] * s, k[2] * k[1] * v + k[0] * s, c + k[2] * k[2] * v};
    memcpy(R, M, sizeof(M));
}

int main(int argc, char** argv) {
    if (argc < 3 || strcmp(argv[1], "golden")) {
        fprintf(stderr, "usage: %s golden <out.bin>\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    CoSLAM* co = new CoSLAM();
    const int nCases = 150;
    fwrite(&nCases, 4, 1, f);
    int nTrue = 0;
    for (int cs = 0; cs < nCases; ++cs) {
        const int L = 1 + (int)(urand() * 24);
        const double sigma = cs % 3 == 0 ? 10.0 : (cs % 3 == 1 ? 3.0 : 1.0);
        const double K[9] = {520 + 10 * urand(), 0, 320 + 4 * nrand(), 0, 518 + 10 * urand(), 240 + 4 * nrand(), 0, 0, 1};
        MapPoint mp(-2 + 4 * urand(), -1.5 + 3 * urand(), 6 + 6 * urand(), 0);
        double A[9];
        const double cscale = (cs % 5 == 0) ? 0.3 : 0.02;
        for (int q = 0; q < 9; ++q) A[q] = cscale * nrand();
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) mp.cov[3 * r + c] = A[3 * r] * A[3 * c] + A[3 * r + 1] * A[3 * c + 1] + A[3 * r + 2] * A[3 * c + 2] + (r == c ? 1e-6 : 0);
        // the frame where (if anywhere) the track leaves the point's projection, and by how much (in sigmas)
        const int badAt = (cs % 4 == 0) ? -1 : (int)(urand() * L);
        const double badBy = (cs % 7 == 0) ? 1.02 : (cs % 7 == 1 ? 0.98 : 3.0);   // just outside / just inside / far outside
        std::vector<CamPoseItem*> cams(L);
        std::vector<FeaturePoint*> fps(L);
        std::vector<double> rec;
        for (int j = 0; j < L; ++j) {   // j = 0: the newest frame
            double w[3] = {0.02 * nrand(), 0.02 * nrand() + 0.004 * j, 0.01 * nrand()}, R[9], t[3] = {0.05 * j + 0.01 * nrand(), 0.01 * nrand(), 0.01 * nrand()};
            rodrigues(w, R);
            cams[j] = new CamPoseItem();
            cams[j]->f = 100 - j, cams[j]->camId = 0;
            memcpy(cams[j]->R, R, 72), memcpy(cams[j]->t, t, 24);
            double X[3], m[2];
            for (int r = 0; r < 3; ++r) X[r] = R[3 * r] * mp.M[0] + R[3 * r + 1] * mp.M[1] + R[3 * r + 2] * mp.M[2] + t[r];
            m[0] = (K[0] * X[0] + K[1] * X[1] + K[2] * X[2]) / X[2];
            m[1] = (K[4] * X[1] + K[5] * X[2]) / X[2];
            // inside the gate: a fraction of sigma; at the bad frame: badBy x the gate's radius along a random direction
            // (the gate is the ellipse of J cov J^T + sigma^2 I: at least sigma in every direction, so badBy < 1 may still pass)
            const double a = 6.283185307179586 * urand();
            const double rad = (j == badAt) ? badBy * sigma * (1 + 2 * urand() * (cscale > 0.1)) : 0.3 * sigma * urand();
            m[0] += rad * cos(a), m[1] += rad * sin(a);
            fps[j] = new FeaturePoint(100 - j, 0, m[0], m[1]);
            fps[j]->setIntrinsic(K);
            fps[j]->setCameraPose(cams[j]);
            rec.insert(rec.end(), R, R + 9), rec.insert(rec.end(), t, t + 3), rec.insert(rec.end(), m, m + 2);
        }
        for (int j = 0; j + 1 < L; ++j) fps[j]->preFrame = fps[j + 1], fps[j + 1]->nextFrame = fps[j];
        fps[L - 1]->preFrame = nullptr;
        const int verdict = co->staticCheckMergability(&mp, fps[0], sigma) ? 1 : 0;
        nTrue += verdict;
        fwrite(&L, 4, 1, f), fwrite(&sigma, 8, 1, f), fwrite(K, 8, 9, f), fwrite(mp.M, 8, 3, f), fwrite(mp.cov, 8, 9, f);
        fwrite(rec.data(), 8, rec.size(), f), fwrite(&verdict, 4, 1, f);
        for (int j = 0; j < L; ++j) delete fps[j], delete cams[j];
    }
    fclose(f);
    printf("ref_mergability_test: %d cases, %d mergeable\n", nCases, nTrue);
    return (nTrue > 40 && nTrue < nCases - 40) ? 0 : 1;
}
