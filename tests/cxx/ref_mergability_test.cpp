// ref_mergability_test.cpp -- the reference's OWN CoSLAM::staticCheckMergability (src/app/SL_CoSLAM.cpp:714-729) on feature tracks
// built with the reference's classes; writes the cases and its verdicts for tests/golden/make_golden.py (CPU only).
//
// oracle/Makefile compiles /root/reference/src/app/SL_CoSLAM.cpp IN PLACE (through the pipe that rewrites its three `pointer > 0`
// comparisons to `!= 0`) against oracle/ref_shim/.  A case = one map point (M, cov), one camera's intrinsics, a track of L
// feature points linked through FeaturePoint::preFrame, each with the pose of its frame (FeaturePoint::cam) -- clean tracks,
// tracks with one bad frame at the head / in the middle / at the tail, points with a large or a tiny covariance.
//   ref_mergability_test golden <out.bin>        (golden_long <out.bin>: long tracks, see golden_long below)
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

// golden_long: LONG tracks (most of them 200-420 frames) laid out the way the device holds them -- nCams pose sequences of T frames
// (frame f of camera c has ONE pose; every track of the camera shares it), per camera nTracks tracks, each with its own map point,
// first frame f1 and pixels for the frames f1 .. T - 1 -- and the verdict of the reference's own staticCheckMergability on the
// chain fp(T - 1) -> preFrame -> ... -> fp(f1).  Cases: clean tracks; ONE frame outside the gate at the newest frame / inside the
// newest 64 / just beyond them / deep in the tail / at the first frame, just outside, just inside or far outside; short tracks.
//   layout: int32 nCams, T, nTracks; double sigma; per camera: K[9], T x (R[9], t[3]); per camera per track: M[3], cov[9], int32 f1,
//   (T - f1) x m[2] oldest first, int32 badFrame (-1 none), int32 verdict
static int golden_long(const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    CoSLAM* co = new CoSLAM();
    const int nCams = 3, T = 420, nTracks = 48;
    const double sigma = 10.0;
    fwrite(&nCams, 4, 1, f), fwrite(&T, 4, 1, f), fwrite(&nTracks, 4, 1, f), fwrite(&sigma, 8, 1, f);
    std::vector<std::vector<CamPoseItem*>> cams(nCams);
    std::vector<std::vector<double>> Ks(nCams);
    for (int c = 0; c < nCams; ++c) {
        const double K[9] = {520 + 10 * urand(), 0, 320 + 4 * nrand(), 0, 518 + 10 * urand(), 240 + 4 * nrand(), 0, 0, 1};
        Ks[c].assign(K, K + 9);
        fwrite(K, 8, 9, f);
        // a smooth closed path (what a hand-held camera looking at one scene does over 400 frames): the point stays in view
        const double ph0 = 6.283185307179586 * urand();
        for (int fr = 0; fr < T; ++fr) {
            const double ph = ph0 + 6.283185307179586 * fr / 160.0;
            double w[3] = {0.03 * sin(ph) + 0.002 * nrand(), 0.05 * cos(0.7 * ph) + 0.002 * nrand(), 0.02 * sin(1.3 * ph)}, R[9];
            double t[3] = {0.35 * sin(ph) + 0.003 * nrand(), 0.1 * cos(ph), 0.15 * (1 - cos(ph)) + 0.003 * nrand()};
            rodrigues(w, R);
            CamPoseItem* cp = new CamPoseItem();
            cp->f = fr, cp->camId = c;
            memcpy(cp->R, R, 72), memcpy(cp->t, t, 24);
            cams[c].push_back(cp);
            fwrite(R, 8, 9, f), fwrite(t, 8, 3, f);
        }
    }
    int nTrue = 0, nAll = 0;
    for (int c = 0; c < nCams; ++c)
        for (int k = 0; k < nTracks; ++k) {
            const double* K = Ks[c].data();
            MapPoint mp(-1.5 + 3 * urand(), -1.0 + 2 * urand(), 7 + 5 * urand(), 0);
            double A[9];
            const double cscale = (k % 6 == 0) ? 0.2 : 0.02;
            for (int q = 0; q < 9; ++q) A[q] = cscale * nrand();
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc)
                    mp.cov[3 * r + cc] = A[3 * r] * A[3 * cc] + A[3 * r + 1] * A[3 * cc + 1] + A[3 * r + 2] * A[3 * cc + 2] + (r == cc ? 1e-6 : 0);
            // length: mostly 200 .. T; every eighth track short (inside the 64-frame window); one full-length
            int L = (k % 8 == 7) ? 5 + (int)(urand() * 55) : 200 + (int)(urand() * (T - 200));
            if (k == 0) L = T;
            if (k == 1) L = 65;   // one frame of tail
            if (k == 2) L = 64;   // exactly the window
            const int f1 = T - L;
            // the one bad frame, as a walk depth j (0 = newest)
            int badJ = -1;
            switch (k % 6) {
                case 0: badJ = -1; break;                                   // clean
                case 1: badJ = (int)(urand() * (L < 64 ? L : 64)); break;    // inside the window
                case 2: badJ = L > 66 ? 64 + (int)(urand() * 3) : -1; break;  // just beyond the window
                case 3: badJ = L > 70 ? 64 + (int)(urand() * (L - 64)) : -1; break;  // anywhere in the tail
                case 4: badJ = L - 1; break;                                 // the track's first frame
                case 5: badJ = 0; break;                                     // this frame
            }
            const double badBy = (k % 5 == 0) ? 1.03 : (k % 5 == 1 ? 0.97 : 2.5);
            const int badFrame = badJ >= 0 ? T - 1 - badJ : -1;
            std::vector<FeaturePoint*> fps;
            std::vector<double> ms;
            for (int fr = f1; fr < T; ++fr) {
                const CamPoseItem* cp = cams[c][fr];
                double X[3], m[2];
                for (int r = 0; r < 3; ++r) X[r] = cp->R[3 * r] * mp.M[0] + cp->R[3 * r + 1] * mp.M[1] + cp->R[3 * r + 2] * mp.M[2] + cp->t[r];
                m[0] = (K[0] * X[0] + K[1] * X[1] + K[2] * X[2]) / X[2];
                m[1] = (K[4] * X[1] + K[5] * X[2]) / X[2];
                const double a = 6.283185307179586 * urand();
                const double rad = (fr == badFrame) ? badBy * sigma * (1 + 2 * urand() * (cscale > 0.1)) : 0.35 * sigma * urand();
                m[0] += rad * cos(a), m[1] += rad * sin(a);
                FeaturePoint* fp = new FeaturePoint(fr, c, m[0], m[1]);
                fp->setIntrinsic(K);
                fp->setCameraPose(cams[c][fr]);
                if (!fps.empty()) fp->preFrame = fps.back(), fps.back()->nextFrame = fp;
                fps.push_back(fp);
                ms.push_back(m[0]), ms.push_back(m[1]);
            }
            const int verdict = co->staticCheckMergability(&mp, fps.back(), sigma) ? 1 : 0;
            nTrue += verdict, ++nAll;
            fwrite(mp.M, 8, 3, f), fwrite(mp.cov, 8, 9, f), fwrite(&f1, 4, 1, f), fwrite(ms.data(), 8, ms.size(), f);
            fwrite(&badFrame, 4, 1, f), fwrite(&verdict, 4, 1, f);
            for (FeaturePoint* fp : fps) delete fp;
        }
    fclose(f);
    printf("ref_mergability_test golden_long: %d tracks, %d mergeable\n", nAll, nTrue);
    return (nTrue > nAll / 6 && nTrue < nAll - nAll / 6) ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "golden_long")) return golden_long(argv[2]);
    if (argc < 3 || strcmp(argv[1], "golden")) {
        fprintf(stderr, "usage: %s golden|golden_long <out.bin>\n", argv[0]);
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
