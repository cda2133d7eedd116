// ref_intracam_newpts_test.cpp -- the reference's OWN SingleSLAM::newMapPoints (src/app/SL_SingleSLAM.cpp:922-1004: what
// CoSLAM::genNewMapPoints calls for a camera that is ready for a key frame, src/app/SL_CoSLAM.cpp:1310-1330) with
// getUnMappedAndTrackedFeatPts (:152-172) and refineTriangulation (:1005-1049), compiled in place, on tracks built with the reference's
// classes (Track2D / Track2DNode / FeaturePoint / CamPoseItem).  Writes the scenes and what the reference made of them for
// tests/golden/make_golden.py (CPU only).
//   ref_intracam_newpts_test golden <out.bin>
// A scene = ONE camera with H frames of poses (newest first), N slots each with a track of L frames ending at the current frame (or no
// track); tracks: shorter than Param::nMinFeatTrkLen (20: skipped), already mapped (skipped), dynamic (skipped), of a point in front of a
// camera that moved sideways (a new map point), with too little parallax (the covariance test throws it out), with a pixel off the
// point's projection in one view (the re-projection test throws it out), behind the camera.
// Layout: int32 nScenes; per scene: int32 H, N, curFrame; double K[9], iK[9], sigma, maxEpiErr; int32 minTrackLen; H x (R[9], t[3]);
// per slot: int32 L (0: empty), mapped, dynamic, L x m[2] (newest first); then int32 nNew and per new point (in the order the reference
// made them: slot order): int32 slot, firstFrame, double M[3], cov[9].
// LibVisualSLAM's binTriangulate / getBinTriangulateCovMat / reprojErrorSingle / isAtCameraBack / triangulateMultiView are OUR
// definitions (ref_shim/ref_triangulate_impl.cpp): the vectors pin the loop -- which tracks are taken, which two views,
// which tests in which order, the refinement's view.  TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "app/SL_CoSLAM.h"
#include "app/SL_GlobParam.h"

void getInvK(const double* K, double* iK);

static unsigned long long g_rng = 0xD1B54A32D192ED03ull;
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
template <class T> static void put(FILE* f, const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }
static void puti(FILE* f, int v) { fwrite(&v, 4, 1, f); }

int main(int argc, char** argv) {
    if (argc < 3 || strcmp(argv[1], "golden")) {
        fprintf(stderr, "usage: %s golden <out.bin>\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    const int nScenes = 3;
    puti(f, nScenes);
    int nNewAll = 0, nCand = 0, nTracks = 0;
    for (int sc = 0; sc < nScenes; ++sc) {
        const int H = 40 + 15 * sc, N = 160, curFrame = 500 + 13 * sc, W = 640, Hh = 480;
        SingleSLAM* s = new SingleSLAM();
        s->camId = sc % 3;
        const double K[9] = {520 + 10 * sc, 0, 320, 0, 515 + 10 * sc, 240, 0, 0, 1};
        double iK[9];
        getInvK(K, iK);
        s->K.cloneFrom(K, 3, 3), s->iK.cloneFrom(iK, 3, 3);
        s->m_tracker.m_frame = curFrame;
        s->m_tracker.m_nMaxCorners = N;
        s->m_tracker.m_tks = new Track2D[N];
        s->m_rgb.resize(W, Hh);
        memset(s->m_rgb.data, 90, (size_t)W * Hh * 3);
        KeyPose* kp = new KeyPose();
        kp->imgSmall.resize(192, 144);
        memset(kp->imgSmall.data, 100, 192 * 144);
        kp->imgScale = 0.3;
        s->m_lastKeyPos = kp;
        const double sigma = Const::PIXEL_ERR_VAR, maxEpiErr = 2.0;
        puti(f, H), puti(f, N), puti(f, curFrame);
        put(f, K, 9), put(f, iK, 9), put(f, &sigma, 1), put(f, &maxEpiErr, 1);
        puti(f, Param::nMinFeatTrkLen);
        std::vector<CamPoseItem*> cams(H);
        for (int j = H - 1; j >= 0; --j) {   // oldest first; j = 0: the current frame.  A camera moving sideways, slowly turning
            double w[3] = {0.004 * nrand(), 0.0025 * j + 0.002 * nrand(), 0.001 * nrand()}, R[9];
            rodrigues(w, R);
            const double pos[3] = {-0.045 * j * (sc == 2 ? 0.15 : 1.0), 0.004 * j, 0.006 * j};   // (scene 2: hardly any baseline)
            double t[3];
            for (int r = 0; r < 3; ++r) t[r] = -(R[3 * r] * pos[0] + R[3 * r + 1] * pos[1] + R[3 * r + 2] * pos[2]);
            cams[j] = s->m_camPos.add(curFrame - j, s->camId, R, t);
        }
        for (int j = 0; j < H; ++j) put(f, cams[j]->R, 9), put(f, cams[j]->t, 3);
        MapPoint* someone = new MapPoint(0, 0, 5, curFrame - 30);
        for (int k = 0; k < N; ++k) {
            Track2D& tk = s->m_tracker.m_tks[k];
            tk.id = k;
            const double u = urand();
            int L = u < 0.1 ? 0 : (u < 0.25 ? 3 + (int)(urand() * 17) : 21 + (int)(urand() * (H - 21)));
            if (L > H) L = H;
            const int mapped = L > 0 && urand() < 0.12, dyn = L > 0 && urand() < 0.1;
            const int kind = (int)(urand() * 10);   // 0: behind the camera, 1: one view 30 px off, 2: far away (parallax too small), else: fine
            double X[3] = {-2.5 + 5 * urand(), -1.5 + 3 * urand(), 4 + 8 * urand()};
            if (kind == 0) X[2] = -3 - 3 * urand();
            if (kind == 2) X[2] = 400 + 300 * urand();
            puti(f, L), puti(f, mapped), puti(f, dyn);
            FeaturePoint* newer = nullptr;
            std::vector<FeaturePoint*> fps(L);
            for (int j = 0; j < L; ++j) {
                const double* R = cams[j]->R;
                const double* t = cams[j]->t;
                double Xc[3], m[2];
                for (int r = 0; r < 3; ++r) Xc[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r];
                m[0] = (K[0] * Xc[0] + K[1] * Xc[1] + K[2] * Xc[2]) / Xc[2] + 0.3 * nrand();
                m[1] = (K[4] * Xc[1] + K[5] * Xc[2]) / Xc[2] + 0.3 * nrand();
                if (kind == 1 && j == L - 1) m[0] += 30;
                put(f, m, 2);
                FeaturePoint* fp = new FeaturePoint(curFrame - j, s->camId, m[0], m[1]);
                fp->setIntrinsic(K);
                fp->setCameraPose(cams[j]);
                fp->type = dyn ? TYPE_FEATPOINT_DYNAMIC : TYPE_FEATPOINT_STATIC;
                fp->mpt = mapped ? someone : nullptr;
                fp->preFrame = nullptr;
                if (newer) newer->preFrame = fp, fp->nextFrame = newer;
                newer = fp;
                fps[j] = fp;
            }
            for (int j = L - 1; j >= 0; --j) tk.add(new Track2DNode(fps[j]));   // oldest first: f1 = the track's first frame, tail = this frame's node
            if (L > 0) ++nTracks;
            if (L >= 21 && !mapped && !dyn) ++nCand;
        }
        std::vector<MapPoint*> fresh;
        const int n = s->newMapPoints(fresh, maxEpiErr, 0.75);
        if (n != (int)fresh.size()) return 3;
        puti(f, n);
        for (MapPoint* p : fresh) {
            FeaturePoint* fp = p->pFeatures[s->camId];
            int slot = -1;
            for (int k = 0; k < N; ++k)
                if (!s->m_tracker.m_tks[k].empty() && s->m_tracker.m_tks[k].tail->pt == fp) slot = k;
            puti(f, slot), puti(f, p->firstFrame);
            put(f, p->M, 3), put(f, p->cov, 9);
            if (slot < 0 || !p->isLocalStatic() || p->lastFrame != curFrame) return 4;
        }
        nNewAll += n;
    }
    fclose(f);
    printf("ref_intracam_newpts_test: %d scenes, %d tracks, %d long unmapped static ones, %d new map points\n", nScenes, nTracks, nCand, nNewAll);
    return (nNewAll > 40 && nNewAll < nCand) ? 0 : 1;
}
