// ref_classify_test.cpp -- the reference's OWN CoSLAM::mapPointsClassify (src/app/SL_CoSLAM.cpp:418-520) with isStaticPoint,
// isStaticPointExclude, isDynamicPoint, isLittleMove, isStaticRemovable (src/slam/SL_CoSLAMHelper.cpp:67-330) on map points and
// feature tracks built with the reference's classes; writes the scenes and what the reference made of them for
// tests/golden/make_golden.py (CPU only).
//
// oracle/Makefile compiles /root/reference/src/app/SL_CoSLAM.cpp and src/slam/SL_CoSLAMHelper.cpp IN PLACE against oracle/ref_shim/
// (LibVisualSLAM's helpers are OUR definitions, ref_shim/ref_triangulate_impl.cpp + shim_impl.cpp: the vectors pin the state machine
// and the helpers' loops).  A scene = nCams cameras with H frames of poses, nPts map points on CoSLAM::curMapPts, per point and
// camera possibly a feature: of the current frame, or -- a camera that lost the point -- of a frame a few frames back, each with
// its track behind it.  The points are made to walk every branch: uncertain and new (truly static, old enough or not; moving;
// inconsistent), uncertain and old (moving; static with one camera's view grossly off -> that view is detached; inconsistent),
// dynamic (moving; standing still with the counter below / at the threshold of 50 frames -> back to static), seen by one camera,
// and certain static ones that are not looked at.
//   ref_classify_test golden <out.bin>
//   ref_classify_test golden_relink <out.bin>
// golden_relink: the features as the registration loops and lost tracks leave MapPoint::pFeatures over time (src/app/SL_CoSLAM.cpp:775-779,
// :997-1000: `pFeat->preFrame = p->pFeatures[iCam]`): per point and camera 1-3 segments of consecutive frames, newest first, gaps in
// between, the newest at the current frame or a few frames back (a stale head; also in the camera whose view is off, so that the view
// mapPointsClassify detaches can be a stale one), 70 frames of history so that chains reach behind isStaticPoint's window of 60 frames.
// Same layout except per point and camera: int32 nSeg, featDynamic, per segment int32 j0 (history entry of its newest node), L, L x m[2].
// Layout of out.bin: int32 nScenes; per scene: int32 nCams, H, nPts, curFrame; double pixelVar; per camera K[9], iK[9]; per camera
// and history entry (newest first) R[9], t[3]; per point: M[3], cov[9], int32 localType, uncertain, newPt, staticFrameNum,
// firstFrame, per camera int32 L (0 = no feature), int32 back (the feature's frame = curFrame - back), int32 featDynamic,
// L x m[2] (newest first); then per point what the reference left: M[3], cov[9], int32 localType, uncertain, newPt, staticFrameNum,
// per camera int32 hasFeature, int32 featDynamic.   TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "app/SL_CoSLAM.h"
#include "app/SL_GlobParam.h"

void getInvK(const double* K, double* iK);

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

template <class T> static void put(FILE* f, const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }
static void puti(FILE* f, int v) { fwrite(&v, 4, 1, f); }

int main(int argc, char** argv) {
    if (argc < 3 || (strcmp(argv[1], "golden") && strcmp(argv[1], "golden_relink"))) {
        fprintf(stderr, "usage: %s golden|golden_relink <out.bin>\n", argv[0]);
        return 2;
    }
    const bool relink = !strcmp(argv[1], "golden_relink");
    if (relink) g_rng = 0xD1B54A32D192ED03ull;
    int nChains = 0, nStale = 0, nLinked = 0, nStaleOff = 0;
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    const int nScenes = 3;
    puti(f, nScenes);
    int outcome[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 0 untouched certain, 1 -> static, 2 -> dynamic, 3 -> false, 4 stays uncertain, 5 detached, 6 counter++, 7 back to static
    for (int sc = 0; sc < nScenes; ++sc) {
        const int nCams = 3 + sc % 3, H = relink ? 70 : 64, nPts = 72, curFrame = 300 + 7 * sc;
        const double pixelVar = 12.0;  // CoSLAM::poseUpdate: mapPointsClassify(12.0) (src/app/SL_CoSLAM.cpp:385)
        CoSLAM* co = new CoSLAM();
        co->numCams = nCams;
        co->curFrame = curFrame;
        puti(f, nCams), puti(f, H), puti(f, nPts), puti(f, curFrame);
        put(f, &pixelVar, 1);
        std::vector<std::vector<double> > Ks(nCams, std::vector<double>(9));
        for (int c = 0; c < nCams; ++c) {
            const double K[9] = {515 + 15 * urand(), 0.0, 320 + 4 * nrand(), 0, 512 + 15 * urand(), 240 + 4 * nrand(), 0, 0, 1};
            double iK[9];
            getInvK(K, iK);
            memcpy(Ks[c].data(), K, 72);
            put(f, K, 9), put(f, iK, 9);
        }
        std::vector<std::vector<CamPoseItem*> > cams(nCams, std::vector<CamPoseItem*>(H));
        for (int c = 0; c < nCams; ++c)
            for (int j = 0; j < H; ++j) {  // j = 0: the current frame
                double w[3] = {0.01 * c + 0.001 * nrand(), 0.10 * c - 0.0015 * j, 0.0005 * j}, R[9];
                rodrigues(w, R);
                const double pos[3] = {1.6 * c - 0.035 * j + 0.002 * nrand(), 0.05 * c + 0.004 * j, -0.01 * j};
                double t[3];
                for (int r = 0; r < 3; ++r) t[r] = -(R[3 * r] * pos[0] + R[3 * r + 1] * pos[1] + R[3 * r + 2] * pos[2]);
                cams[c][j] = new CamPoseItem();
                cams[c][j]->f = curFrame - j, cams[c][j]->camId = c;
                memcpy(cams[c][j]->R, R, 72), memcpy(cams[c][j]->t, t, 24);
                put(f, R, 9), put(f, t, 3);
            }
        std::vector<MapPoint*> pts(nPts);
        for (int p = 0; p < nPts; ++p) {
            const int kind = p % 12;
            // 0 certain static | 1 uncertain new static old | 2 uncertain new static young | 3 uncertain new moving | 4 uncertain new garbage
            // 5 uncertain old moving | 6 uncertain old, one camera off | 7 uncertain old garbage | 8 dynamic moving | 9 dynamic still, counter low
            // 10 dynamic still, counter at 50 | 11 one camera only (uncertain)
            const bool moving = kind == 3 || kind == 5 || kind == 8;
            const bool garbage = kind == 4 || kind == 7;
            const double X0[3] = {-1.5 + 6 * urand(), -1.5 + 3 * urand(), 8 + 5 * urand()};
            const double vel[3] = {moving ? 0.05 * (urand() < 0.5 ? -1 : 1) : 0, moving ? 0.02 * nrand() : 0, moving ? 0.02 * nrand() : 0};
            MapPoint* mp = new MapPoint(X0[0] + 0.03 * nrand(), X0[1] + 0.03 * nrand(), X0[2] + 0.06 * nrand(), curFrame - 45);
            double A[9];
            for (int q = 0; q < 9; ++q) A[q] = 0.04 * nrand();
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) mp->cov[3 * r + c] = A[3 * r] * A[3 * c] + A[3 * r + 1] * A[3 * c + 1] + A[3 * r + 2] * A[3 * c + 2] + (r == c ? 1e-5 : 0);
            if (kind >= 8 && kind <= 10) mp->setLocalDynamic(); else mp->setLocalStatic();
            if ((kind >= 1 && kind <= 7) || kind == 11) mp->setUncertain();
            if (kind == 8 && p % 24 == 8) mp->setUncertain();   // a dynamic point that is uncertain too goes the uncertain way
            mp->bNewPt = (kind >= 1 && kind <= 4) || kind == 11;
            mp->firstFrame = kind == 2 ? curFrame - 12 : curFrame - 45;
            mp->staticFrameNum = kind == 10 ? 50 : (kind == 9 ? (int)(urand() * 40) : 0);
            mp->lastFrame = curFrame;
            put(f, mp->M, 3), put(f, mp->cov, 9);
            puti(f, mp->iLocalType), puti(f, mp->bUncertain ? 1 : 0), puti(f, mp->bNewPt ? 1 : 0), puti(f, mp->staticFrameNum), puti(f, mp->firstFrame);
            const int offCam = (kind == 6) ? (int)(urand() * nCams) : -1;   // the camera whose view is grossly off
            int nCur = 0;
            for (int c = 0; c < nCams; ++c) {
                // which cameras see the point: kind 11 one camera; kind 6 all (the removable test needs more than two); else most
                bool has = kind == 11 ? (c == p % nCams) : (kind == 6 ? true : (urand() < 0.85 || c < 2));
                int back = (has && kind != 11 && c >= 2 && urand() < 0.2) ? 1 + (int)(urand() * 8) : 0;   // a camera that lost the point
                if (relink && has && kind == 6 && c == offCam && c >= 2 && urand() < 0.5) back = 1 + (int)(urand() * 8);
                int L = has ? 2 + (int)(urand() * (H - 2 - back)) : 0;
                const int dyn = (kind >= 8 && kind <= 10) ? (urand() < 0.7) : (urand() < 0.1);
                std::vector<std::pair<int, int> > segs;   // (history entry of the newest node, nodes)
                if (!relink) {
                    puti(f, L), puti(f, back), puti(f, dyn);
                    if (L > 0) segs.push_back(std::make_pair(back, L));
                } else if (has) {
                    const double u2 = urand();
                    const int want = u2 < 0.3 ? 1 : (u2 < 0.7 ? 2 : 3);
                    int j = back;
                    for (int q = 0; q < want && j < H; ++q) {
                        int Lq = 1 + (int)(urand() * (q == 0 && want > 1 ? 12 : 30));   // (a freshly re-linked head is short)
                        if (q == 0 && want > 1 && urand() < 0.25) Lq = 1;
                        if (j + Lq > H) Lq = H - j;
                        segs.push_back(std::make_pair(j, Lq));
                        j += Lq + 1 + (int)(urand() * 10);   // the frames in which the camera did not see the point
                    }
                    L = 1;
                    puti(f, (int)segs.size()), puti(f, dyn);
                    ++nChains, nStale += back > 0, nLinked += segs.size() > 1, nStaleOff += (c == offCam && back > 0);
                } else {
                    L = 0;
                    puti(f, 0), puti(f, dyn);
                }
                FeaturePoint* newer = nullptr;
                for (size_t sg = 0; sg < segs.size(); ++sg) {
                    if (relink) puti(f, segs[sg].first), puti(f, segs[sg].second);
                    for (int j = segs[sg].first; j < segs[sg].first + segs[sg].second; ++j) {   // j: history entry of this feature's frame
                        const double* R = cams[c][j]->R;
                        const double* t = cams[c][j]->t;
                        const double* K = Ks[c].data();
                        const double X[3] = {X0[0] - vel[0] * j, X0[1] - vel[1] * j, X0[2] - vel[2] * j};
                        double Xc[3], m[2];
                        for (int r = 0; r < 3; ++r) Xc[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r];
                        m[0] = (K[0] * Xc[0] + K[1] * Xc[1] + K[2] * Xc[2]) / Xc[2] + 0.4 * nrand();
                        m[1] = (K[4] * Xc[1] + K[5] * Xc[2]) / Xc[2] + 0.4 * nrand();
                        if (garbage) m[0] += 60 * nrand(), m[1] += 60 * nrand();
                        if (c == offCam) m[0] += 45, m[1] -= 38;
                        put(f, m, 2);
                        FeaturePoint* fp = new FeaturePoint(curFrame - j, c, m[0], m[1]);
                        fp->setIntrinsic(K);
                        fp->setCameraPose(cams[c][j]);
                        fp->type = dyn ? TYPE_FEATPOINT_DYNAMIC : TYPE_FEATPOINT_STATIC;
                        fp->preFrame = nullptr;
                        fp->mpt = mp;
                        if (newer) newer->preFrame = fp, fp->nextFrame = newer;   // (across a gap: what :777-778 assigns)
                        else mp->pFeatures[c] = fp;
                        newer = fp;
                    }
                }
                if (L > 0 && back == 0) ++nCur;
            }
            mp->numVisCam = nCur;   // MapPoint::updateVisCamNum(curFrame), as mapStateUpdate leaves it (:1183)
            pts[p] = mp;
            if (nCur > 0) co->curMapPts.add(mp);   // (a point no camera sees in this frame has left the current list)
        }
        co->mapPointsClassify(pixelVar);
        for (int p = 0; p < nPts; ++p) {
            MapPoint* mp = pts[p];
            put(f, mp->M, 3), put(f, mp->cov, 9);
            puti(f, mp->iLocalType), puti(f, mp->bUncertain ? 1 : 0), puti(f, mp->bNewPt ? 1 : 0), puti(f, mp->staticFrameNum);
            for (int c = 0; c < nCams; ++c) {
                puti(f, mp->pFeatures[c] ? 1 : 0);
                puti(f, mp->pFeatures[c] ? (mp->pFeatures[c]->type == TYPE_FEATPOINT_DYNAMIC) : 0);
            }
            const int kind = p % 12;
            for (int q = 0; q < 3; ++q)
                if (!(fabs(mp->M[q]) < 1e4)) return 3;   // the scenes are built so that no triangulation is degenerate
            if (kind == 0) ++outcome[0];
            else if (mp->iLocalType == TYPE_MAP_FALSE) ++outcome[3];
            else if (mp->bUncertain) ++outcome[4];
            else if (mp->iLocalType == TYPE_MAP_DYNAMIC) ++outcome[(kind >= 9 && kind <= 10) ? 6 : 2];
            else ++outcome[kind == 6 ? 5 : (kind == 10 ? 7 : 1)];
        }
        co->curMapPts.clearWithoutRelease();
    }
    fclose(f);
    if (relink) printf("ref_classify_test golden_relink: %d chains, %d with a stale head (%d of them the view that is off), %d with linked segments\n", nChains, nStale, nStaleOff, nLinked);
    printf("ref_classify_test: certain %d, -> static %d, -> dynamic %d, -> false %d, still uncertain %d, view detached %d, dynamic kept %d, back to static %d\n",
           outcome[0], outcome[1], outcome[2], outcome[3], outcome[4], outcome[5], outcome[6], outcome[7]);
    for (int q = 0; q < 8; ++q)
        if (outcome[q] < 5) return 1;   // every branch must have been walked
    return 0;
}
