// ref_update_points_test.cpp -- the reference's OWN RobustBundleRTS::updateNewPosesPoints (src/app/SL_CoSLAMRobustBA.cpp:248-271)
// with updateStaticPointPosition / updateDynamicPointPosition (src/slam/SL_CoSLAMHelper.cpp:338-394, 455-484) on map points and
// feature tracks built with the reference's classes; writes the scenes and what the reference made of them for
// tests/golden/make_golden.py (CPU only).
//
// oracle/Makefile compiles /root/reference/src/slam/SL_CoSLAMHelper.cpp and src/app/SL_CoSLAMRobustBA.cpp IN PLACE against
// oracle/ref_shim/ (LibVisualSLAM's triangulation helpers are OUR definitions, ref_shim/ref_triangulate_impl.cpp: the vectors
// pin the loops -- which points are touched, which views are taken and in which order).  A scene = nCams cameras with H frames of
// poses each (CamPoseItem), nPts map points on CoSLAM::curMapPts / actMapPts, per point and camera possibly a feature point of the
// current frame with a track of L <= H frames behind it (FeaturePoint::preFrame).  Scenes: moving rigs, a rig that stands still for
// part of the history (equal angles: the first in the backward walk wins), cameras that only rotate (angle 0: no second view).
//   ref_update_points_test golden <out.bin>
//   ref_update_points_test golden_relink <out.bin>
// golden_relink: chains as the registration loops leave them (src/app/SL_CoSLAM.cpp:775-779, :997-1000: `pFeat->preFrame =
// p->pFeatures[iCam]` hangs the point's OLD chain behind the feature of a new track, whose own earlier frames drop out) and as a lost
// track leaves them (nothing clears MapPoint::pFeatures[c]: the feature of an OLDER frame stays, and every loop here takes it as a
// view with the pose of its own frame).  Per point and camera: 0-3 segments of consecutive frames, newest first, gaps in between, the
// newest at the current frame or a few frames back (stale).  Same file layout except per point and camera: int32 nSeg, featDynamic,
// per segment int32 j0 (history entry of its newest node), L, L x m[2].  Longer histories (40 / 70 frames), 4 scenes.
// Layout of out.bin: int32 nScenes; per scene: int32 nCams, H, nPts, firstKeyFrame, curFrame; double sigma; per camera K[9], iK[9];
// per camera and history entry (newest first) R[9], t[3]; per point: M[3], cov[9], int32 localType, uncertain, lastFrame, isCurrent,
// per camera int32 L, int32 featDynamic, L x m[2] (newest first); then per point the reference's M[3], cov[9]; then per point int32
// refined, and CoSLAM::refineMapPoint's M[3], cov[9] for a copy of the point as it stood before (refined = 0: not called, unchanged);
// then int32 nPairs and per pair of temporary points handed to CoSLAM::checkUnify: int32 p1, p2, per camera int32 has1, has2, M1[3],
// M2[3], int32 ok, M[3], cov[9].
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "app/SL_CoSLAM.h"
#include "app/SL_CoSLAMRobustBA.h"
#include "app/SL_GlobParam.h"

void getInvK(const double* K, double* iK);

static unsigned long long g_rng = 0x2545F4914F6CDD1Dull;
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
    int nStale = 0, nLinked = 0, nChains = 0;
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    const int nScenes = relink ? 4 : 6;
    puti(f, nScenes);
    int nTouched = 0, nTotal = 0, nMoved2 = 0, nRefined = 0, nUnify = 0, nUnifyAll = 0;
    for (int sc = 0; sc < nScenes; ++sc) {
        const int nCams = 2 + sc % 4, H = relink ? 40 + 30 * (sc % 2) : 6 + 5 * (sc % 3), nPts = 60, curFrame = 200 + sc, firstKey = curFrame - 12;
        // kind of motion: 0 moving rig, 1 stands still for the older half of the history, 2 rotation only
        const int motion = sc == 3 ? 1 : (sc == 4 ? 2 : 0);
        CoSLAM* co = new CoSLAM();
        co->numCams = nCams;
        const double sigma = Const::PIXEL_ERR_VAR;
        puti(f, nCams), puti(f, H), puti(f, nPts), puti(f, firstKey), puti(f, curFrame);
        put(f, &sigma, 1);
        std::vector<std::vector<double> > Ks(nCams, std::vector<double>(9));
        for (int c = 0; c < nCams; ++c) {
            const double K[9] = {515 + 15 * urand(), (c % 2) ? 0.3 : 0.0, 320 + 4 * nrand(), 0, 512 + 15 * urand(), 240 + 4 * nrand(), 0, 0, 1};
            double iK[9];
            getInvK(K, iK);
            memcpy(Ks[c].data(), K, 72);
            put(f, K, 9), put(f, iK, 9);
        }
        std::vector<std::vector<CamPoseItem*> > cams(nCams, std::vector<CamPoseItem*>(H));
        for (int c = 0; c < nCams; ++c)
            for (int j = 0; j < H; ++j) {   // j = 0: the newest frame
                const int jj = (motion == 1 && j >= H / 2) ? H / 2 : j;   // the older half repeats one pose
                double w[3] = {0.01 * c, 0.12 * c - 0.006 * jj, 0.002 * jj}, R[9];
                if (motion != 1) w[0] += 0.002 * nrand(), w[2] += 0.002 * nrand();
                rodrigues(w, R);
                double pos[3] = {1.5 * c - 0.09 * jj, 0.05 * c + 0.01 * jj, -0.03 * jj};
                if (motion == 2) pos[0] = 1.5 * c, pos[1] = 0.05 * c, pos[2] = 0;
                if (motion == 0) pos[0] += 0.004 * nrand(), pos[1] += 0.004 * nrand(), pos[2] += 0.004 * nrand();
                double t[3];
                for (int r = 0; r < 3; ++r) t[r] = -(R[3 * r] * pos[0] + R[3 * r + 1] * pos[1] + R[3 * r + 2] * pos[2]);
                if (motion == 1 && j > H / 2) memcpy(R, cams[c][H / 2]->R, 72), memcpy(t, cams[c][H / 2]->t, 24);   // bit-identical poses
                cams[c][j] = new CamPoseItem();
                cams[c][j]->f = curFrame - j, cams[c][j]->camId = c;
                memcpy(cams[c][j]->R, R, 72), memcpy(cams[c][j]->t, t, 24);
                put(f, R, 9), put(f, t, 3);
            }
        std::vector<MapPoint*> pts(nPts);
        std::vector<FeaturePoint*> allFp;
        for (int p = 0; p < nPts; ++p) {
            const double X[3] = {-2 + 7 * urand(), -1.5 + 3 * urand(), 7 + 6 * urand()};
            MapPoint* mp = new MapPoint(X[0] + 0.05 * nrand(), X[1] + 0.05 * nrand(), X[2] + 0.1 * nrand(), curFrame - 30);
            double A[9];
            for (int q = 0; q < 9; ++q) A[q] = 0.05 * nrand();
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) mp->cov[3 * r + c] = A[3 * r] * A[3 * c] + A[3 * r + 1] * A[3 * c + 1] + A[3 * r + 2] * A[3 * c + 2] + (r == c ? 1e-6 : 0);
            const double u = urand();
            if (u < 0.22) mp->setLocalDynamic();
            else if (u < 0.27) mp->setFalse();
            else mp->setLocalStatic();
            if (urand() < 0.1) mp->setUncertain();
            mp->lastFrame = (urand() < 0.12) ? firstKey - (int)(3 * urand()) : curFrame - (int)(4 * urand());   // some at / before the first key frame
            const int isCur = urand() < 0.75;
            put(f, mp->M, 3), put(f, mp->cov, 9);
            puti(f, mp->iLocalType), puti(f, mp->bUncertain ? 1 : 0), puti(f, mp->lastFrame), puti(f, isCur);
            for (int c = 0; c < nCams && relink; ++c) {
                const double u2 = urand();
                const int want = u2 < 0.2 ? 0 : (u2 < 0.4 ? 1 : (u2 < 0.8 ? 2 : 3));
                const int dyn = urand() < 0.35;
                int j = (want > 0 && urand() < 0.3) ? 1 + (int)(urand() * 6) : 0;   // a stale head: the camera lost the point j frames ago
                std::vector<std::pair<int, int> > segs;   // (newest entry, nodes)
                for (int q = 0; q < want && j < H; ++q) {
                    int L = 1 + (int)(urand() * (q == 0 && want > 1 ? 10 : 25));   // (a freshly re-linked head is 1 node long)
                    if (q == 0 && want > 1 && urand() < 0.3) L = 1;
                    if (j + L > H) L = H - j;
                    segs.push_back(std::make_pair(j, L));
                    j += L + 1 + (int)(urand() * 12);   // the frames in which the camera did not see the point
                }
                puti(f, (int)segs.size()), puti(f, dyn);
                if (!segs.empty()) ++nChains, nStale += segs[0].first > 0, nLinked += segs.size() > 1;
                FeaturePoint* newer = nullptr;
                for (size_t q = 0; q < segs.size(); ++q) {
                    puti(f, segs[q].first), puti(f, segs[q].second);
                    for (int jj = segs[q].first; jj < segs[q].first + segs[q].second; ++jj) {
                        const double* R = cams[c][jj]->R;
                        const double* t = cams[c][jj]->t;
                        const double* K = Ks[c].data();
                        double Xc[3], m[2];
                        for (int r = 0; r < 3; ++r) Xc[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r];
                        m[0] = (K[0] * Xc[0] + K[1] * Xc[1] + K[2] * Xc[2]) / Xc[2] + 0.6 * nrand();
                        m[1] = (K[4] * Xc[1] + K[5] * Xc[2]) / Xc[2] + 0.6 * nrand();
                        put(f, m, 2);
                        FeaturePoint* fp = new FeaturePoint(curFrame - jj, c, m[0], m[1]);
                        fp->setIntrinsic(K);
                        fp->setCameraPose(cams[c][jj]);
                        fp->type = dyn ? TYPE_FEATPOINT_DYNAMIC : TYPE_FEATPOINT_STATIC;
                        fp->preFrame = nullptr;
                        if (newer) newer->preFrame = fp, fp->nextFrame = newer;   // (across a gap: what :777-778 assigns)
                        else mp->pFeatures[c] = fp;
                        newer = fp;
                        allFp.push_back(fp);
                    }
                }
            }
            for (int c = 0; c < nCams && !relink; ++c) {
                const int L = (urand() < 0.3) ? 0 : 1 + (int)(urand() * H);
                const int dyn = urand() < 0.35;
                puti(f, L), puti(f, dyn);
                FeaturePoint* newer = nullptr;
                for (int j = 0; j < L; ++j) {
                    const double* R = cams[c][j]->R;
                    const double* t = cams[c][j]->t;
                    const double* K = Ks[c].data();
                    double Xc[3], m[2];
                    for (int r = 0; r < 3; ++r) Xc[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r];
                    m[0] = (K[0] * Xc[0] + K[1] * Xc[1] + K[2] * Xc[2]) / Xc[2] + 0.6 * nrand();
                    m[1] = (K[4] * Xc[1] + K[5] * Xc[2]) / Xc[2] + 0.6 * nrand();
                    put(f, m, 2);
                    FeaturePoint* fp = new FeaturePoint(curFrame - j, c, m[0], m[1]);
                    fp->setIntrinsic(K);
                    fp->setCameraPose(cams[c][j]);
                    fp->type = dyn ? TYPE_FEATPOINT_DYNAMIC : TYPE_FEATPOINT_STATIC;
                    fp->preFrame = nullptr;
                    if (newer) newer->preFrame = fp, fp->nextFrame = newer;
                    else mp->pFeatures[c] = fp;
                    newer = fp;
                    allFp.push_back(fp);
                }
            }
            pts[p] = mp;
            if (isCur) co->curMapPts.add(mp); else co->actMapPts.add(mp);
        }
        // CoSLAM::refineMapPoint (src/app/SL_CoSLAM.cpp:666-713: what the registration loops call on a point that just gained a feature,
        // :896, :948, :1166) on a COPY of every point that at least two cameras see -- the same views and the same two helper calls as
        // updateStaticPointPosition, whatever the point's type, and no frame test
        std::vector<int> refSel(nPts, 0);
        std::vector<double> refM(3 * nPts), refCov(9 * nPts);
        for (int c = 0; c < nCams; ++c) {
            co->slam[c].K.resize(3, 3);
            memcpy(co->slam[c].K.data, Ks[c].data(), 72);
        }
        for (int p = 0; p < nPts; ++p) {
            int seen = 0;
            for (int c = 0; c < nCams; ++c) seen += pts[p]->pFeatures[c] != nullptr;
            MapPoint tmp(*pts[p]);
            if (seen >= 2) {
                co->refineMapPoint(&tmp);
                refSel[p] = 1;
                ++nRefined;
            }
            memcpy(&refM[3 * p], tmp.M, 24), memcpy(&refCov[9 * p], tmp.cov, 72);
        }
        // CoSLAM::checkUnify (src/app/SL_CoSLAM.cpp:561-665) on pairs of temporary points: the two halves of one point's cameras (the same
        // physical point: they should unify) and two different points (they should not); features of this frame, whole tracks
        struct UnifyRec {
            int p1, p2, ok;
            int has1[SLAM_MAX_NUM], has2[SLAM_MAX_NUM];
            double M1[3], M2[3], M[3], cov[9];
        };
        std::vector<UnifyRec> uni;
        for (int p = 0; p + 1 < nPts; ++p) {
            int seen = 0;
            for (int c = 0; c < nCams; ++c) seen += pts[p]->pFeatures[c] != nullptr;
            for (int mode = 0; mode < 2; ++mode) {
                if (mode == 0 && seen < 2) continue;
                MapPoint A(*pts[p]), B(mode == 0 ? *pts[p] : *pts[p + 1]);
                UnifyRec r;
                memset(&r, 0, sizeof(r));
                r.p1 = p, r.p2 = mode == 0 ? p : p + 1;
                if (mode == 0) {   // the cameras that see the point dealt out alternately
                    int k = 0;
                    for (int c = 0; c < nCams; ++c)
                        if (pts[p]->pFeatures[c]) {
                            if (k++ % 2) A.pFeatures[c] = nullptr; else B.pFeatures[c] = nullptr;
                        }
                    B.M[0] += 0.02, B.M[1] -= 0.015, B.M[2] += 0.03;
                }
                int nA = 0, nB = 0;
                for (int c = 0; c < nCams; ++c) {
                    r.has1[c] = A.pFeatures[c] != nullptr, r.has2[c] = B.pFeatures[c] != nullptr;
                    nA += r.has1[c], nB += r.has2[c];
                }
                if (nA + nB < 2 || nA == 0 || nB == 0) continue;
                memcpy(r.M1, A.M, 24), memcpy(r.M2, B.M, 24);
                r.ok = co->checkUnify(&A, &B, r.M, r.cov, Const::PIXEL_ERR_VAR) ? 1 : 0;
                bool fin = true;
                for (int q = 0; q < 3; ++q) fin = fin && fabs(r.M[q]) < 1e6;
                if (!fin) continue;   // (a degenerate union -- rotation-only rig -- is not a test vector)
                uni.push_back(r);
                nUnify += r.ok, ++nUnifyAll;
            }
        }
        std::vector<double> before(3 * nPts);
        for (int p = 0; p < nPts; ++p) memcpy(&before[3 * p], pts[p]->M, 24);
        RobustBundleRTS ba;
        ba.setCoSLAM(co);
        KeyFrame kf(firstKey);
        ba.firstKeyFrame = &kf;
        ba.updateNewPosesPoints();
        for (int p = 0; p < nPts; ++p) {
            put(f, pts[p]->M, 3), put(f, pts[p]->cov, 9);
            ++nTotal;
            if (memcmp(&before[3 * p], pts[p]->M, 24)) ++nTouched;
            for (int q = 0; q < 3; ++q)
                if (!(fabs(pts[p]->M[q]) < 1e3)) ++nMoved2;
        }
        for (int p = 0; p < nPts; ++p) puti(f, refSel[p]), put(f, &refM[3 * p], 3), put(f, &refCov[9 * p], 9);
        puti(f, (int)uni.size());
        for (const UnifyRec& r : uni) {
            puti(f, r.p1), puti(f, r.p2);
            for (int c = 0; c < nCams; ++c) puti(f, r.has1[c]), puti(f, r.has2[c]);
            put(f, r.M1, 3), put(f, r.M2, 3);
            puti(f, r.ok), put(f, r.M, 3), put(f, r.cov, 9);
        }
        co->curMapPts.clearWithoutRelease(), co->actMapPts.clearWithoutRelease();
    }
    fclose(f);
    printf("ref_update_points_test: %d scenes, %d of %d points re-triangulated, %d wild coordinates; refineMapPoint on %d points; checkUnify: %d of %d pairs unify\n",
           nScenes, nTouched, nTotal, nMoved2, nRefined, nUnify, nUnifyAll);
    if (relink) {
        printf("  chains: %d, with a stale head %d, with an older segment linked behind %d\n", nChains, nStale, nLinked);
        return (nTouched > nTotal / 5 && nMoved2 == 0 && nRefined > nTotal / 5 && nStale > 40 && nLinked > 100 && nUnify > 10) ? 0 : 1;
    }
    return (nTouched > nTotal / 4 && nTouched < nTotal && nMoved2 == 0 && nRefined > nTotal / 4 && nUnify > 20 && nUnifyAll - nUnify > 20) ? 0 : 1;
}
