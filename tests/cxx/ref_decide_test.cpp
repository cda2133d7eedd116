// ref_decide_test.cpp -- the reference's OWN CoSLAM::curStaticPointsRegInGroup (src/app/SL_CoSLAM.cpp:854-898) with
// curStaticPointRegInGroup (:731-830, bMerge == false), staticCheckMergability (:714-729), refineMapPoint (:666-713) and
// SingleSLAM's searchMahaNearestFeatPt (src/app/SL_SingleSLAM.cpp:1141-1164) on map points, cameras and feature tracks built with the
// reference's classes; writes the scenes as structure-of-arrays records and what the reference made of them -- which feature
// carries which map point afterwards, every point's position and covariance -- for tests/golden/make_golden.py (CPU only).
//
// oracle/Makefile compiles SL_CoSLAM.cpp, SL_CoSLAMHelper.cpp and SL_SingleSLAM.cpp IN PLACE against oracle/ref_shim/ (LibVisualSLAM's
// helpers are OUR definitions: the vectors pin the reference's LOOPS -- who is visited in which order, what ends a walk, who gets
// refined when).  compareFeaturePt cuts two NCC blocks and returns true whatever their score (:546-558): the block functions are
// stubbed here.  A scene: nCams cameras with a short pose history, nPts map points on curMapPts (certain static ones, plus
// uncertain / dynamic / false ones that must not be visited), per point and camera one of: a feature of this frame that already
// carries the point; an unmapped feature at its projection with a consistent track (-> attached); one whose track is inconsistent
// further back (-> not mergeable); a DYNAMIC one (-> passed by); one that already carries ANOTHER point (-> the walk ends there);
// nothing.  Twin points a few millimetres apart compete for the same features; distractor features fill the frames.
//   ref_decide_test golden <out.bin>
// out.bin (int32 / float64): nScenes; per scene: nCams Hh N nPts curFrame W H withDynamic withMerge; pixelVar; per camera K[9], then per history entry
// (newest first) R[9] t[3]; per camera and slot: L (0: empty) isStatic slot2map, L x m[2] (newest first); per point M[3] cov[9] flags
// (1 dynamic, 2 false, 4 uncertain) pointFeat[nCams]; then the reference's result: nRegged nReggedDynamic; per camera slot2map[N]; per point
// M[3] cov[9]; per point flags pointFeat[nCams] afterwards.   TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "app/SL_CoSLAM.h"
#include "app/SL_GlobParam.h"

// compareFeaturePt's block functions: off the path of what is pinned here (it returns true whatever they compute)
void getNCCBlock(const ImgG&, double, double, NCCBlock&) {}
double matchNCCBlock(NCCBlock*, NCCBlock*) { return 1.0; }
bool NCCBlock::computeScaled(const ImgG&, double, double, double) { return true; }
// un-vendored; only picks the camera whose feature compareFeaturePt looks at (no influence on what is pinned): distance of the centres
double getCamDist(const CamPoseItem* a, const CamPoseItem* b) {
    double d2 = 0;
    for (int i = 0; i < 3; ++i) {
        const double ca = -(a->R[i] * a->t[0] + a->R[3 + i] * a->t[1] + a->R[6 + i] * a->t[2]);
        const double cb = -(b->R[i] * b->t[0] + b->R[3 + i] * b->t[1] + b->R[6 + i] * b->t[2]);
        d2 += (ca - cb) * (ca - cb);
    }
    return sqrt(d2);
}

static unsigned long long g_rng = 0xC2B2AE3D27D4EB4Full;
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

struct Slot {
    int L = 0, isStatic = 1, s2m = -1;
    std::vector<double> m;   // L x 2, newest first
    FeaturePoint* tail = nullptr;
};

int main(int argc, char** argv) {
    if (argc < 3 || (strcmp(argv[1], "golden") && strcmp(argv[1], "golden_relink"))) {
        fprintf(stderr, "usage: %s golden|golden_relink <out.bin>\n", argv[0]);
        return 2;
    }
    // golden_relink: MapPoint::pFeatures as the registration loops and lost tracks leave them over time -- a camera that lost a point keeps its
    // last feature (a STALE head: every walk of checkUnify / refineMapPoint takes it as a view, curStaticPointRegInGroup looks for a new
    // feature there and links the old chain behind it, :775-779; at a unification it blocks the hand-over in its camera, :806-816, and the
    // other point's stale features move too), and a feature's preFrame chain may jump into an older track.  Scenes 5, 6 (bMerge) and 0, 3.
    // Added to the file per point and camera (behind the point's record): int32 staleHead (1: the point's feature there is a stale one),
    // isStatic of that feature, nExtra segments of consecutive frames on dead tracks -- the stale head's run first, then what is linked behind
    // -- each int32 j0 (history entry of its newest node), L, L x m[2]; behind the reference's result per point and camera: int32
    // staleOwner (the point that owned the stale feature now held there, -1 none), preOwner (the owner of the dead-track chain linked
    // directly behind the live feature now held there, -1 none).
    const bool relink = !strcmp(argv[1], "golden_relink");
    if (relink) g_rng = 0xA0761D6478BD642Full;
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    const int nScenes = relink ? 4 : 7;   // 0-2: the static points' registration alone; 3, 4: more certainly dynamic points with DYNAMIC candidates, and
                             // curDynamicPointsRegInGroup behind it (CoSLAM::currentMapPointsRegister's order, :834-853); 5, 6: the same with
                             // bMerge == true (every 50th frame, CoSLAMThread.cpp:117-118): a walk that meets a feature of another static
                             // point asks checkUnify and, on a yes, takes that point's features (:791-826) -- more twins here
    puti(f, nScenes);
    int tot[6] = {0, 0, 0, 0, 0, 0};   // attached, not mergeable, dynamic passed by, walks ended by a mapped feature, twins, regged
    for (int scq = 0; scq < nScenes; ++scq) {
        static const int relinkScenes[4] = {5, 6, 0, 3};
        const int sc = relink ? relinkScenes[scq] : scq;
        const bool dyn = sc >= 3, merge = sc >= 5;
        const int nCams = merge ? sc - 2 : (dyn ? sc : 3 + sc), Hh = 20, nBase = 110 + 20 * sc, curFrame = 200 + 11 * sc, W = 640, H = 480;
        const double pixelVar = 10.0;   // Const::PIXEL_ERR_VAR as CoSLAMThread.cpp:117 passes it
        CoSLAM* co = new CoSLAM();
        co->numCams = nCams;
        co->curFrame = curFrame;
        const double K[9] = {520, 0, 320, 0, 520, 240, 0, 0, 1};
        const double iK[9] = {1 / 520.0, 0, -320 / 520.0, 0, 1 / 520.0, -240 / 520.0, 0, 0, 1};
        const double kud[7] = {0, 0, 0, 0, 0, 0, 0};
        std::vector<std::vector<CamPoseItem*> > cams(nCams, std::vector<CamPoseItem*>(Hh));
        std::vector<std::vector<Slot> > slots(nCams);
        CameraGroup group;
        for (int c = 0; c < nCams; ++c) {
            SingleSLAM* s = &co->slam[c];
            s->camId = c, s->W = W, s->H = H;
            s->K.cloneFrom(K, 3, 3), s->iK.cloneFrom(iK, 3, 3), s->k_ud.cloneFrom(kud, 7, 1);
            s->videoReader = new VideoReader();
            s->videoReader->_w = W, s->videoReader->_h = H;
            s->m_tracker.m_frame = curFrame;
            group.addCam(c);
            for (int j = Hh - 1; j >= 0; --j) {   // oldest first into the pose list; j = 0: the current frame
                double w[3] = {0.01 * c + 0.001 * nrand(), 0.08 * (c - 0.5 * (nCams - 1)) - 0.0015 * j, 0.0005 * j}, R[9];
                rodrigues(w, R);
                const double pos[3] = {1.3 * (c - 0.5 * (nCams - 1)) - 0.035 * j + 0.002 * nrand(), 0.05 * c + 0.004 * j, -0.01 * j};
                double t[3];
                for (int r = 0; r < 3; ++r) t[r] = -(R[3 * r] * pos[0] + R[3 * r + 1] * pos[1] + R[3 * r + 2] * pos[2]);
                cams[c][j] = s->m_camPos.add(curFrame - j, c, R, t);
            }
        }
        auto project = [&](int c, int j, const double* X, double* m) {
            const double* R = cams[c][j]->R;
            const double* t = cams[c][j]->t;
            double Xc[3];
            for (int r = 0; r < 3; ++r) Xc[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r];
            m[0] = K[0] * Xc[0] / Xc[2] + K[2], m[1] = K[4] * Xc[1] / Xc[2] + K[5];
            return Xc[2] > 0.5 && m[0] > 8 && m[0] < W - 8 && m[1] > 8 && m[1] < H - 8;
        };
        // a feature of this frame with its track behind it: `off` pixels added from `offFrom` frames back on (an inconsistent track)
        auto add_track = [&](int c, const double* X, int L, bool isStatic, MapPoint* owner, int ownerIdx, double off, int offFrom) {
            Slot sl;
            sl.L = L, sl.isStatic = isStatic ? 1 : 0, sl.s2m = ownerIdx;
            FeaturePoint* newer = nullptr;
            for (int q = 0; q < L; ++q) {
                double m[2];
                project(c, q, X, m);
                m[0] += 0.35 * nrand(), m[1] += 0.35 * nrand();
                if (q >= offFrom) m[0] += off, m[1] -= 0.7 * off;
                sl.m.push_back(m[0]), sl.m.push_back(m[1]);
                FeaturePoint* fp = new FeaturePoint(curFrame - q, c, m[0], m[1]);
                fp->setIntrinsic(co->slam[c].K.data);
                fp->setCameraPose(cams[c][q]);
                fp->type = isStatic ? TYPE_FEATPOINT_STATIC : TYPE_FEATPOINT_DYNAMIC;
                fp->mpt = owner;
                if (newer)
                    newer->preFrame = fp, fp->nextFrame = newer;
                else
                    sl.tail = fp;
                newer = fp;
            }
            co->slam[c].m_featPts.add(sl.tail);   // the frame's list in slot order: searchMahaNearestFeatPt walks it
            slots[c].push_back(sl);
            return (int)slots[c].size() - 1;
        };
        struct Pt {
            MapPoint* mp;
            double X[3];
        };
        std::vector<Pt> pts;
        // chains on dead tracks (relink): per (point, camera) the segments and whether the first of them is the point's (stale) feature there
        struct DeadSeg {
            int j0, L;
            std::vector<double> m;
            FeaturePoint *newest, *oldest;
        };
        std::map<std::pair<int, int>, std::vector<DeadSeg> > dead;
        std::map<std::pair<int, int>, int> staleHead, staleStatic;
        std::map<const FeaturePoint*, int> deadOwner;   // every node of a dead-track chain -> the point that owned it when the scene was built
        auto add_dead = [&](int c, const double* X, int j0, int L, bool isStatic, MapPoint* owner, int ownerIdx) {
            DeadSeg sg;
            sg.j0 = j0, sg.L = L, sg.newest = sg.oldest = nullptr;
            FeaturePoint* newer = nullptr;
            for (int q = 0; q < L; ++q) {
                double m[2];
                project(c, j0 + q, X, m);
                m[0] += 0.35 * nrand(), m[1] += 0.35 * nrand();
                sg.m.push_back(m[0]), sg.m.push_back(m[1]);
                FeaturePoint* fp = new FeaturePoint(curFrame - (j0 + q), c, m[0], m[1]);
                fp->setIntrinsic(co->slam[c].K.data);
                fp->setCameraPose(cams[c][j0 + q]);
                fp->type = isStatic ? TYPE_FEATPOINT_STATIC : TYPE_FEATPOINT_DYNAMIC;
                fp->mpt = owner;
                deadOwner[fp] = ownerIdx;
                if (newer) newer->preFrame = fp, fp->nextFrame = newer;
                else sg.newest = fp;
                newer = fp;
            }
            sg.oldest = newer;
            return sg;
        };
        // the point's pFeatures in camera c gets (more of) a chain on dead tracks: a stale head when it holds no feature there, else segments
        // linked behind the live feature's own track (whose oldest node is `liveOldest`, `liveL` frames long)
        auto give_chain = [&](int p, int c, FeaturePoint* liveOldest, int liveL, bool isStatic) {
            const std::pair<int, int> key(p, c);
            int j = liveOldest ? liveL + (int)(urand() * 3) : 1 + (int)(urand() * 5);
            const int want = 1 + (urand() < 0.35 ? 1 : 0);
            FeaturePoint* link = liveOldest;
            for (int q = 0; q < want && j + 1 < Hh; ++q) {
                int L = 1 + (int)(urand() * 5);
                if (j + L > Hh) L = Hh - j;
                DeadSeg sg = add_dead(c, pts[p].X, j, L, isStatic, pts[p].mp, p);
                if (link) link->preFrame = sg.newest, sg.newest->nextFrame = link;
                else pts[p].mp->pFeatures[c] = sg.newest, staleHead[key] = 1, staleStatic[key] = isStatic ? 1 : 0;
                link = sg.oldest;
                dead[key].push_back(sg);
                j += L + 1 + (int)(urand() * 3);
            }
        };
        auto new_point = [&](const double* X, int kind) {
            Pt P;
            memcpy(P.X, X, 24);
            P.mp = new MapPoint(X[0] + 0.02 * nrand(), X[1] + 0.02 * nrand(), X[2] + 0.04 * nrand(), curFrame - 40);
            double A[9];
            for (int q = 0; q < 9; ++q) A[q] = 0.03 * nrand();
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) P.mp->cov[3 * r + c] = A[3 * r] * A[3 * c] + A[3 * r + 1] * A[3 * c + 1] + A[3 * r + 2] * A[3 * c + 2] + (r == c ? 1e-5 : 0);
            if (kind == 8) P.mp->setLocalDynamic(); else P.mp->setLocalStatic();
            if (kind == 7) P.mp->setUncertain();
            if (kind == 9) P.mp->setFalse();
            P.mp->bNewPt = false;
            P.mp->lastFrame = curFrame;
            pts.push_back(P);
            return (int)pts.size() - 1;
        };
        for (int b = 0; b < nBase; ++b) {
            const double X[3] = {-2.2 + 4.4 * urand(), -1.4 + 2.8 * urand(), 7 + 5 * urand()};
            int kind = b % 10;
            if (dyn && (kind == 3 || kind == 5)) kind = 8;   // (three in ten certainly dynamic)
            const bool dk = dyn && kind == 8;               // a dynamic point of a scene that registers them: its features are DYNAMIC ones
            const int p = new_point(X, kind);
            const bool twin = (kind < 7 || dk) && urand() < (merge ? 0.45 : 0.12);
            int p2 = -1;
            if (twin) {
                const double X2[3] = {X[0] + 0.004, X[1] - 0.003, X[2] + 0.005};
                p2 = new_point(X2, dk ? 8 : 0);
                ++tot[4];
            }
            int nHas = 0;
            for (int c = 0; c < nCams; ++c) {
                double m[2];
                if (!project(c, 0, X, m)) continue;
                double r = urand();
                if (c == nCams - 1 && nHas == 0) r = 0;   // every point is in the current list through at least one camera
                const int L = 2 + (int)(urand() * (Hh - 2));
                if (r < 0.42) {
                    const int Lh = relink && urand() < 0.3 ? 1 + (int)(urand() * 4) : L;   // (relink: a short live track with the OLD chain linked behind it)
                    const int s = add_track(c, X, relink ? Lh : L, !dk, pts[p].mp, p, 0, 0);
                    pts[p].mp->pFeatures[c] = slots[c][s].tail;
                    ++nHas;
                    if (relink && Lh != L) {
                        FeaturePoint* o = slots[c][s].tail;
                        while (o->preFrame) o = o->preFrame;
                        give_chain(p, c, o, Lh, !dk);
                    }
                    if (twin && urand() < 0.5) {   // the twin holds a feature of its own in this camera
                        const int s2 = add_track(c, pts[p2].X, L, !dk, pts[p2].mp, p2, 0, 0);
                        pts[p2].mp->pFeatures[c] = slots[c][s2].tail;
                    }
                } else if (r < 0.72) {
                    const double ut = urand();   // (in this order: what the two draws in one argument list compiled to before)
                    const int Lc = urand() < 0.15 ? 1 : L;
                    add_track(c, X, Lc, dk ? ut < 0.2 : ut < 0.9, nullptr, -1, 0, 0);   // a candidate (10 % DYNAMIC; 80 % for a dynamic point)
                } else if (r < 0.80) {
                    add_track(c, X, L < 4 ? 4 : L, !dk, nullptr, -1, 22.0, 2);                     // inconsistent from two frames back on
                } else if (r < 0.89 && p > 4) {
                    // a feature at the projection that already carries ANOTHER point (one without a feature in this camera yet)
                    for (int tries = 0; tries < 12; ++tries) {
                        const int q = (int)(urand() * p);
                        if (q == p2 || pts[q].mp->pFeatures[c]) continue;
                        const int s = add_track(c, X, L, !dk, pts[q].mp, q, 0, 0);
                        pts[q].mp->pFeatures[c] = slots[c][s].tail;
                        break;
                    }
                }
            }
            if (relink)   // cameras in which the point holds no feature of this frame: some keep a stale one
                for (int c = 0; c < nCams; ++c) {
                    double m[2];
                    if (!pts[p].mp->pFeatures[c] && project(c, 0, X, m) && urand() < 0.4) give_chain(p, c, nullptr, 0, !dk);
                    if (twin && !pts[p2].mp->pFeatures[c] && project(c, 0, pts[p2].X, m) && urand() < 0.3) give_chain(p2, c, nullptr, 0, !dk);
                }
            if (twin) {   // make sure the twin is in the current list too
                bool any = false;
                for (int c = 0; c < nCams; ++c) any |= pts[p2].mp->pFeatures[c] != nullptr;
                if (!any) {
                    const int c = 0;
                    double m[2];
                    if (project(c, 0, pts[p2].X, m)) {
                        const int s2 = add_track(c, pts[p2].X, 6, !dk, pts[p2].mp, p2, 0, 0);
                        pts[p2].mp->pFeatures[c] = slots[c][s2].tail;
                    }
                }
            }
        }
        for (int c = 0; c < nCams; ++c)   // distractors
            for (int d = 0; d < 70; ++d) {
                const double X[3] = {-2.5 + 5 * urand(), -1.6 + 3.2 * urand(), 6 + 7 * urand()};
                double m[2];
                if (project(c, 0, X, m)) add_track(c, X, 1 + (int)(urand() * 8), urand() < 0.85, nullptr, -1, 0, 0);
            }
        const int nPts = (int)pts.size();
        int N = 0;
        for (int c = 0; c < nCams; ++c) N = std::max(N, (int)slots[c].size());
        for (int p = 0; p < nPts; ++p) {
            pts[p].mp->updateVisCamNum(curFrame);
            if (pts[p].mp->numVisCam > 0) co->curMapPts.add(pts[p].mp);
        }
        // ---- inputs
        puti(f, nCams), puti(f, Hh), puti(f, N), puti(f, nPts), puti(f, curFrame), puti(f, W), puti(f, H), puti(f, dyn ? 1 : 0), puti(f, merge ? 1 : 0);
        put(f, &pixelVar, 1);
        for (int c = 0; c < nCams; ++c) {
            put(f, K, 9);
            for (int j = 0; j < Hh; ++j) put(f, cams[c][j]->R, 9), put(f, cams[c][j]->t, 3);
        }
        std::map<const MapPoint*, int> idx;
        for (int p = 0; p < nPts; ++p) idx[pts[p].mp] = p;
        for (int c = 0; c < nCams; ++c)
            for (int s = 0; s < N; ++s) {
                if (s >= (int)slots[c].size()) {
                    puti(f, 0), puti(f, 1), puti(f, -1);
                    continue;
                }
                const Slot& sl = slots[c][s];
                puti(f, sl.L), puti(f, sl.isStatic), puti(f, sl.s2m);
                put(f, sl.m.data(), sl.m.size());
            }
        for (int p = 0; p < nPts; ++p) {
            MapPoint* mp = pts[p].mp;
            put(f, mp->M, 3), put(f, mp->cov, 9);
            puti(f, (mp->isLocalDynamic() ? 1 : 0) | (mp->isFalse() ? 2 : 0) | (mp->isUncertain() ? 4 : 0));
            for (int c = 0; c < nCams; ++c) {
                int s = -1;
                for (int q = 0; q < (int)slots[c].size(); ++q)
                    if (slots[c][q].tail == mp->pFeatures[c]) s = q;
                puti(f, mp->pFeatures[c] ? s : -1);
            }
            if (relink)
                for (int c = 0; c < nCams; ++c) {
                    const std::pair<int, int> key(p, c);
                    puti(f, staleHead.count(key) ? 1 : 0), puti(f, staleStatic.count(key) ? staleStatic[key] : 1);
                    const std::vector<DeadSeg>& v = dead[key];
                    puti(f, (int)v.size());
                    for (size_t q = 0; q < v.size(); ++q) puti(f, v[q].j0), puti(f, v[q].L), put(f, v[q].m.data(), v[q].m.size());
                }
        }
        // ---- the reference
        const int nRegged = co->curStaticPointsRegInGroup(group, pixelVar, merge);
        const int nReggedDyn = dyn ? co->curDynamicPointsRegInGroup(group, pixelVar, merge) : 0;
        tot[5] += nRegged + nReggedDyn;
        puti(f, nRegged), puti(f, nReggedDyn);
        for (int c = 0; c < nCams; ++c)
            for (int s = 0; s < N; ++s) {
                int m = -1;
                if (s < (int)slots[c].size() && slots[c][s].tail->mpt) m = idx[slots[c][s].tail->mpt];
                puti(f, m);
                if (s < (int)slots[c].size() && slots[c][s].s2m < 0 && m >= 0) ++tot[0];
            }
        for (int p = 0; p < nPts; ++p) put(f, pts[p].mp->M, 3), put(f, pts[p].mp->cov, 9);
        int nMerged = 0;
        for (int p = 0; p < nPts; ++p) {   // the points' types afterwards (a point unified away is false) and their features of this frame
            MapPoint* mp = pts[p].mp;
            const int fl = (mp->isLocalDynamic() ? 1 : 0) | (mp->isFalse() ? 2 : 0) | (mp->isUncertain() ? 4 : 0);
            puti(f, fl);
            for (int c = 0; c < nCams; ++c) {
                int sIdx = -1;
                if (mp->pFeatures[c] && mp->pFeatures[c]->f == curFrame)
                    for (int q = 0; q < (int)slots[c].size(); ++q)
                        if (slots[c][q].tail == mp->pFeatures[c]) sIdx = q;
                puti(f, sIdx);
            }
            if (relink)
                for (int c = 0; c < nCams; ++c) {
                    const FeaturePoint* fp = mp->pFeatures[c];
                    int staleOwner = -1, preOwner = -1;
                    if (fp && fp->f != curFrame) staleOwner = deadOwner.count(fp) ? deadOwner[fp] : -2;
                    if (fp && fp->f == curFrame) {   // the first dead-track node behind the live feature's own run of consecutive frames
                        const FeaturePoint* o = fp;
                        while (o->preFrame && !deadOwner.count(o->preFrame)) o = o->preFrame;
                        if (o->preFrame) preOwner = deadOwner[o->preFrame];
                    }
                    puti(f, staleOwner), puti(f, preOwner);
                }
        }
        for (int p = 0; p < nPts; ++p) nMerged += pts[p].mp->isFalse() && (p % 1 == 0) ? 1 : 0;
        tot[3] += nMerged;
        printf("scene %d: %d cameras, %d slots, %d points (%d on the current list): %d static + %d dynamic points registered; %d stale heads, %d dead-track segments\n",
               sc, nCams, N, nPts, co->curMapPts.getNum(), nRegged, nReggedDyn, (int)staleHead.size(), (int)deadOwner.size());
        co->curMapPts.clearWithoutRelease();
    }
    fclose(f);
    printf("ref_decide_test: %d features attached, %d twins, %d points registered, %d points false afterwards\n", tot[0], tot[4], tot[5], tot[3]);
    return tot[0] > 100 ? 0 : 1;
}
