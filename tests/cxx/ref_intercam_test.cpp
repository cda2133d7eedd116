// ref_intercam_test.cpp -- the reference's OWN InterCamPoseEstimator::addMapPoints (src/app/SL_InterCamPoseEstimator.cpp:18-91, through
// its own SingleSLAM::chooseStaticFeatPts / chooseDynamicFeatPts, src/app/SL_SingleSLAM.cpp:345-447) on cameras built with the
// reference's classes; writes the cameras' state as structure-of-arrays records and the flattened vecPts3D / vecMeas2D it produced,
// for tests/golden/make_golden.py (CPU only).
//
// oracle/Makefile compiles SL_InterCamPoseEstimator.cpp and SL_SingleSLAM.cpp IN PLACE (never copied) against oracle/ref_shim/.
// A scene = nc cameras with N tracker slots each; map points of every kind: certain static (seen by one or several cameras),
// certain dynamic, uncertain + new, uncertain + old, false; features of this frame and stale ones (a camera that lost the point keeps
// its last feature: pFeatures[c]->f < curFrame); tracks born in this frame; blocks with several candidates; more dynamic points than
// maxDyn + 1 in the first scene, fewer in the second.
//   ref_intercam_test golden <out.bin>
// Layout of out.bin (int32 / float64): nScenes; per scene: nc N nMap frame W H nColBlk nRowBlk ptsStride; mapPts[nMap][3];
// mapFlags[nMap] (as int32: CS_MAP_* bits); newPt[nMap] (int32); pointFeat[nMap][nc] (int32); per camera: xy[2N] (x then y),
// state[N], slot2map[N], isStatic[N] (int32: FeaturePoint::type == STATIC), trackSpan[2N] (first, last frame); then the reference's result: nStatic nDynamic P nObs; pts[P][3]; obs_ptr[P+1];
// obs_cam[nObs]; obs_xy[nObs][2]; pointMap[P]; per camera Rs[9] Ts[3] of the problem, then R[9] t[3] of m_camPos.current().
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#include "app/SL_CoSLAM.h"
#include "app/SL_InterCamPoseEstimator.h"

static unsigned long long g_rng = 0xD1B54A32D192ED03ull;
static double urand() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return (double)(g_rng >> 11) / 9007199254740992.0;
}

template <class T>
static void put(FILE* f, const std::vector<T>& v) {
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
}
static void puti(FILE* f, int v) { fwrite(&v, 4, 1, f); }

static int scene(FILE* f, int nc, int nMap, int nDynWanted) {
    const int W = 640, H = 480, frame = 40, PTS = 192;
    const double K[9] = {0.82 * W, 0, W / 2.0, 0, 0.82 * W, H / 2.0, 0, 0, 1};
    const double iK[9] = {1 / K[0], 0, -K[2] / K[0], 0, 1 / K[4], -K[5] / K[4], 0, 0, 1};
    const double kud[7] = {0, 0, 0, 0, 0, 0, 0};
    CoSLAM* co = (CoSLAM*)calloc(1, sizeof(CoSLAM));   // (the class's constructor lives with the GUI; only numCams / curFrame / slam[] are touched)
    co->numCams = nc;
    co->curFrame = frame;
    // the map in ONE array: address order (what the reference's std::map<MapPoint*, int> iterates in) = index order
    MapPoint* mpts = (MapPoint*)calloc(nMap, sizeof(MapPoint));
    std::vector<int> flags(nMap), newPt(nMap);
    for (int i = 0; i < nMap; ++i) {
        new (&mpts[i]) MapPoint(-5 + 10 * urand(), -3 + 6 * urand(), 6 + 8 * urand(), 0);
        const double u = urand();
        MapPoint* p = &mpts[i];
        p->bNewPt = urand() < 0.4;
        if (i >= nMap - nDynWanted) {   // the dynamic tail: certain dynamic, or uncertain (new / old)
            if (u < 0.6)
                p->setLocalDynamic();
            else if (u < 0.85)
                p->setLocalDynamic(), p->setUncertain();
            else
                p->setLocalStatic(), p->setUncertain();
        } else if (u < 0.85) {
            p->setLocalStatic();
        } else if (u < 0.92) {
            p->setLocalStatic(), p->setUncertain();
        } else {
            p->setFalse();
        }
        flags[i] = (p->isLocalDynamic() ? 1 : 0) | (p->isFalse() ? 2 : 0) | (p->isUncertain() ? 4 : 0);
        newPt[i] = p->bNewPt ? 1 : 0;
    }
    int N = 0;
    std::vector<std::vector<double> > xy(nc);
    std::vector<std::vector<int> > state(nc), s2m(nc), sel(nc), ftype(nc), span(nc);
    std::vector<int> npts(nc);
    for (int c = 0; c < nc; ++c) {
        SingleSLAM* s = new (&co->slam[c]) SingleSLAM();
        s->camId = c;
        s->W = W, s->H = H;
        s->blkW = W / s->nColBlk, s->blkH = H / s->nRowBlk;   // SL_SingleSLAM.cpp:270-271
        s->K.cloneFrom(K, 3, 3);
        s->iK.cloneFrom(iK, 3, 3);
        s->k_ud.cloneFrom(kud, 7, 1);
        {   // the camera's current pose (addMapPoints copies it, :33-36)
            double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0.1 * c + 0.01 * urand(), 0.02 * urand(), 0.03 * urand()};
            R[1] = 0.01 * (c + 1), R[3] = -0.01 * (c + 1);
            s->m_camPos.add(frame, c, R, t);
        }
        // (GPUKLT::init would create the device tracker: the slot table is all chooseStatic / chooseDynamicFeatPts look at)
        s->m_tracker.m_nMaxCorners = 1200;
        s->m_tracker.m_tks = new Track2D[1200];
        N = s->m_tracker.m_nMaxCorners;
        xy[c].assign(2 * N, 0.0), state[c].assign(N, -1), s2m[c].assign(N, -1), sel[c].assign(PTS, -1), ftype[c].assign(N, 0), span[c].assign(2 * N, -1);
        // slots: a random subset carries a track; a track's tail is a feature of THIS frame; about half of them mapped
        for (int i = 0; i < N; ++i) {
            if (urand() < 0.35) continue;                       // an empty slot
            Track2D& tk = s->m_tracker.m_tks[i];
            const int len = 1 + (int)(urand() * 6);
            const double x = 2 + (W - 4) * urand(), y = 2 + (H - 4) * urand();
            FeaturePoint* fp = 0;
            for (int j = len - 1; j >= 0; --j) {
                fp = s->m_featPts.add(frame - j, c, x - 0.8 * j, y + 0.5 * j);
                tk.add(fp);
            }
            fp->type = urand() < 0.7 ? TYPE_FEATPOINT_STATIC : TYPE_FEATPOINT_DYNAMIC;
            xy[c][i] = fp->x, xy[c][N + i] = fp->y;
            state[c][i] = len == 1 ? 1 : 0;
            ftype[c][i] = fp->type == TYPE_FEATPOINT_STATIC ? 1 : 0;
            span[c][i] = frame - (len - 1), span[c][N + i] = frame;
            if (urand() < 0.55) {
                // a map point no other slot of this camera carries yet (MapPoint::pFeatures[c] is one feature per camera)
                for (int tries = 0; tries < 8; ++tries) {
                    const bool wantDyn = nDynWanted > 0 && urand() < 0.25;
                    const int m = wantDyn ? nMap - nDynWanted + (int)(urand() * nDynWanted) : (int)(urand() * (nMap - nDynWanted));
                    if (mpts[m].pFeatures[c]) continue;
                    fp->mpt = &mpts[m];
                    mpts[m].pFeatures[c] = fp;
                    s2m[c][i] = m;
                    break;
                }
            }
        }
        // stale features: points this camera saw some frames ago and lost (pFeatures[c]->f < curFrame)
        for (int q = 0; q < nMap / 12; ++q) {
            const int m = (int)(urand() * nMap);
            if (mpts[m].pFeatures[c]) continue;
            FeaturePoint* old = s->m_featPts.add(frame - 3, c, 5 + 600 * urand(), 5 + 450 * urand());
            old->mpt = &mpts[m];
            mpts[m].pFeatures[c] = old;
        }
    }
    for (int i = 0; i < nMap; ++i) mpts[i].updateVisCamNum(frame);   // numVisCam: cameras with a feature of THIS frame
    // the hand-back's packing of every camera: chooseStaticFeatPts' list, the features with a map point, as slots
    int nStaticExpected = 0;
    for (int c = 0; c < nc; ++c) {
        SingleSLAM* s = &co->slam[c];
        std::vector<FeaturePoint*> chosen;
        s->chooseStaticFeatPts(chosen);
        int k = 0;
        for (size_t q = 0; q < chosen.size(); ++q) {
            if (!chosen[q]->mpt) continue;
            for (int i = 0; i < N; ++i)
                if (!s->m_tracker.m_tks[i].empty() && s->m_tracker.m_tks[i].tail->pt == chosen[q]) sel[c][k++] = i;
        }
        npts[c] = k;
        nStaticExpected += k;
    }
    // a map point the frame's classification detached between the hand-back and now: its feature loses the point
    for (int c = 0; c < nc; ++c)
        if (npts[c] > 3) {
            const int i = sel[c][2];
            FeaturePoint* fp = co->slam[c].m_tracker.m_tks[i].tail->pt;
            fp->mpt->pFeatures[c] = 0;
            fp->mpt->updateVisCamNum(frame);
            fp->mpt = 0;
            s2m[c][i] = -1;
            --nStaticExpected;
        }
    InterCamPoseEstimator est;
    est.setCoSLAM(co);
    est.addMapPoints();
    (void)nStaticExpected;   // (the reference votes again after the detach: another mapped track of the block may take the place)
    // ---- dump
    const int hd[9] = {nc, N, nMap, frame, W, H, co->slam[0].nColBlk, co->slam[0].nRowBlk, PTS};
    fwrite(hd, 4, 9, f);
    std::vector<double> M(3 * nMap);
    std::vector<int> pf((size_t)nMap * nc, -1);
    for (int i = 0; i < nMap; ++i) {
        M[3 * i] = mpts[i].x, M[3 * i + 1] = mpts[i].y, M[3 * i + 2] = mpts[i].z;
        for (int c = 0; c < nc; ++c) {
            FeaturePoint* fp = mpts[i].pFeatures[c];
            if (!fp || fp->f != frame) continue;
            for (int s = 0; s < N; ++s)
                if (!co->slam[c].m_tracker.m_tks[s].empty() && co->slam[c].m_tracker.m_tks[s].tail->pt == fp) pf[(size_t)i * nc + c] = s;
        }
    }
    put(f, M), put(f, flags), put(f, newPt), put(f, pf);
    for (int c = 0; c < nc; ++c) put(f, xy[c]), put(f, state[c]), put(f, s2m[c]), put(f, ftype[c]), put(f, span[c]);
    const int P = (int)est.vecPts3D.size();
    std::vector<double> pts(3 * P), oxy;
    std::vector<int> optr(P + 1, 0), ocam, pmap(P, -1);
    for (int i = 0; i < P; ++i) {
        pts[3 * i] = est.vecPts3D[i].x, pts[3 * i + 1] = est.vecPts3D[i].y, pts[3 * i + 2] = est.vecPts3D[i].z;
        for (size_t j = 0; j < est.vecMeas2D[i].size(); ++j)
            ocam.push_back(est.vecMeas2D[i][j].viewId), oxy.push_back(est.vecMeas2D[i][j].x), oxy.push_back(est.vecMeas2D[i][j].y);
        optr[i + 1] = (int)ocam.size();
        for (int m = 0; m < nMap; ++m)   // which map point it is: by value (the reference keeps no index)
            if (mpts[m].x == pts[3 * i] && mpts[m].y == pts[3 * i + 1] && mpts[m].z == pts[3 * i + 2]) pmap[i] = m;
    }
    const int tail[4] = {est.m_numStatic, est.m_numDynamic, P, (int)ocam.size()};
    fwrite(tail, 4, 4, f);
    put(f, pts), put(f, optr), put(f, ocam), put(f, oxy), put(f, pmap);
    for (int c = 0; c < nc; ++c) {   // the cameras of the problem and the poses they were taken from
        fwrite(est.Rs[c].data, 8, 9, f), fwrite(est.Ts[c].data, 8, 3, f);
        fwrite(co->slam[c].m_camPos.current()->R, 8, 9, f), fwrite(co->slam[c].m_camPos.current()->t, 8, 3, f);
    }
    printf("scene: %d cameras x %d slots, %d map points: %d static + %d dynamic points, %zu measurements\n", nc, N, nMap, est.m_numStatic,
           est.m_numDynamic, ocam.size());
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3 || strcmp(argv[1], "golden")) {
        fprintf(stderr, "usage: %s golden <out.bin>\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    puti(f, 3);
    int rc = scene(f, 3, 1400, 260);   // far more dynamic candidates than maxDyn + 1
    rc |= scene(f, 2, 900, 30);        // fewer
    rc |= scene(f, 4, 1200, 0);        // none
    fclose(f);
    return rc;
}
