// ref_newpts_test.cpp -- the reference's OWN featTracksFromMatches (src/app/SL_NewMapPointsInterCam.cpp:631-690),
// NewMapPtsNCC::reconstructTracks (:194-270) and NewMapPts::decidePointType (:25-91) on cameras, features and matches built with the
// reference's classes; writes the inputs as structure-of-arrays records and what the reference made of them (the tracks in their
// numbering, the new map points with position, covariance, type, features and the features' reprojErr), for
// tests/golden/make_golden.py (CPU only).
//
// oracle/Makefile compiles SL_NewMapPointsInterCam.cpp IN PLACE (never copied) against oracle/ref_shim/.  matchBetween (NCC blocks,
// the fundamental matrix, the un-vendored greedy matchers) is not on this driver's path: the matches are given (one-to-one lists per
// consecutive camera pair: true correspondences, wrong ones that the re-projection gate must reject, chains over three and more
// cameras, features matched INTO but not onward); the names matchBetween needs are defined here as aborts so that the class links.
// Certain-dynamic map points with features of this frame drive decidePointType's mask (a new uncertain point with a feature within
// 20 pixels of one stays uncertain, every other becomes certain static: setLocalStatic() clears bUncertain).
//   ref_newpts_test golden <out.bin>
// out.bin (int32 / float64): nScenes; per scene: nc N frame W H; K[9]; per camera R[9] t[3]; per camera xy[2N] (x then y),
// isStatic[N] (int32: FeaturePoint::type == STATIC), nDynFeat, dynXY[nDynFeat][2] (features of this frame on CERTAIN dynamic map
// points), nOther, otherXY[nOther][2] (on uncertain-dynamic / static points: must not count); per pair nMatch, (i, j)[nMatch]; then
// the reference's result: nTracks; per track len, (cam, idx)[len]; nNew; per new point M[3] cov[9] flags (1 dynamic, 2 false, 4
// uncertain) firstFrame feat[nc] (index into the camera's list or -1); per camera reprojErr[N] (0 where never written).
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "app/SL_CoSLAM.h"
#include "app/SL_NewMapPointsInterCam.h"

// ---- matchBetween's callees: not on this driver's path
static void off_path(const char* what) {
    fprintf(stderr, "ref_newpts_test: %s is not on this driver's path\n", what);
    abort();
}
void getDisparityMat(const Mat_d&, const Mat_d&, const Mat_d&, const Mat_d&, double, Mat_d&) { off_path("getDisparityMat"); }
int greedyGuidedNCCMatch(const Mat_d&, const Mat_d&, Matching&) { off_path("greedyGuidedNCCMatch"); return 0; }
int greedyNCCMatch(const Mat_d&, Matching&) { off_path("greedyNCCMatch"); return 0; }
void getNCCBlocks(const ImgG&, Mat_d&, PtrVec<NCCBlock>&, double) { off_path("getNCCBlocks"); }
void getEpiNccMat(const double*, const Mat_d&, const Mat_d&, const PtrVec<NCCBlock>&, const PtrVec<NCCBlock>&, double, double, Mat_d&, Mat_d&, double) {
    off_path("getEpiNccMat");
}
bool NCCBlock::computeScaled(const ImgG&, double, double, double) { off_path("NCCBlock::computeScaled"); return false; }   // (NewMapPtsNCC::output)

static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
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

static int scene(FILE* f, int nc, int N, int nDynPts) {
    const int W = 640, H = 480, frame = 57;
    const double K[9] = {0.82 * W, 0, W / 2.0, 0, 0.82 * W, H / 2.0, 0, 0, 1};
    const double iK[9] = {1 / K[0], 0, -K[2] / K[0], 0, 1 / K[4], -K[5] / K[4], 0, 0, 1};
    const double kud[7] = {0, 0, 0, 0, 0, 0, 0};
    CoSLAM* co = (CoSLAM*)calloc(1, sizeof(CoSLAM));   // (the constructor lives with the GUI; numCams / curFrame / slam[] are what is touched)
    co->numCams = nc;
    co->curFrame = frame;
    NewMapPtsNCC* np = new NewMapPtsNCC();
    np->numCams = nc, np->pCoSLAM = co, np->m_curFrame = frame;
    std::vector<std::vector<double> > Rc(nc, std::vector<double>(9)), tc(nc, std::vector<double>(3));
    for (int c = 0; c < nc; ++c) {
        SingleSLAM* s = new (&co->slam[c]) SingleSLAM();
        s->camId = c, s->W = W, s->H = H;
        s->K.cloneFrom(K, 3, 3), s->iK.cloneFrom(iK, 3, 3), s->k_ud.cloneFrom(kud, 7, 1);
        const double a = 0.06 * (c - 0.5 * (nc - 1));   // a fan around the scene: rotation about y, the centre moved along x
        double R[9] = {cos(a), 0, -sin(a), 0, 1, 0, sin(a), 0, cos(a)};
        const double C[3] = {1.4 * (c - 0.5 * (nc - 1)), 0.05 * c, 0.0};
        double t[3];
        for (int i = 0; i < 3; ++i) t[i] = -(R[3 * i] * C[0] + R[3 * i + 1] * C[1] + R[3 * i + 2] * C[2]);
        s->m_camPos.add(frame, c, R, t);
        s->m_tracker.m_frame = frame;
        memcpy(Rc[c].data(), R, 72), memcpy(tc[c].data(), t, 24);
        np->m_camGroup.addCam(c);
        memcpy(np->m_K[c], K, 72), memcpy(np->m_invK[c], iK, 72), memcpy(np->m_R[c], R, 72), memcpy(np->m_t[c], t, 24);
    }
    // scene points and their (noisy) projections: feature i of camera c shows point ptOf[c][i]
    const int nScenePts = 3 * N;
    std::vector<double> P(3 * nScenePts);
    for (int p = 0; p < nScenePts; ++p) P[3 * p] = -4.5 + 9 * urand(), P[3 * p + 1] = -2.5 + 5 * urand(), P[3 * p + 2] = 7 + 7 * urand();
    auto project = [&](int c, int p, double& u, double& v) {
        const double* R = Rc[c].data();
        const double* t = tc[c].data();
        const double X = R[0] * P[3 * p] + R[1] * P[3 * p + 1] + R[2] * P[3 * p + 2] + t[0];
        const double Y = R[3] * P[3 * p] + R[4] * P[3 * p + 1] + R[5] * P[3 * p + 2] + t[1];
        const double Z = R[6] * P[3 * p] + R[7] * P[3 * p + 1] + R[8] * P[3 * p + 2] + t[2];
        u = K[0] * X / Z + K[2], v = K[4] * Y / Z + K[5];
        return Z > 0 && u > 3 && u < W - 3 && v > 3 && v < H - 3;
    };
    std::vector<std::vector<int> > ptOf(nc), isStatic(nc);
    std::vector<std::vector<double> > xy(nc);
    for (int c = 0; c < nc; ++c) {
        SingleSLAM* s = &co->slam[c];
        xy[c].assign(2 * N, 0.0), ptOf[c].assign(N, -1), isStatic[c].assign(N, 1);
        np->pFeatPts[c].clear();
        int i = 0;
        for (int p = 0; p < nScenePts && i < N; ++p) {
            double u, v;
            if (!project(c, p, u, v) || urand() < 0.3) continue;
            u += 0.5 * (urand() - 0.5), v += 0.5 * (urand() - 0.5);
            FeaturePoint* fp = s->m_featPts.add(frame, c, u, v);
            fp->setIntrinsic(s->K.data), fp->setCameraPose(s->m_camPos.current());
            fp->type = urand() < 0.75 ? TYPE_FEATPOINT_STATIC : TYPE_FEATPOINT_DYNAMIC;
            np->pFeatPts[c].push_back(fp);
            xy[c][i] = fp->x, xy[c][N + i] = fp->y, ptOf[c][i] = p, isStatic[c][i] = fp->type == TYPE_FEATPOINT_STATIC ? 1 : 0;
            ++i;
        }
        for (; i < N; ++i) {   // (the list is N long: fill with features of points nobody else sees)
            FeaturePoint* fp = s->m_featPts.add(frame, c, 5 + (W - 10) * urand(), 5 + (H - 10) * urand());
            fp->setIntrinsic(s->K.data), fp->setCameraPose(s->m_camPos.current());
            fp->type = TYPE_FEATPOINT_STATIC;
            np->pFeatPts[c].push_back(fp);
            xy[c][i] = fp->x, xy[c][N + i] = fp->y;
        }
    }
    // map points that are already there: certain dynamic ones (their features of this frame make decidePointType's mask), and
    // uncertain-dynamic / static ones whose features must not
    std::vector<std::vector<double> > dynXY(nc), otherXY(nc);
    MapPoint* mpts = (MapPoint*)calloc(nDynPts + 16, sizeof(MapPoint));
    for (int q = 0; q < nDynPts + 16; ++q) {
        MapPoint* mp = new (&mpts[q]) MapPoint(0, 0, 9, 3);
        const bool certainDyn = q < nDynPts;
        if (certainDyn)
            mp->setLocalDynamic();
        else if (q & 1)
            mp->setLocalDynamic(), mp->setUncertain();
        else
            mp->setLocalStatic();
        for (int c = 0; c < nc; ++c) {
            if (urand() < 0.4) continue;
            // near a candidate feature now and then, so that the mask matters
            const int near = (int)(urand() * N);
            const double x = urand() < 0.6 ? xy[c][near] + 30 * (urand() - 0.5) : 5 + (W - 10) * urand();
            const double y = urand() < 0.6 ? xy[c][N + near] + 30 * (urand() - 0.5) : 5 + (H - 10) * urand();
            FeaturePoint* fp = co->slam[c].m_featPts.add(frame, c, x, y);
            fp->mpt = mp;
            mp->pFeatures[c] = fp;
            (certainDyn ? dynXY : otherXY)[c].push_back(fp->x), (certainDyn ? dynXY : otherXY)[c].push_back(fp->y);
        }
    }
    // the matches of consecutive pairs: one-to-one; true correspondences, some wrong ones, some dropped
    std::vector<std::vector<int> > mi(nc - 1), mj(nc - 1);
    Matching* matches = new Matching[SLAM_MAX_NUM];
    for (int a = 0; a + 1 < nc; ++a) {
        std::vector<int> byPt(nScenePts, -1);
        std::vector<char> used(N, 0);
        for (int j = 0; j < N; ++j)
            if (ptOf[a + 1][j] >= 0) byPt[ptOf[a + 1][j]] = j;
        matches[a].clear();
        for (int i = 0; i < N; ++i) {
            int j = -1;
            const double u = urand();
            if (ptOf[a][i] >= 0 && byPt[ptOf[a][i]] >= 0 && u < 0.6)
                j = byPt[ptOf[a][i]];
            else if (u > 0.93)
                j = (int)(urand() * N);   // a wrong match
            if (j < 0 || used[j]) continue;
            used[j] = 1;
            matches[a].add(i, j, 0.0);
            mi[a].push_back(i), mj[a].push_back(j);
        }
    }
    Track2D* tks = new Track2D[SLAM_MAX_TRACKNUM];
    const int ntks = featTracksFromMatches(nc, np->pFeatPts, matches, tks);
    const int npts = np->reconstructTracks(tks, ntks, frame, np->newMapPts, 2, 3.0);
    np->decidePointType();
    // ---- dump
    const int hd[5] = {nc, N, frame, W, H};
    fwrite(hd, 4, 5, f);
    fwrite(K, 8, 9, f);
    for (int c = 0; c < nc; ++c) put(f, Rc[c]), put(f, tc[c]);
    for (int c = 0; c < nc; ++c) {
        put(f, xy[c]), put(f, isStatic[c]);
        puti(f, (int)dynXY[c].size() / 2), put(f, dynXY[c]);
        puti(f, (int)otherXY[c].size() / 2), put(f, otherXY[c]);
    }
    for (int a = 0; a + 1 < nc; ++a) {
        puti(f, (int)mi[a].size());
        for (size_t q = 0; q < mi[a].size(); ++q) puti(f, mi[a][q]), puti(f, mj[a][q]);
    }
    auto index_of = [&](int c, const FeaturePoint* fp) {
        for (int i = 0; i < N; ++i)
            if (np->pFeatPts[c][i] == fp) return i;
        return -1;
    };
    puti(f, ntks);
    int nLong = 0;
    for (int k = 0; k < ntks; ++k) {
        puti(f, tks[k].length());
        nLong += tks[k].length() >= 3;
        for (Track2DNode* nd = tks[k].head.next; nd; nd = nd->next) puti(f, nd->f), puti(f, index_of(nd->f, nd->pt));
    }
    puti(f, (int)np->newMapPts.size());
    int nStatic = 0, nUnc = 0, nDyn = 0;
    for (size_t q = 0; q < np->newMapPts.size(); ++q) {
        MapPoint* mp = np->newMapPts[q];
        fwrite(mp->M, 8, 3, f), fwrite(mp->cov, 8, 9, f);
        const int fl = (mp->isLocalDynamic() ? 1 : 0) | (mp->isFalse() ? 2 : 0) | (mp->isUncertain() ? 4 : 0);
        puti(f, fl), puti(f, mp->firstFrame);
        nStatic += fl == 0, nUnc += fl == 4, nDyn += fl == 1;
        for (int c = 0; c < nc; ++c) puti(f, mp->pFeatures[c] ? index_of(c, mp->pFeatures[c]) : -1);
    }
    for (int c = 0; c < nc; ++c) {
        std::vector<double> rp(N);
        for (int i = 0; i < N; ++i) rp[i] = np->pFeatPts[c][i]->reprojErr;
        put(f, rp);
    }
    printf("scene: %d cameras x %d features: %d tracks (%d over three or more cameras), %d new points (%d), %d certain static, %d uncertain, %d dynamic\n",
           nc, N, ntks, nLong, (int)np->newMapPts.size(), npts, nStatic, nUnc, nDyn);
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
    int rc = scene(f, 4, 400, 40);
    rc |= scene(f, 3, 250, 0);     // no dynamic points: every uncertain new point becomes certain static
    rc |= scene(f, 6, 300, 120);
    fclose(f);
    return rc;
}
