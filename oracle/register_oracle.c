/*
 * oracle/register_oracle.c -- CPU restatement of the search step of CoSLAM's map-point registration.
 *
 * TEST INFRASTRUCTURE ONLY (see klt_oracle.h).  Follows, statement by statement, the part that is the same in the three
 * registration loops of the reference -- for one map point p and every camera of the group:
 *   CoSLAM::curStaticPointRegInGroup        /root/reference/src/app/SL_CoSLAM.cpp:731-757   (cov sigma = pixelErrVar,
 *                                                                                           maxDist = 3 pixelErrVar)
 *   CoSLAM::curDynamicPointRegInGroup       /root/reference/src/app/SL_CoSLAM.cpp:955-980   (same, maxDist = 4 pixelErrVar)
 *   CoSLAM::activeMapPointRegisterInGroup   /root/reference/src/app/SL_CoSLAM.cpp:1118-1145 (cov sigma = 2.5 pixelErrVar,
 *                                                                                           maxDist = 3 pixelErrVar)
 *   searchMahaNearestFeatPt                 /root/reference/src/app/SL_SingleSLAM.cpp:1141-1164
 *   first iteration of staticCheckMergability   SL_CoSLAM.cpp:714-729 (the candidate feature itself, p = fp)
 * What the loops do with the candidate afterwards (NCC comparison of image blocks, the walk over the candidate's earlier
 * frames, pointer updates, refineMapPoint, checkUnify) is pointer-graph glue and stays with the caller.
 *
 * searchMahaNearestFeatPt is pinned: tests/cxx/ref_register_test.cpp drives the reference's own function (compiled in
 * place from SL_SingleSLAM.cpp) over a FeaturePoints list built with the reference's classes, and
 * tests/golden/register_golden.npz holds its answers.  Note what it really does: the inverse covariance is scaled by
 * 1 / maxDist and the feature with the smallest distance wins WITHOUT any threshold on that distance (there is no
 * `d < 1` test), first in list order on ties (strict <).
 *
 * PARITY UNPINNED for the external LibVisualSLAM helpers isAtCameraBack, project, getProjectionCovMat, mat22Inv,
 * mahaDist2 (only their calls are in the reference).  Definitions used here, in oracle/ref_shim/shim_impl.cpp and in
 * coslam_amd/csrc/register.hip:
 *   isAtCameraBack(R, t, M)                 (R M + t).z < 0
 *   project(K, R, t, M, m)                  m = pi(K (R M + t)), row-major 3 x 3
 *   getProjectionCovMat(K,R,t,M,cov,var,s)  var = J cov J^T + s^2 I, J = d pi(K (R M + t)) / dM (2 x 3)
 *   mat22Inv                                adjugate / determinant
 *   mahaDist2(a, b, ivar)                   (a - b)^T ivar (a - b)
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include "klt_oracle.h"

int org_is_at_camera_back(const double R[9], const double t[3], const double M[3]) {
    const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
    return Z < 0.0;
}

void org_project(const double K[9], const double R[9], const double t[3], const double M[3], double m[2]) {
    const double X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0];
    const double Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1];
    const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
    const double u = (K[0] * X + K[1] * Y) + K[2] * Z;
    const double v = (K[3] * X + K[4] * Y) + K[5] * Z;
    const double w = (K[6] * X + K[7] * Y) + K[8] * Z;
    m[0] = u / w;
    m[1] = v / w;
}

void org_projection_cov(const double K[9], const double R[9], const double t[3], const double M[3], const double cov[9],
                        double var[4], double sigma) {
    const double X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0];
    const double Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1];
    const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
    double KR[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) KR[3 * i + j] = (K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j]) + K[3 * i + 2] * R[6 + j];
    const double u = (K[0] * X + K[1] * Y) + K[2] * Z;
    const double v = (K[3] * X + K[4] * Y) + K[5] * Z;
    const double w = (K[6] * X + K[7] * Y) + K[8] * Z;
    const double ww = w * w;
    double J[6], JC[6];
    for (int j = 0; j < 3; j++) {
        J[j] = (KR[j] * w - u * KR[6 + j]) / ww;
        J[3 + j] = (KR[3 + j] * w - v * KR[6 + j]) / ww;
    }
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) JC[3 * i + j] = (J[3 * i] * cov[j] + J[3 * i + 1] * cov[3 + j]) + J[3 * i + 2] * cov[6 + j];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
            const double s = (JC[3 * i] * J[3 * j] + JC[3 * i + 1] * J[3 * j + 1]) + JC[3 * i + 2] * J[3 * j + 2];
            var[2 * i + j] = (i == j) ? s + sigma * sigma : s;
        }
}

static void mat22_inv(const double A[4], double iA[4]) {
    const double det = A[0] * A[3] - A[1] * A[2];
    iA[0] = A[3] / det;
    iA[1] = -A[1] / det;
    iA[2] = -A[2] / det;
    iA[3] = A[0] / det;
}

static double maha_dist2(const double a[2], double bx, double by, const double ivar[4]) {
    const double dx = a[0] - bx, dy = a[1] - by;
    return dx * (ivar[0] * dx + ivar[1] * dy) + dy * (ivar[2] * dx + ivar[3] * dy);
}

/* searchMahaNearestFeatPt(featPts, f, m, var, maxDist), SL_SingleSLAM.cpp:1141-1164.  The frame's list is the slots
 * GPUKLT::addToFeaturePoints added this frame (state 0 or 1), in slot order; xy = x[N] then y[N] (undistorted pixels).
 * Returns the slot or -1 when the frame has no feature; *dmin = the scaled distance of the winner. */
int org_search_maha_nearest(int N, const double* xy, const int* state, const double m[2], const double var[4], double maxDist,
                            double* dmin) {
    double ivar[4];
    mat22_inv(var, ivar);                                    /* :1148 */
    for (int k = 0; k < 4; k++) ivar[k] = ivar[k] * (1 / maxDist); /* :1149 matScale(2, 2, ivar, 1 / maxDist, ivar) */
    double dMin = DBL_MAX;
    int pMin = -1;
    for (int i = 0; i < N; i++) {
        if (state[i] != 0 && state[i] != 1) continue;
        const double d = maha_dist2(m, xy[i], xy[N + i], ivar); /* :1157 */
        if (d < dMin) {                                       /* :1158: no threshold, first wins ties */
            dMin = d;
            pMin = i;
        }
    }
    if (dmin) *dmin = dMin;
    return pMin;
}

/* The search step of the three registration loops for P map points x nCams cameras.
 *   Ks, Rs, ts        nCams x 9 / 9 / 3: slam[iCam].K, m_camPos.current()->R, ->t
 *   xy, state, slot2map, isDynamic   per camera (pointer arrays): this frame's hand-back records (N slots)
 *   Ms, covs          P x 3, P x 9: MapPoint::M, MapPoint::cov
 *   pointFeat         P x nCams: slot of p->pFeatures[iCam] if that feature is of the current frame, else -1 (:737-738)
 *   sigmaSearch       sigma passed to getProjectionCovMat for the search (pixelErrVar, or 2.5 pixelErrVar: :1131-1134)
 *   maxDist           3 pixelErrVar / 4 pixelErrVar (:756, :979, :1143)
 *   sigmaMerge        pixelVar of staticCheckMergability (pixelErrVar)
 * Outputs, P x nCams each: slot (>= 0 the nearest feature; -1 the point already has a feature of this frame in the camera;
 * -2 behind the camera; -3 projects outside the image; -4 the frame has no features), m (x2), var (x4), dist (scaled),
 * flags (bit 0: candidate has no map point, FeaturePoint::mpt == 0; bit 1: candidate is TYPE_FEATPOINT_DYNAMIC; bit 2:
 * mahaDist2(project(M), candidate, inv(var(sigmaMerge))) <= 1, the candidate's own term of staticCheckMergability). */
void org_register_search(int nCams, int N, int W, int H, const double* Ks, const double* Rs, const double* ts,
                         const double* const* xy, const int* const* state, const int* const* slot2map,
                         const unsigned char* const* isDynamic, int P, const double* Ms, const double* covs,
                         const int* pointFeat, double sigmaSearch, double maxDist, double sigmaMerge, int* slot, double* m_out,
                         double* var_out, double* dist, int* flags) {
    for (int p = 0; p < P; p++)
        for (int c = 0; c < nCams; c++) {
            const size_t o = (size_t)p * nCams + c;
            const double *K = Ks + 9 * c, *R = Rs + 9 * c, *t = ts + 3 * c, *M = Ms + 3 * p;
            double m[2] = {0, 0}, var[4] = {0, 0, 0, 0};
            slot[o] = -1;
            dist[o] = 0;
            flags[o] = 0;
            memset(m_out + 2 * o, 0, 2 * sizeof(double));
            memset(var_out + 4 * o, 0, 4 * sizeof(double));
            if (pointFeat[o] >= 0) continue;             /* :737-738 */
            if (org_is_at_camera_back(R, t, M)) {        /* :740-742 */
                slot[o] = -2;
                continue;
            }
            org_project(K, R, t, M, m);                  /* :744-745 */
            memcpy(m_out + 2 * o, m, sizeof(m));
            if (m[0] < 0 || m[0] >= W || m[1] < 0 || m[1] >= H) { /* :746-748 */
                slot[o] = -3;
                continue;
            }
            org_projection_cov(K, R, t, M, covs + 9 * p, var, sigmaSearch); /* :750-753 */
            memcpy(var_out + 4 * o, var, sizeof(var));
            double d;
            const int s = org_search_maha_nearest(N, xy[c], state[c], m, var, maxDist, &d); /* :755-756 */
            if (s < 0) {
                slot[o] = -4;
                continue;
            }
            slot[o] = s;
            dist[o] = d;
            int fl = 0;
            if (slot2map[c][s] < 0) fl |= 1;             /* :759 pFeat->mpt == 0 */
            if (isDynamic[c] && isDynamic[c][s]) fl |= 2; /* :758 pFeat->type */
            {                                            /* staticCheckMergability, first iteration (:716-725) */
                double v2[4], iv[4];
                org_projection_cov(K, R, t, M, covs + 9 * p, v2, sigmaMerge);
                mat22_inv(v2, iv);
                if (!(maha_dist2(m, xy[c][s], xy[c][N + s], iv) > 1.0)) fl |= 4;
            }
            flags[o] = fl;
        }
}

/* CoSLAM::staticCheckMergability(mp, fp, pixelVar), SL_CoSLAM.cpp:714-729: the candidate feature AND every earlier feature of its
 * track (fp, fp->preFrame, ...) must lie within Mahalanobis distance 1 of the map point's projection under the pose of its own
 * frame (p->cam: the pose that frame was given), covariance J cov J^T + pixelVar^2 I; the walk stops at the first failure.
 * History as in poseupdate_oracle.c: entry j = the frame j steps back (0: this frame): histR [nHist][9], histT [nHist][3],
 * histXY [nHist][2N]; the track of `slot` covers the len newest entries.  Returns 1 (mergeable) or 0.
 * (Pinned: tests/golden/mergability_golden.npz holds results of the reference's own function compiled in place.) */
int org_static_check_mergability(const double K[9], int nHist, const double* histR, const double* histT, const double* histXY, int N,
                                 int slot, int len, const double M[3], const double cov[9], double pixelVar) {
    const int depth = len < nHist ? len : nHist;
    for (int j = 0; j < depth; j++) {
        double rm[2], var[4], ivar[4];
        org_project(K, histR + 9 * (size_t)j, histT + 3 * (size_t)j, M, rm);
        org_projection_cov(K, histR + 9 * (size_t)j, histT + 3 * (size_t)j, M, cov, var, pixelVar);
        mat22_inv(var, ivar);
        const double* h = histXY + (size_t)j * 2 * N;
        if (maha_dist2(rm, h[slot], h[N + slot], ivar) > 1.0) return 0;
    }
    return 1;
}

/* ... for every candidate of one camera's column of a registration search (slot[p * slotStride]: the candidate's slot or < 0);
 * out[p * slotStride] = 1 mergeable, 0 not, 255 no candidate */
void org_register_mergability_cam(const double K[9], int nHist, const double* histR, const double* histT, const double* histXY, int N,
                                  const int* trackSpan, int P, const double* Ms, const double* covs, const int* slot, int slotStride,
                                  double pixelVar, unsigned char* out) {
    for (int p = 0; p < P; p++) {
        const int s = slot[(size_t)p * slotStride];
        if (s < 0) {
            out[(size_t)p * slotStride] = 255;
            continue;
        }
        const int len = trackSpan[s] >= 0 ? trackSpan[N + s] - trackSpan[s] + 1 : 0;
        /* a track longer than the history: the reference walks the whole preFrame chain, the frames beyond the history cannot be judged ->
         * "2" whatever the held frames say (cs_register_mergability_dev's convention: never attached, not walked) */
        int v = len > nHist ? 2 : org_static_check_mergability(K, nHist, histR, histT, histXY, N, s, len, Ms + 3 * (size_t)p, covs + 9 * (size_t)p, pixelVar);
        out[(size_t)p * slotStride] = (unsigned char)v;
    }
}

/* The decision half of CoSLAM::curStaticPointsRegInGroup / curDynamicPointsRegInGroup with bMerge == false (reference
 * src/app/SL_CoSLAM.cpp:854-898, 731-830, 904-1020) over the tables of ONE search: for every camera o in order the certainly static
 * (kinds bit 0) -- then the certainly dynamic (bit 1) -- points with a feature of this frame in o, in map order; each walks the cameras,
 * passes by those where it holds a feature, where nothing was found or the candidate's type is the other kind's, attaches the candidate
 * when that is unmapped and mergeable over its whole track, and stops at one that carries a point.  The same walks as
 * oracle/__init__.py's register_decide_static (which this is checked against), in C for the CPU baseline.
 * slot / flags / mergeable: P x nCams; mapFlags [P]; pointFeat [P][nCams] and slot2map [nCams][N] in / out; attached [P][nCams], regged [P]
 * out.  Returns the number of features attached. */
int org_register_decide(int P, int nCams, int N, const int* slot, const int* flags, const unsigned char* mergeable, const unsigned char* mapFlags,
                        int* pointFeat, int* slot2map, int mapBase, int kinds, unsigned char* attached, unsigned char* regged) {
    int nAtt = 0;
    for (size_t k = 0; k < (size_t)P * nCams; k++) attached[k] = 0;
    for (int p = 0; p < P; p++) regged[p] = 0;
    for (int kind = 0; kind < 2; kind++) {
        if (!(kinds & (1 << kind))) continue;
        for (int o = 0; o < nCams; o++)
            for (int p = 0; p < P; p++) {
                if ((mapFlags[p] & 7) != kind || pointFeat[(size_t)p * nCams + o] < 0) continue;
                int breg = 0;
                for (int i = 0; i < nCams; i++) {
                    const size_t k = (size_t)p * nCams + i;
                    if (pointFeat[k] >= 0) continue;
                    const int s = slot[k];
                    if (s < 0 || s >= N) continue;
                    if (((flags[k] >> 1) & 1) != kind) continue;
                    int* owner = slot2map + (size_t)i * N + s;
                    if (*owner >= 0) break;
                    if (mergeable[k] == 1) {
                        *owner = mapBase + p;
                        pointFeat[k] = s;
                        attached[k] = 1;
                        breg = 1, nAtt++;
                    }
                }
                if (breg) regged[p] = 1;
            }
    }
    return nAtt;
}
