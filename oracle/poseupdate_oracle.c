/* poseupdate_oracle.c -- TEST INFRASTRUCTURE (never linked or loaded by the product): CPU restatement of what a frame does with
 * a camera's new pose, after intraCamEstimate:
 *   SingleSLAM::poseUpdate3D, second half     /root/reference/src/app/SL_SingleSLAM.cpp:672-708  per static mapped track node:
 *       project, getProjectionCovMat, mat22Inv, mahaDist2 against 2.0 (6.0 with largeErr); inlier: reprojErr = the distance,
 *       seqTriangulate updates the map point and its covariance IN PLACE; outlier: reprojErr = the pixel distance, the map
 *       point is setUncertain() (src/slam/SL_MapPoint.cpp:110-113)
 *   SingleSLAM::getStaticMappedTrackNodes     :60-75   the nodes: non-empty tracks in slot order whose tail feature carries a
 *       map point that isCertainStatic() (SL_MapPoint.h:105-107: !bUncertain && iLocalType == TYPE_MAP_STATIC)
 *   SingleSLAM::detectDynamicFeaturePoints    :784-824 per track of length >= minLen whose tail feature is unmapped or on a
 *       certain-dynamic point: walk the track backwards, count the past positions whose epipolar error against the current one
 *       (F from the two frames' poses) is >= maxEpiErr; more than minOutNum: the feature becomes DYNAMIC, else an unmapped one
 *       becomes STATIC.  NOTE :799: the loop's counter `f` is never advanced, so `f < maxLen` never ends the walk -- it runs to
 *       the head of the track or until the count exceeds minOutNum.  Restated as written: maxLen is accepted and has no effect;
 *       the walk is bounded by the history this restatement is handed (H frames).
 *   SingleSLAM::getUnMappedAndDynamicTrackNodes :91-105
 *   propagateFeatureStates' type / reprojErr hand-down along a track (:34-58) is the per-slot persistence of the two arrays.
 * The cameras of a frame run one after the other (CoSLAM::parallelPoseUpdate, src/app/SL_CoSLAM.cpp:398-410): a map point seen
 * by several cameras is updated in camera order, and a point one camera made uncertain is no node of the next.
 *
 * The LOOPS are pinned against the reference's own SL_SingleSLAM.cpp compiled in place (tests/cxx/ref_ba_dropin_test.cpp, part
 * 3, through tests/test_cxx_dropin_gpu.py).  PARITY UNPINNED for the helpers they call, which live in un-vendored LibVisualSLAM
 * (only their calls are in the reference): project, getProjectionCovMat, mat22Inv, mahaDist2, dist2 as in register_oracle.c;
 *   seqTriangulate(K, R, t, m, M, cov, sigma)   one Kalman update of (M, cov) from the measurement m with noise sigma^2 I:
 *       J = d project / dM at M, S = J cov J^T + sigma^2 I, G = cov J^T S^-1, M += G (m - project(M)), cov -= G (cov J^T)^T
 *   formEMat(R1, t1, R2, t2, E)                 E = [t]x R with R = R2 R1^T, t = t2 - R t1
 *   getFMat(iK1, iK2, E, F)                     F = iK2^T E iK1
 *   epipolarError(F, a, b)                      distance of a from the line F (b, 1)
 * A slot's feature of this frame: hand-back state 0 (tracked) or 1 (new).  Map flags: bit 0 dynamic, bit 1 false, bit 2 uncertain. */
#include <math.h>
#include <string.h>

#include "klt_oracle.h"

#define OPU_DYNAMIC 1
#define OPU_FALSE 2
#define OPU_UNCERTAIN 4

static int certain_static(unsigned char f) { return (f & (OPU_DYNAMIC | OPU_FALSE | OPU_UNCERTAIN)) == 0; }
static int certain_dynamic(unsigned char f) { return (f & (OPU_DYNAMIC | OPU_FALSE | OPU_UNCERTAIN)) == OPU_DYNAMIC; }

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

void opu_seq_triangulate(const double K[9], const double R[9], const double t[3], const double m[2], double M[3], double cov[9],
                         double sigma) {
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
    double J[6], PJt[6], S[4], iS[4], G[6];
    for (int j = 0; j < 3; j++) {
        J[j] = (KR[j] * w - u * KR[6 + j]) / ww;
        J[3 + j] = (KR[3 + j] * w - v * KR[6 + j]) / ww;
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 2; c++) PJt[2 * r + c] = (cov[3 * r] * J[3 * c] + cov[3 * r + 1] * J[3 * c + 1]) + cov[3 * r + 2] * J[3 * c + 2];
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) {
            const double s = (J[3 * r] * PJt[c] + J[3 * r + 1] * PJt[2 + c]) + J[3 * r + 2] * PJt[4 + c];
            S[2 * r + c] = (r == c) ? s + sigma * sigma : s;
        }
    mat22_inv(S, iS);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 2; c++) G[2 * r + c] = PJt[2 * r] * iS[c] + PJt[2 * r + 1] * iS[2 + c];
    const double e0 = m[0] - u / w, e1 = m[1] - v / w;
    double nc[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) nc[3 * r + c] = cov[3 * r + c] - (G[2 * r] * PJt[2 * c] + G[2 * r + 1] * PJt[2 * c + 1]);
    for (int r = 0; r < 3; r++) M[r] = M[r] + (G[2 * r] * e0 + G[2 * r + 1] * e1);
    memcpy(cov, nc, sizeof(nc));
}

/* the gate loop of ONE camera (:672-708): slots in order.  reprojErr [N] in/out, numOut out; returns the number of nodes. */
int opu_gate_camera(const double K[9], const double R[9], const double t[3], int N, const double* xy, const int* state,
                    const int* slot2map, int nMap, double* mapPts, double* mapCov, unsigned char* mapFlags, int largeErr,
                    double sigma, double* reprojErr, int* numOut) {
    const double errThres = largeErr ? 6.0 : 2.0; /* :673 */
    int num = 0, nOut = 0;
    for (int i = 0; i < N; i++) {
        if (!(state[i] == 0 || state[i] == 1)) continue;
        const int mp = slot2map[i];
        if (mp < 0 || mp >= nMap || !certain_static(mapFlags[mp])) continue; /* getStaticMappedTrackNodes */
        num++;
        double* pM = mapPts + 3 * (size_t)mp;
        double* pCov = mapCov + 9 * (size_t)mp;
        double rm[2], var[4], ivar[4];
        org_project(K, R, t, pM, rm);                     /* :678 */
        org_projection_cov(K, R, t, pM, pCov, var, sigma); /* :679 */
        mat22_inv(var, ivar);
        const double err = maha_dist2(rm, xy[i], xy[N + i], ivar); /* :681 */
        if (err < errThres) {
            reprojErr[i] = err;                                        /* :683 */
            const double m[2] = {xy[i], xy[N + i]};
            opu_seq_triangulate(K, R, t, m, pM, pCov, sigma);          /* :684-685 (what follows there recomputes err and drops it) */
        } else {
            nOut++;
            const double dx = rm[0] - xy[i], dy = rm[1] - xy[N + i];
            reprojErr[i] = sqrt(dx * dx + dy * dy);                    /* :701-702 dist2 */
            mapFlags[mp] |= OPU_UNCERTAIN;                             /* :704 */
        }
    }
    *numOut = nOut;
    return num;
}

static void mat33_ab(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, T, sizeof(T));
}
void opu_form_emat(const double* R1, const double* t1, const double* R2, const double* t2, double* E) {
    const double R1t[9] = {R1[0], R1[3], R1[6], R1[1], R1[4], R1[7], R1[2], R1[5], R1[8]};
    double R[9], t[3];
    mat33_ab(R2, R1t, R);
    for (int i = 0; i < 3; ++i) t[i] = t2[i] - (R[3 * i] * t1[0] + R[3 * i + 1] * t1[1] + R[3 * i + 2] * t1[2]);
    const double Tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    mat33_ab(Tx, R, E);
}
void opu_get_fmat(const double* iK1, const double* iK2, const double* E, double* F) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += iK2[3 * k + i] * E[3 * k + j];
            T[3 * i + j] = s;
        }
    mat33_ab(T, iK1, F);
}
double opu_epipolar_error(const double* F, double ax, double ay, double bx, double by) {
    const double l0 = F[0] * bx + F[1] * by + F[2], l1 = F[3] * bx + F[4] * by + F[5], l2 = F[6] * bx + F[7] * by + F[8];
    const double n = sqrt(l0 * l0 + l1 * l1);
    return fabs(l0 * ax + l1 * ay + l2) / (n > 0 ? n : 1.0);
}

/* detectDynamicFeaturePoints of ONE camera.  History: the last H frames, entry j = the frame j steps back (j = 0: this frame):
 *   histR [H][9], histT [H][3] the camera's pose of that frame (FeaturePoint::cam), histXY [H][2N] the slots' undistorted pixels;
 *   a slot's track covers the len = last - first + 1 newest entries (trackSpan).  isStatic [N] in/out (FeaturePoint::type ==
 *   TYPE_FEATPOINT_STATIC).  Returns the number of features made dynamic. */
int opu_detect_dynamic_camera(const double iK[9], int N, int H, int nHist, const double* histR, const double* histT,
                              const double* histXY, const int* state, const int* slot2map, const int* trackSpan, int nMap,
                              const unsigned char* mapFlags, int maxLen, int minLen, int minOutNum, double maxEpiErr,
                              unsigned char* isStatic) {
    (void)maxLen; /* :799: `f` is never advanced */
    (void)H;
    int k = 0;
    for (int i = 0; i < N; i++) {
        if (!(state[i] == 0 || state[i] == 1)) continue;
        if (state[i] == 1) isStatic[i] = 1; /* a new FeaturePoint: type(0) = static (src/slam/SL_FeaturePoint.cpp:23) */
        const int len = trackSpan[i] >= 0 ? trackSpan[N + i] - trackSpan[i] + 1 : 0;
        if (len < minLen) continue; /* :96 */
        const int mp = slot2map[i];
        const int mapped = mp >= 0 && mp < nMap;
        if (mapped && !certain_dynamic(mapFlags[mp])) continue; /* :99 */
        const double* R0 = histR;
        const double* t0 = histT;
        const double m0x = histXY[i], m0y = histXY[N + i];
        int nOut = 0;
        const int depth = len < nHist ? len : nHist;
        for (int j = 0; j < depth && nOut <= minOutNum; j++) { /* :799 */
            double E[9], F[9];
            opu_form_emat(histR + 9 * (size_t)j, histT + 3 * (size_t)j, R0, t0, E); /* :806 */
            opu_get_fmat(iK, iK, E, F);
            const double* h = histXY + (size_t)j * 2 * N;
            if (opu_epipolar_error(F, m0x, m0y, h[i], h[N + i]) >= maxEpiErr) nOut++; /* :809-811 */
        }
        if (nOut > minOutNum) {
            isStatic[i] = 0; /* TYPE_FEATPOINT_DYNAMIC */
            k++;
        } else if (!mapped) {
            isStatic[i] = 1;
        }
    }
    return k;
}
