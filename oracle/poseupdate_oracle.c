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

/* ---- RobustBundleRTS::updateNewPosesPoints (/root/reference/src/app/SL_CoSLAMRobustBA.cpp:248-271) --------------------------
 * Behind every bundle adjustment + relaxation of the non-key frames, every map point seen after the window's first key frame
 * (MapPoint::lastFrame > firstKeyFrame->f) is triangulated again from the moved poses:
 *   updateStaticPointPosition  (src/slam/SL_CoSLAMHelper.cpp:338-394)   a locally static point: per camera that holds a feature
 *       of it, TWO views -- that feature (pose of its frame) and the feature of the SAME track, further back, whose camera
 *       centre subtends the largest angle with the current one at the point (getAbsRadiansBetween(M, C0, C) > maxAngle,
 *       maxAngle from 0: the first of equal angles in the backward walk wins, an angle of 0 never does) -- then
 *       triangulateMultiView over all views (normalised image points) and getTriangulateCovMat at the new point;
 *   updateDynamicPointPosition (:455-484)   a locally dynamic point: this frame's features only, provided at least one of them
 *       is TYPE_FEATPOINT_DYNAMIC and there are two;
 *   points of the ACTIVE list are only ever updated as static ones (:266-269 tests isLocalStatic() twice).
 * With fewer than two views the point is left alone.  The walk is bounded by the history handed in (nHist frames).
 * PARITY: the loop is pinned against the reference's own SL_CoSLAMHelper.cpp + SL_CoSLAMRobustBA.cpp compiled in place
 * (tests/cxx/ref_update_points_test.cpp -> tests/golden/update_points_golden.npz).  UNPINNED (un-vendored LibVisualSLAM, only
 * their calls are in the reference) are the helpers, defined here as
 *   getCameraCenter(R, t, C)             C = -R^T t
 *   getAbsRadiansBetween(M, C0, C)       the angle at M between C0 - M and C - M, acos(d / sqrt(|a|^2 |b|^2)); compared through
 *                                        its COSINE (cmpAcos = 0: what the kernel does -- no libm on either side) or as the
 *                                        angle itself (cmpAcos = 1: the literal restatement; the two choose the same views
 *                                        unless two cosines differ by less than acos resolves)
 *   getInvK / normPoint(iK, m, nm)       nm = the dehomogenised iK (m, 1); iK is an input here
 *   triangulateMultiView(n, Rs, ts, nms, M)   linear least squares of the 2n equations (r1 - x r3) M = x t3 - t1,
 *                                        (r2 - y r3) M = y t3 - t2 through the normal equations, 3x3 symmetric inverse by cofactors
 *   getTriangulateCovMat(n, Ks, Rs, ts, M, cov, sigma)   cov = sigma^2 (sum_i J_i^T J_i)^-1, J_i = d project_i / dM at M */
static void cam_center(const double* R, const double* t, double* C) {
    for (int i = 0; i < 3; i++) C[i] = -((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]);
}
static double cos_between(const double* M, const double* C0, const double* C) {
    const double a[3] = {C0[0] - M[0], C0[1] - M[1], C0[2] - M[2]}, b[3] = {C[0] - M[0], C[1] - M[1], C[2] - M[2]};
    const double d = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    const double na = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2], nb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    return d / sqrt(na * nb);
}
/* symmetric 3x3: N = {n00, n01, n02, n11, n12, n22}; returns the cofactors c (same order) and the determinant */
static double sym33_cof(const double* N, double* c) {
    c[0] = N[3] * N[5] - N[4] * N[4];
    c[1] = N[2] * N[4] - N[1] * N[5];
    c[2] = N[1] * N[4] - N[2] * N[3];
    c[3] = N[0] * N[5] - N[2] * N[2];
    c[4] = N[1] * N[2] - N[0] * N[4];
    c[5] = N[0] * N[3] - N[1] * N[1];
    return (N[0] * c[0] + N[1] * c[1]) + N[2] * c[2];
}
typedef struct {
    double N[6], g[3];
} opu_normal_eq;
static void ne_add_view(opu_normal_eq* E, const double* iK, const double* R, const double* t, double mx, double my) {
    const double w = (iK[6] * mx + iK[7] * my) + iK[8];
    const double x = ((iK[0] * mx + iK[1] * my) + iK[2]) / w, y = ((iK[3] * mx + iK[4] * my) + iK[5]) / w; /* normPoint */
    const double a0[3] = {R[0] - x * R[6], R[1] - x * R[7], R[2] - x * R[8]}, a1[3] = {R[3] - y * R[6], R[4] - y * R[7], R[5] - y * R[8]};
    const double b0 = x * t[2] - t[0], b1 = y * t[2] - t[1];
    static const int I[6] = {0, 0, 0, 1, 1, 2}, J[6] = {0, 1, 2, 1, 2, 2};
    for (int q = 0; q < 6; q++) E->N[q] = E->N[q] + (a0[I[q]] * a0[J[q]] + a1[I[q]] * a1[J[q]]);
    for (int q = 0; q < 3; q++) E->g[q] = E->g[q] + (a0[q] * b0 + a1[q] * b1);
}
static void cov_add_view(double* S, const double* K, const double* R, const double* t, const double* M) {
    const double X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0];
    const double Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1];
    const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
    double KR[9], Jm[6];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) KR[3 * i + j] = (K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j]) + K[3 * i + 2] * R[6 + j];
    const double u = (K[0] * X + K[1] * Y) + K[2] * Z, v = (K[3] * X + K[4] * Y) + K[5] * Z, w = (K[6] * X + K[7] * Y) + K[8] * Z;
    const double ww = w * w;
    for (int j = 0; j < 3; j++) {
        Jm[j] = (KR[j] * w - u * KR[6 + j]) / ww;
        Jm[3 + j] = (KR[3 + j] * w - v * KR[6 + j]) / ww;
    }
    static const int I[6] = {0, 0, 0, 1, 1, 2}, J[6] = {0, 1, 2, 1, 2, 2};
    for (int q = 0; q < 6; q++) S[q] = S[q] + (Jm[I[q]] * Jm[J[q]] + Jm[3 + I[q]] * Jm[3 + J[q]]);
}

/* A point's feature in a camera AS THE REFERENCE HOLDS IT: MapPoint::pFeatures[c] -- of this frame, or an older one when the camera
 * lost the point (nothing ever clears the pointer) -- with the FeaturePoint::preFrame chain behind it.  The chain is a run of
 * consecutive frames on the feature's own slot, frame - 1 .. first, and then whatever the registration loops linked behind it:
 * `pFeat->preFrame = p->pFeatures[iCam]` (/root/reference/src/app/SL_CoSLAM.cpp:775-779, :997-1000) hangs the point's OLD chain
 * behind the feature of a new track (whose own earlier frames drop out of the chain: the assignment overwrites their link) --
 * segments {slot, last, first, next} of a per-camera pool, `seg` the first of them or -1.
 *   featRef [nMap][nCams][4] = {slot (< 0 none), frame, first, seg};  segPool [nCams][segCap][4] = {slot, last, first, next}.
 * Pixels and poses of a node come from the history at entry curFrame - frame; a node older than the history (entry >= nHist) ends
 * the walk, and so does the walkCap-th node of a chain (the bound every walk of this restatement has; the reference has none). */
typedef struct {
    int curFrame, nHist, walkCap, segCap, N, cmpAcos;
    const int* segPool; /* or NULL */
} opu_chain_ctx;

/* the widest-parallax node behind the feature `ref` of camera c (first of equal angles in the walk, an angle of 0 never):
 * /root/reference/src/slam/SL_CoSLAMHelper.cpp:362-373.  Returns 1 and (slot, history entry) of the node, 0 if there is none. */
static int chain_widest(const opu_chain_ctx* X, int c, const int ref[4], const double* hR, const double* hT, const double* M,
                        const double* C0, int* bestSlot, int* bestEntry) {
    int found = 0, nodes = 1; /* (the feature itself is node 0) */
    double bestCos = 1.0, bestAngle = 0.0;
    int slot = ref[0], hi = ref[1] - 1, lo = ref[2], seg = ref[3];
    for (;;) {
        for (int f = hi; f >= lo; f--) {
            const int j = X->curFrame - f;
            if (j >= X->nHist || nodes >= X->walkCap) return found;
            nodes++;
            double Cj[3];
            cam_center(hR + 9 * (size_t)j, hT + 3 * (size_t)j, Cj);
            const double cv = cos_between(M, C0, Cj);
            if (X->cmpAcos) {
                const double ang = fabs(acos(cv));
                if (ang > bestAngle) bestAngle = ang, *bestSlot = slot, *bestEntry = j, found = 1;
            } else if (cv < bestCos)
                bestCos = cv, *bestSlot = slot, *bestEntry = j, found = 1;
        }
        if (seg < 0 || !X->segPool || seg >= X->segCap) return found;
        const int* g = X->segPool + ((size_t)c * X->segCap + seg) * 4;
        slot = g[0], hi = g[1], lo = g[2], seg = g[3];
    }
}

/* the reference of point m in camera c: from featRef when given, else the feature of this frame pointFeat names with its slot's track
 * behind it (frames counted from curFrame = 0: only differences are used) */
static void chain_ref(const int* featRef, const int* pointFeat, const int* trackSpan, int nCams, int N, int curFrame, int m, int c, int ref[4]) {
    if (featRef) {
        memcpy(ref, featRef + ((size_t)m * nCams + c) * 4, 16);
        return;
    }
    const int s = pointFeat[(size_t)m * nCams + c];
    ref[0] = s, ref[1] = curFrame, ref[2] = curFrame, ref[3] = -1;
    if (s >= 0) {
        const int f1 = trackSpan[(size_t)c * 2 * N + s], f2 = trackSpan[(size_t)c * 2 * N + N + s];
        if (f1 >= 0) ref[2] = curFrame - (f2 - f1);
    }
}

/* Layouts: Ks / iKs [nCams][9]; histR [nCams][nHist][9], histT [nCams][nHist][3], histXY [nCams][nHist][2N] with entry 0 = this
 * frame (the poses as they stand AFTER the adjustment); trackSpan [nCams][2N] (first | last frame of the slot's track),
 * featStatic [nCams][N] (1 = TYPE_FEATPOINT_STATIC); pointFeat [nMap][nCams] = the slot of the point's feature of this frame in
 * that camera, < 0 none; lastFrame [nMap] or NULL (= every point passes :250); isCurrent [nMap] or NULL (= all on curMapPts).
 * featRef / segPool (or NULL): the features as references with their chains (above) -- then pointFeat and trackSpan are not read,
 * stale features count as views with the pose of their own frame, and the walks follow the links.
 * chosen (or NULL): [nMap][nCams] the history entry taken as the second view (-1 none), for the tests.
 * Returns the number of points re-triangulated; *nStat / *nDyn count them by kind. */
static int update_points_core(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR,
                              const double* histT, const double* histXY, const int* trackSpan, const unsigned char* featStatic,
                              int nMap, const int* pointFeat, const int* lastFrame, const unsigned char* isCurrent,
                              int firstKeyFrame, double* mapPts, double* mapCov, const unsigned char* mapFlags, double sigma,
                              int cmpAcos, int* chosen, int* nStat, int* nDyn, int refine, const unsigned char* select,
                              const int* featRef, const int* segPool, int segCap, int curFrame, int walkCap, const unsigned char* refStatic) {
    const opu_chain_ctx X = {featRef ? curFrame : 0, nHist, featRef ? walkCap : nHist, segCap, N, cmpAcos, segPool};
    int nUpd = 0, ns = 0, nd = 0;
    for (int m = 0; m < nMap; m++) {
        if (chosen)
            for (int c = 0; c < nCams; c++) chosen[(size_t)m * nCams + c] = -1;
        if (refine && select && !select[m]) continue;
        if (!refine && lastFrame && lastFrame[m] <= firstKeyFrame) continue; /* :250, :261 */
        const unsigned char fl = refine ? 0 : mapFlags[m]; /* refineMapPoint asks nothing about the point's type */
        const int locStatic = (fl & (OPU_DYNAMIC | OPU_FALSE)) == 0, locDynamic = (fl & (OPU_DYNAMIC | OPU_FALSE)) == OPU_DYNAMIC;
        const int cur = isCurrent ? isCurrent[m] != 0 : 1;
        double* M = mapPts + 3 * (size_t)m;
        opu_normal_eq E;
        memset(&E, 0, sizeof(E));
        int numView = 0, firstE[64], second[64];
        if (locStatic) { /* updateStaticPointPosition */
            for (int c = 0; c < nCams; c++) {
                second[c] = -2;
                int ref[4];
                chain_ref(featRef, pointFeat, trackSpan, nCams, N, X.curFrame, m, c, ref);
                const int s = ref[0], j0 = X.curFrame - ref[1];
                if (s < 0 || j0 >= nHist) continue; /* (a feature older than the history is no view) */
                const double* hR = histR + (size_t)c * nHist * 9;
                const double* hT = histT + (size_t)c * nHist * 3;
                const double* hXY = histXY + (size_t)c * nHist * 2 * N;
                const double* iK = iKs + 9 * c;
                firstE[c] = j0;
                ne_add_view(&E, iK, hR + 9 * (size_t)j0, hT + 3 * (size_t)j0, hXY[(size_t)j0 * 2 * N + s], hXY[(size_t)j0 * 2 * N + N + s]); /* :347-356 */
                numView++;
                double C0[3];
                cam_center(hR + 9 * (size_t)j0, hT + 3 * (size_t)j0, C0);
                int best = -1, bs = s;
                if (!chain_widest(&X, c, ref, hR, hT, M, C0, &bs, &best)) best = -1; /* :362-373 fp = fp->preFrame */
                second[c] = best;
                if (chosen) chosen[(size_t)m * nCams + c] = best;
                if (best >= 0) { /* :374-383 */
                    ne_add_view(&E, iK, hR + 9 * (size_t)best, hT + 3 * (size_t)best, hXY[(size_t)best * 2 * N + bs],
                                hXY[(size_t)best * 2 * N + N + bs]);
                    numView++;
                }
            }
        } else if (locDynamic && cur) { /* updateDynamicPointPosition; :266-269 never reaches it for the active list */
            int nDynamic = 0;
            for (int c = 0; c < nCams; c++) {
                second[c] = -2;
                int ref[4];
                chain_ref(featRef, pointFeat, trackSpan, nCams, N, X.curFrame, m, c, ref);
                const int s = ref[0], j0 = X.curFrame - ref[1];
                if (s < 0 || j0 >= nHist) continue;
                second[c] = -1, firstE[c] = j0;
                ne_add_view(&E, iKs + 9 * c, histR + ((size_t)c * nHist + j0) * 9, histT + ((size_t)c * nHist + j0) * 3,
                            histXY[((size_t)c * nHist + j0) * 2 * N + s], histXY[((size_t)c * nHist + j0) * 2 * N + N + s]);
                numView++;
                /* fp->type: of a stale feature what it was in its own frame -- refStatic [nMap][nCams] when given (the caller's snapshot),
                 * else the slot's entry of featStatic whatever the feature's age */
                if (refStatic ? !refStatic[(size_t)m * nCams + c] : !featStatic[(size_t)c * N + s]) nDynamic++;
            }
            if (nDynamic < 1) continue; /* :475 */
        } else
            continue;
        if (numView < 2) continue; /* :388, :475 */
        double cf[6];
        const double det = sym33_cof(E.N, cf);
        M[0] = ((cf[0] * E.g[0] + cf[1] * E.g[1]) + cf[2] * E.g[2]) / det;
        M[1] = ((cf[1] * E.g[0] + cf[3] * E.g[1]) + cf[4] * E.g[2]) / det;
        M[2] = ((cf[2] * E.g[0] + cf[4] * E.g[1]) + cf[5] * E.g[2]) / det;
        double S[6] = {0, 0, 0, 0, 0, 0};
        for (int c = 0; c < nCams; c++) {
            if (second[c] == -2) continue;
            const double* hR = histR + (size_t)c * nHist * 9;
            const double* hT = histT + (size_t)c * nHist * 3;
            cov_add_view(S, Ks + 9 * c, hR + 9 * (size_t)firstE[c], hT + 3 * (size_t)firstE[c], M);
            if (second[c] >= 0) cov_add_view(S, Ks + 9 * c, hR + 9 * (size_t)second[c], hT + 3 * (size_t)second[c], M);
        }
        const double dS = sym33_cof(S, cf), s2 = sigma * sigma;
        double* cov = mapCov + 9 * (size_t)m;
        cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
        cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
        cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
        nUpd++;
        if (locStatic) ns++; else nd++;
    }
    if (nStat) *nStat = ns;
    if (nDyn) *nDyn = nd;
    return nUpd;
}

int opu_update_new_poses_points(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR,
                                const double* histT, const double* histXY, const int* trackSpan, const unsigned char* featStatic,
                                int nMap, const int* pointFeat, const int* lastFrame, const unsigned char* isCurrent,
                                int firstKeyFrame, double* mapPts, double* mapCov, const unsigned char* mapFlags, double sigma,
                                int cmpAcos, int* chosen, int* nStat, int* nDyn) {
    return update_points_core(nCams, N, nHist, Ks, iKs, histR, histT, histXY, trackSpan, featStatic, nMap, pointFeat, lastFrame, isCurrent,
                              firstKeyFrame, mapPts, mapCov, mapFlags, sigma, cmpAcos, chosen, nStat, nDyn, 0, 0, 0, 0, 0, 0, 0, 0);
}

/* the same with the features as references (featRef / segPool: above update_points_core) */
int opu_update_new_poses_points_ref(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR,
                                    const double* histT, const double* histXY, const unsigned char* featStatic, int nMap,
                                    const int* featRef, const unsigned char* refStatic, const int* segPool, int segCap, int curFrame, int walkCap, const int* lastFrame,
                                    const unsigned char* isCurrent, int firstKeyFrame, double* mapPts, double* mapCov,
                                    const unsigned char* mapFlags, double sigma, int cmpAcos, int* chosen, int* nStat, int* nDyn) {
    return update_points_core(nCams, N, nHist, Ks, iKs, histR, histT, histXY, 0, featStatic, nMap, 0, lastFrame, isCurrent, firstKeyFrame,
                              mapPts, mapCov, mapFlags, sigma, cmpAcos, chosen, nStat, nDyn, 0, 0, featRef, segPool, segCap, curFrame, walkCap, refStatic);
}

/* CoSLAM::refineMapPoint (/root/reference/src/app/SL_CoSLAM.cpp:666-713) for the points `select` names (NULL: all): what the
 * registration loops call on a map point that has just gained a feature (:896, :948, :1166).  The views are those of
 * updateStaticPointPosition -- per camera holding a feature, that feature and the widest-parallax one further back on its track --
 * followed by the same triangulateMultiView + getTriangulateCovMat, whatever the point's type and without a frame test.  The
 * reference does not look at the number of views (with one, its least-squares call is rank deficient); here a point with fewer
 * than two views is left alone.  Pinned like the function above (tests/cxx/ref_update_points_test.cpp calls the reference's own
 * refineMapPoint on a copy of every point that two cameras see).  Returns the number of points refined. */
int opu_refine_map_points(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                          const double* histXY, const int* trackSpan, int nMap, const int* pointFeat, const unsigned char* select,
                          double* mapPts, double* mapCov, double sigma, int cmpAcos) {
    return update_points_core(nCams, N, nHist, Ks, iKs, histR, histT, histXY, trackSpan, 0, nMap, pointFeat, 0, 0, 0, mapPts, mapCov, 0,
                              sigma, cmpAcos, 0, 0, 0, 1, select, 0, 0, 0, 0, 0, 0);
}
int opu_refine_map_points_ref(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                              const double* histXY, int nMap, const int* featRef, const int* segPool, int segCap, int curFrame, int walkCap,
                              const unsigned char* select, double* mapPts, double* mapCov, double sigma, int cmpAcos) {
    return update_points_core(nCams, N, nHist, Ks, iKs, histR, histT, histXY, 0, 0, nMap, 0, 0, 0, 0, mapPts, mapCov, 0, sigma, cmpAcos, 0, 0, 0,
                              1, select, featRef, segPool, segCap, curFrame, walkCap, 0);
}

/* ---- SingleSLAM::newMapPoints (/root/reference/src/app/SL_SingleSLAM.cpp:922-1004) for ONE camera ---------------------------------------
 * What CoSLAM::genNewMapPoints calls for a camera that IsReadyForKeyFrame (SL_CoSLAM.cpp:1310-1330): every unmapped feature on a track of
 * at least minTrackLen frames (getUnMappedAndTrackedFeatPts, :152-172: tk.f2 - tk.f1 >= trackLen) is triangulated from its OWN track --
 * the track's oldest feature within the history (the walk stops at a feature without a pose, :937) against the current one
 * (binTriangulate), thrown out when the point lies behind the camera, when it is nearer to the camera than the square root of its
 * covariance's trace (:960-962), when one of the two views re-projects further off than maxEpiErr; refineTriangulation (:1005-1049) then
 * triangulates the current view with the widest-parallax one of the track, and the tests run once more.  A dynamic feature stops the
 * walk (:938-944): the slot's type (isStatic) stands for the whole track here.
 * hist* [nHist] with entry 0 = this frame; histXY [nHist][2N]; trackSpan [2N]; state / slot2map / isStatic [N].
 * Out, in slot order: newSlot, newFirst (MapPoint::firstFrame = the oldest view's frame), newM [..][3], newCov [..][9].  Returns the count.
 * PARITY: the loop is pinned against the reference's own function compiled in place (tests/cxx/ref_intracam_newpts_test.cpp ->
 * tests/golden/intracam_newpts_golden.npz); binTriangulate / getBinTriangulateCovMat = triangulateMultiView / getTriangulateCovMat over
 * the two views in the order given (un-vendored LibVisualSLAM: our definitions on both sides). */
static void tri2(const double* iK, const double* R1, const double* t1, double x1, double y1, const double* R2, const double* t2, double x2, double y2,
                 double* M) {
    opu_normal_eq E;
    memset(&E, 0, sizeof(E));
    ne_add_view(&E, iK, R1, t1, x1, y1);
    ne_add_view(&E, iK, R2, t2, x2, y2);
    double cf[6];
    const double det = sym33_cof(E.N, cf);
    M[0] = ((cf[0] * E.g[0] + cf[1] * E.g[1]) + cf[2] * E.g[2]) / det;
    M[1] = ((cf[1] * E.g[0] + cf[3] * E.g[1]) + cf[4] * E.g[2]) / det;
    M[2] = ((cf[2] * E.g[0] + cf[4] * E.g[1]) + cf[5] * E.g[2]) / det;
}
static void cov2(const double* K, const double* R1, const double* t1, const double* R2, const double* t2, const double* M, double sigma, double* cov) {
    double S[6] = {0, 0, 0, 0, 0, 0}, cf[6];
    cov_add_view(S, K, R1, t1, M);
    cov_add_view(S, K, R2, t2, M);
    const double dS = sym33_cof(S, cf), s2 = sigma * sigma;
    cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
    cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
    cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
}
static double reproj_err(const double* K, const double* R, const double* t, const double* M, double mx, double my) {
    double rm[2];
    org_project(K, R, t, M, rm);
    const double dx = mx - rm[0], dy = my - rm[1];
    return sqrt(dx * dx + dy * dy);
}
static int behind(const double* R, const double* t, const double* M) { return ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2] < 0; }

int opu_intracam_new_points(const double* K, const double* iK, int N, int nHist, const double* histR, const double* histT, const double* histXY,
                            const int* state, const int* slot2map, const int* trackSpan, const unsigned char* isStatic, int minTrackLen,
                            double maxEpiErr, double sigma, int cmpAcos, int* newSlot, int* newFirst, double* newM, double* newCov) {
    int n = 0;
    for (int k = 0; k < N; k++) {
        if (state[k] != 0 && state[k] != 1) continue;                  /* tk.empty() */
        const int f1 = trackSpan[k], f2 = trackSpan[N + k];
        if (f1 < 0 || f2 - f1 < minTrackLen || slot2map[k] >= 0) continue; /* :159 */
        if (!isStatic[k]) continue;                                     /* :938-944 */
        int jp = f2 - f1;                                               /* the track's first feature ... */
        if (jp > nHist - 1) jp = nHist - 1;                             /* ... or the oldest one that still has a pose (:937) */
        if (jp < 1) continue;
        const double *R0 = histR, *t0 = histT, *Rp = histR + 9 * (size_t)jp, *tp = histT + 3 * (size_t)jp;
        const double cx = histXY[k], cy = histXY[N + k], px = histXY[(size_t)jp * 2 * N + k], py = histXY[(size_t)jp * 2 * N + N + k];
        double M[3], cov[9];
        tri2(iK, Rp, tp, px, py, R0, t0, cx, cy, M);                    /* :950 */
        if (behind(R0, t0, M)) continue;                                /* :953 */
        cov2(K, Rp, tp, R0, t0, M, sigma, cov);                         /* :957 */
        double org[3];
        cam_center(R0, t0, org);
        const double sTr = fabs((cov[0] + cov[4]) + cov[8]);
        const double dx = org[0] - M[0], dy = org[1] - M[1], dz = org[2] - M[2];
        if (sqrt((dx * dx + dy * dy) + dz * dz) < sqrt(sTr)) continue;  /* :960-962 */
        if (!(reproj_err(K, Rp, tp, M, px, py) < maxEpiErr && reproj_err(K, R0, t0, M, cx, cy) < maxEpiErr)) continue; /* :965-970 */
        /* refineTriangulation(cur_fp, M, cov): the current view and the widest-parallax one behind it */
        int best = -1;
        double bestCos = 1.0, bestAngle = 0.0;
        for (int j = 1; j <= jp; j++) {
            double Cj[3];
            cam_center(histR + 9 * (size_t)j, histT + 3 * (size_t)j, Cj);
            const double cv = cos_between(M, org, Cj);
            if (cmpAcos) {
                const double ang = fabs(acos(cv));
                if (ang > bestAngle) bestAngle = ang, best = j;
            } else if (cv < bestCos)
                bestCos = cv, best = j;
        }
        if (best >= 0) {
            const double *Rb = histR + 9 * (size_t)best, *tb = histT + 3 * (size_t)best;
            tri2(iK, R0, t0, cx, cy, Rb, tb, histXY[(size_t)best * 2 * N + k], histXY[(size_t)best * 2 * N + N + k], M);
            cov2(K, R0, t0, Rb, tb, M, sigma, cov);
        }
        const double e1 = reproj_err(K, Rp, tp, M, px, py), e2 = reproj_err(K, R0, t0, M, cx, cy);
        if (behind(R0, t0, M) || behind(Rp, tp, M)) continue;           /* :977-979 */
        if (!(e1 < maxEpiErr && e2 < maxEpiErr)) continue;              /* :980 */
        newSlot[n] = k, newFirst[n] = f2 - jp;
        memcpy(newM + 3 * (size_t)n, M, 24), memcpy(newCov + 9 * (size_t)n, cov, 72);
        n++;
    }
    return n;
}

/* ---- CoSLAM::mapPointsClassify (/root/reference/src/app/SL_CoSLAM.cpp:418-520) --------------------------------------------------
 * Every frame, behind the pose update (CoSLAM::poseUpdate, :381-385: mapStateUpdate(), then mapPointsClassify(12.0)), every map point
 * of the current list that is uncertain (what the gate of poseUpdate3D made of it) or locally dynamic is re-examined:
 *   seen by one camera only                            -> false (:434-438)
 *   uncertain, new (bNewPt)   isStaticPoint over the last 60 frames?  yes: older than 30 frames -> static at the new position
 *                             (else it stays uncertain);  no: isDynamicPoint?  yes -> dynamic at the new position, no -> false
 *   uncertain, not new        isDynamicPoint? yes -> dynamic; no: isStaticRemovable (drop the view with the largest error, static
 *                             from the rest)?  yes -> that feature is detached, static at the new position;  no -> false
 *   dynamic                   isDynamicPoint? no -> false;  yes: moved little (isLittleMove)?  count the frames; beyond 50 and
 *                             isStaticPoint -> static again (its features' types too) -- but the position written last is the
 *                             dynamic one in every branch (:512 follows the if / else)
 * with the helpers of src/slam/SL_CoSLAMHelper.cpp: isStaticPoint (:117-181), isStaticPointExclude (:183-250), isDynamicPoint
 * (:251-312), isLittleMove (:314-330), isStaticRemovable (:67-115).  A point's features are MapPoint::pFeatures[iCam]: the table
 * pointFeat (slot per camera, < 0 none) with, optionally, featFrame (the feature's frame; NULL: all of this frame) and featFirst (the
 * first frame of its track; NULL: trackSpan of the slot) -- a camera that lost the point keeps its last feature in the reference, and
 * isStaticPoint / isLittleMove / isStaticRemovable still use it while it is young enough.  Pixels and poses of a feature's frame and
 * of the frames before it come from the history (entry = curFrame - frame; a feature older than the history is treated as absent).
 * PARITY: the state machine and the helpers' loops are pinned against the reference's own SL_CoSLAM.cpp + SL_CoSLAMHelper.cpp
 * compiled in place (tests/cxx/ref_classify_test.cpp -> tests/golden/classify_golden.npz); UNPINNED are the LibVisualSLAM helpers
 * (as above, plus isAtCameraBack(R, t, M) = (R M + t).z < 0 and dist3 = Euclidean distance). */
typedef struct {
    int nCams, N, nHist, curFrame;
    const double *Ks, *iKs, *histR, *histT, *histXY;
    const int* trackSpan;
    const int *featFrame, *featFirst; /* [nMap][nCams] or NULL */
    const int* featRef;               /* [nMap][nCams][4] or NULL: the features as references (then featFrame / featFirst are not read) */
    const int* segPool;               /* [nCams][segCap][4] or NULL */
    int segCap;
} opu_cls_ctx;

typedef struct {
    int c, j, s; /* camera, history entry, slot */
} opu_view;

/* the feature of point m in camera c: slot, history entry of its frame, that frame, the first frame of the run of consecutive frames behind
 * it, the first linked segment (-1 none); 0 if there is none (or it is older than the history).
 * With references: the table is MapPoint::pFeatures as the END of the previous frame left it (cs_feat_ref_advance_dev) and pointFeat names
 * this frame's features (the hand-back has moved a live pointer along its track, SL_SingleSLAM.cpp:34-60): a feature of this frame is the
 * reference moved on by one frame (a table that is already at this frame is taken as it is); no feature of this frame and an older
 * reference: the camera lost the point, the stale feature stands. */
static int cls_feature_seg(const opu_cls_ctx* X, const int* pf, int m, int c, int* slot, int* j0, int* frame, int* first, int* seg) {
    const int s = pf[(size_t)m * X->nCams + c];
    int f, ff, sg = -1, sl = s;
    if (X->featRef) {
        const int* r = X->featRef + ((size_t)m * X->nCams + c) * 4;
        if (s >= 0) {
            f = X->curFrame;
            if (r[0] == s && (r[1] == f || r[1] == f - 1)) ff = r[2], sg = r[3];
            else ff = X->trackSpan[(size_t)c * 2 * X->N + s];
        } else if (r[0] >= 0 && r[1] < X->curFrame)
            sl = r[0], f = r[1], ff = r[2], sg = r[3];
        else
            return 0;
    } else {
        if (s < 0) return 0;
        f = X->featFrame ? X->featFrame[(size_t)m * X->nCams + c] : X->curFrame;
        ff = X->featFirst ? X->featFirst[(size_t)m * X->nCams + c] : X->trackSpan[(size_t)c * 2 * X->N + s];
    }
    const int j = X->curFrame - f;
    if (j < 0 || j >= X->nHist) return 0;
    *slot = sl, *j0 = j, *frame = f, *first = ff, *seg = sg;
    return 1;
}
static int cls_feature(const opu_cls_ctx* X, const int* pf, int m, int c, int* slot, int* j0, int* frame, int* first) {
    int seg;
    return cls_feature_seg(X, pf, m, c, slot, j0, frame, first, &seg);
}
static const double* cls_R(const opu_cls_ctx* X, int c, int j) { return X->histR + ((size_t)c * X->nHist + j) * 9; }
static const double* cls_t(const opu_cls_ctx* X, int c, int j) { return X->histT + ((size_t)c * X->nHist + j) * 3; }
static void cls_pixel(const opu_cls_ctx* X, int c, int j, int s, double* mx, double* my) {
    const double* h = X->histXY + ((size_t)c * X->nHist + j) * 2 * X->N;
    *mx = h[s], *my = h[X->N + s];
}
/* triangulateMultiView + getTriangulateCovMat over a view list, then the reprojection gate of every view (> 1.0 fails) */
static int cls_triangulate_and_gate(const opu_cls_ctx* X, const opu_view* v, int nv, double sigma, double* M, double* cov, int gate) {
    opu_normal_eq E;
    memset(&E, 0, sizeof(E));
    for (int i = 0; i < nv; i++) {
        double mx, my;
        cls_pixel(X, v[i].c, v[i].j, v[i].s, &mx, &my);
        ne_add_view(&E, X->iKs + 9 * v[i].c, cls_R(X, v[i].c, v[i].j), cls_t(X, v[i].c, v[i].j), mx, my);
    }
    double cf[6];
    const double det = sym33_cof(E.N, cf);
    M[0] = ((cf[0] * E.g[0] + cf[1] * E.g[1]) + cf[2] * E.g[2]) / det;
    M[1] = ((cf[1] * E.g[0] + cf[3] * E.g[1]) + cf[4] * E.g[2]) / det;
    M[2] = ((cf[2] * E.g[0] + cf[4] * E.g[1]) + cf[5] * E.g[2]) / det;
    if (gate == 2) return 1; /* isDynamicPoint looks at the point before it asks for the covariance */
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nv; i++) cov_add_view(S, X->Ks + 9 * v[i].c, cls_R(X, v[i].c, v[i].j), cls_t(X, v[i].c, v[i].j), M);
    const double dS = sym33_cof(S, cf), s2 = sigma * sigma;
    cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
    cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
    cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
    if (!gate) return 1;
    for (int i = 0; i < nv; i++) {
        double rm[2], var[4], ivar[4], mx, my;
        cls_pixel(X, v[i].c, v[i].j, v[i].s, &mx, &my);
        org_project(X->Ks + 9 * v[i].c, cls_R(X, v[i].c, v[i].j), cls_t(X, v[i].c, v[i].j), M, rm);
        org_projection_cov(X->Ks + 9 * v[i].c, cls_R(X, v[i].c, v[i].j), cls_t(X, v[i].c, v[i].j), M, cov, var, sigma);
        mat22_inv(var, ivar);
        if (maha_dist2(rm, mx, my, ivar) > 1.0) return 0;
    }
    return 1;
}
/* isStaticPoint (exclude < 0) / isStaticPointExclude: the views of updateStaticPointPosition inside the window of numFrame frames.  The
 * backward walk `fp = fp->preFrame while fp && fp->f >= firstFrame` (:141-152) runs along the feature's own run of frames and then
 * through the linked segments (each older than the one before it), and ends at the first node before the window. */
static int cls_is_static(const opu_cls_ctx* X, const int* pf, int m, const double* Mold, double sigma, double* M, double* cov, int exclude,
                         int numFrame) {
    opu_view v[64];
    int nv = 0;
    const int firstFrame = X->curFrame - numFrame; /* p->lastFrame - numFrame; lastFrame == curFrame on the current list */
    for (int c = 0; c < X->nCams; c++) {
        int s, j0, f, ff, seg;
        if (c == exclude || !cls_feature_seg(X, pf, m, c, &s, &j0, &f, &ff, &seg) || f < firstFrame) continue;
        v[nv].c = c, v[nv].j = j0, v[nv].s = s, nv++;
        double C0[3];
        cam_center(cls_R(X, c, j0), cls_t(X, c, j0), C0);
        int best = -1, bestSlot = -1, slot = s, hi = f - 1, lo = ff, ended = 0;
        double bestCos = 1.0;
        for (;;) {
            for (int fr = hi; fr >= lo; fr--) {
                const int j = X->curFrame - fr;
                if (fr < firstFrame || j >= X->nHist) {
                    ended = 1;
                    break;
                }
                double Cj[3];
                cam_center(cls_R(X, c, j), cls_t(X, c, j), Cj);
                const double cv = cos_between(Mold, C0, Cj);
                if (cv < bestCos) bestCos = cv, best = j, bestSlot = slot;
            }
            if (ended || seg < 0 || !X->segPool || seg >= X->segCap) break;
            const int* g = X->segPool + ((size_t)c * X->segCap + seg) * 4;
            slot = g[0], hi = g[1], lo = g[2], seg = g[3];
        }
        if (best >= 0) v[nv].c = c, v[nv].j = best, v[nv].s = bestSlot, nv++;
    }
    return cls_triangulate_and_gate(X, v, nv, sigma, M, cov, 1);
}
/* isDynamicPoint: this frame's features only */
static int cls_is_dynamic(const opu_cls_ctx* X, const int* pf, const unsigned char* featStatic, int m, double sigma, double* M, double* cov) {
    opu_view v[32];
    int nv = 0;
    for (int c = 0; c < X->nCams; c++) {
        int s, j0, f, ff;
        if (!cls_feature(X, pf, m, c, &s, &j0, &f, &ff) || f != X->curFrame) continue;
        v[nv].c = c, v[nv].j = 0, v[nv].s = s, nv++;
    }
    (void)featStatic; /* (:269-270 count the dynamic features and never use the count) */
    if (nv < 2) return 0;
    double org[3];
    cam_center(cls_R(X, v[0].c, 0), cls_t(X, v[0].c, 0), org);
    cls_triangulate_and_gate(X, v, nv, sigma, M, cov, 2);
    {
        const double* R = cls_R(X, v[0].c, 0);
        const double* t = cls_t(X, v[0].c, 0);
        if (((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2] < 0) return 0; /* isAtCameraBack */
    }
    /* (the covariance and the gate: the same sums as above, recomputed -- the reference calls the helpers in this order) */
    double S[6] = {0, 0, 0, 0, 0, 0}, cf[6];
    for (int i = 0; i < nv; i++) cov_add_view(S, X->Ks + 9 * v[i].c, cls_R(X, v[i].c, 0), cls_t(X, v[i].c, 0), M);
    const double dS = sym33_cof(S, cf), s2 = sigma * sigma;
    cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
    cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
    cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
    const double sc = (fabs(cov[0]) + fabs(cov[4])) + fabs(cov[8]);
    const double dx = M[0] - org[0], dy = M[1] - org[1], dz = M[2] - org[2];
    if (sqrt((dx * dx + dy * dy) + dz * dz) * 0.2 < sqrt(sc)) return 0; /* :290-293 */
    for (int i = 0; i < nv; i++) {
        double rm[2], var[4], ivar[4], mx, my;
        cls_pixel(X, v[i].c, 0, v[i].s, &mx, &my);
        org_project(X->Ks + 9 * v[i].c, cls_R(X, v[i].c, 0), cls_t(X, v[i].c, 0), M, rm);
        org_projection_cov(X->Ks + 9 * v[i].c, cls_R(X, v[i].c, 0), cls_t(X, v[i].c, 0), M, cov, var, sigma);
        mat22_inv(var, ivar);
        if (maha_dist2(rm, mx, my, ivar) > 1.0) return 0;
    }
    return 1;
}
/* the Mahalanobis distance of a feature (any frame of the history) from the projection of (M, cov) under its own frame's pose */
static double cls_feature_err(const opu_cls_ctx* X, int c, int j, int s, const double* M, const double* cov, double sigma) {
    double rm[2], var[4], ivar[4], mx, my;
    cls_pixel(X, c, j, s, &mx, &my);
    org_project(X->Ks + 9 * c, cls_R(X, c, j), cls_t(X, c, j), M, rm);
    org_projection_cov(X->Ks + 9 * c, cls_R(X, c, j), cls_t(X, c, j), M, cov, var, sigma);
    mat22_inv(var, ivar);
    return maha_dist2(rm, mx, my, ivar);
}

/* One frame's mapPointsClassify over the points that have a feature in this frame.  In / out per point: mapPts, mapCov, mapFlags
 * (bit 0 dynamic, bit 1 false, bit 2 uncertain), newPt (MapPoint::bNewPt), staticFrameNum; in: firstFrame (MapPoint::firstFrame).
 * In / out tables: pointFeat (a detached feature becomes -1), slot2map [nCams][N] (or NULL; the detached feature's slot becomes
 * -1), featStatic [nCams][N] (feature types; a point that returns to static sets its features' types to static).  Returns the number
 * of points examined; *numFalse = those that became false. */
static int classify_core(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                         const double* histXY, const int* trackSpan, unsigned char* featStatic, int* slot2map, int nMap, int* pointFeat,
                         const int* featFrame, const int* featFirst, int* featRef, const int* segPool, int segCap, unsigned char* refStatic,
                         int curFrame, double* mapPts, double* mapCov, unsigned char* mapFlags, unsigned char* newPt, int* staticFrameNum,
                         const int* firstFrame, double pixelVar, int* numFalse) {
    const int FRAME_NUM_FOR_NEWPOINT = 30, FRAME_NUM_FOR_DONTMOVE = 50, NUM_FRAME_CHECK_STATIC = 60;
    opu_cls_ctx X = {nCams, N, nHist, curFrame, Ks, iKs, histR, histT, histXY, trackSpan, featFrame, featFirst, featRef, segPool, segCap};
    int nExamined = 0, nFalse = 0;
    for (int m = 0; m < nMap; m++) {
        /* the current list after mapStateUpdate (:1183-1197): points with a feature in this frame; numVisCam counts those features */
        int numVisCam = 0;
        for (int c = 0; c < nCams; c++) {
            int s, j0, f, ff;
            if (cls_feature(&X, pointFeat, m, c, &s, &j0, &f, &ff) && f == curFrame) numVisCam++;
        }
        if (numVisCam == 0) continue;
        unsigned char fl = mapFlags[m];
        const int uncertain = (fl & OPU_UNCERTAIN) != 0, locDyn = (fl & (OPU_DYNAMIC | OPU_FALSE)) == OPU_DYNAMIC;
        if (!(uncertain || locDyn)) continue; /* :431 */
        nExamined++;
        double* pM = mapPts + 3 * (size_t)m;
        double* pCov = mapCov + 9 * (size_t)m;
#define SET_FALSE() (fl = (unsigned char)((fl & ~OPU_DYNAMIC) | OPU_FALSE))
#define SET_DYNAMIC() (fl = OPU_DYNAMIC, staticFrameNum[m] = 0)
#define SET_STATIC() (fl = 0, staticFrameNum[m] = 0)
#define UPDATE_POS(Mn, Cn) (memcpy(pM, Mn, 24), memcpy(pCov, Cn, 72))
        if (numVisCam == 1) { /* :433-437 */
            SET_FALSE();
            if (!(mapFlags[m] & OPU_FALSE)) nFalse++;
            mapFlags[m] = fl;
            continue;
        }
        double M[3], cov[9];
        if (uncertain) {
            if (newPt[m]) {
                if (cls_is_static(&X, pointFeat, m, pM, pixelVar, M, cov, -1, NUM_FRAME_CHECK_STATIC)) {
                    if (curFrame - firstFrame[m] > FRAME_NUM_FOR_NEWPOINT) { /* p->lastFrame - p->firstFrame */
                        SET_STATIC();
                        newPt[m] = 0;
                        UPDATE_POS(M, cov);
                    }
                } else if (cls_is_dynamic(&X, pointFeat, featStatic, m, pixelVar, M, cov)) {
                    SET_DYNAMIC();
                    UPDATE_POS(M, cov);
                    newPt[m] = 0;
                } else
                    SET_FALSE();
            } else {
                if (cls_is_dynamic(&X, pointFeat, featStatic, m, pixelVar, M, cov)) {
                    SET_DYNAMIC();
                    UPDATE_POS(M, cov);
                } else {
                    /* isStaticRemovable: the view with the largest error (> 1) under the point as it stands; static without it? */
                    int maxI = -1, nVis = 0;
                    double maxErr = 1.0;
                    for (int c = 0; c < nCams; c++) {
                        int s, j0, f, ff;
                        if (!cls_feature(&X, pointFeat, m, c, &s, &j0, &f, &ff)) continue;
                        const double err = cls_feature_err(&X, c, j0, s, pM, pCov, pixelVar);
                        if (err > maxErr) maxErr = err, maxI = c;
                        nVis++;
                    }
                    int out = -1;
                    if (maxI >= 0 && nVis > 2 && cls_is_static(&X, pointFeat, m, pM, pixelVar, M, cov, maxI, NUM_FRAME_CHECK_STATIC)) out = maxI;
                    if (out >= 0) { /* :476-482 */
                        const int s = pointFeat[(size_t)m * nCams + out]; /* (< 0 with references: the view that goes is a stale one) */
                        if (slot2map && s >= 0) slot2map[(size_t)out * N + s] = -1;
                        pointFeat[(size_t)m * nCams + out] = -1;
                        if (featRef) { /* p->pFeatures[outlierViewId] = 0: the chain behind it goes with it */
                            int* r = featRef + ((size_t)m * nCams + out) * 4;
                            r[0] = -1, r[1] = 0, r[2] = 0, r[3] = -1;
                        }
                        SET_STATIC();
                        UPDATE_POS(M, cov);
                    } else
                        SET_FALSE();
                }
            }
        } else { /* locally dynamic (:489-516) */
            if (cls_is_dynamic(&X, pointFeat, featStatic, m, pixelVar, M, cov)) {
                int little = 1;
                for (int c = 0; c < nCams && little; c++) { /* isLittleMove: >= 1 fails */
                    int s, j0, f, ff;
                    if (!cls_feature(&X, pointFeat, m, c, &s, &j0, &f, &ff)) continue;
                    if (cls_feature_err(&X, c, j0, s, M, cov, pixelVar) >= 1) little = 0;
                }
                if (little) {
                    staticFrameNum[m]++;
                    if (staticFrameNum[m] > FRAME_NUM_FOR_DONTMOVE) {
                        double M0[3], cov0[9];
                        if (cls_is_static(&X, pointFeat, m, pM, pixelVar, M0, cov0, -1, NUM_FRAME_CHECK_STATIC)) {
                            SET_STATIC();
                            for (int c = 0; c < nCams; c++) {
                                int s, j0, f, ff;
                                if (!cls_feature(&X, pointFeat, m, c, &s, &j0, &f, &ff)) continue;
                                if (f == curFrame) featStatic[(size_t)c * N + s] = 1;
                                else if (refStatic) refStatic[(size_t)m * nCams + c] = 1; /* (:494-498 set the type of every feature held) */
                            }
                        } else
                            staticFrameNum[m] = 0;
                    }
                } else
                    staticFrameNum[m] = 0;
                UPDATE_POS(M, cov); /* :512: the dynamic triangulation is what stays, also for a point that went back to static */
            } else
                SET_FALSE();
        }
        if ((fl & OPU_FALSE) && !(mapFlags[m] & OPU_FALSE)) nFalse++;
        mapFlags[m] = fl;
#undef SET_FALSE
#undef SET_DYNAMIC
#undef SET_STATIC
#undef UPDATE_POS
    }
    if (numFalse) *numFalse = nFalse;
    return nExamined;
}

int opu_map_points_classify(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                            const double* histXY, const int* trackSpan, unsigned char* featStatic, int* slot2map, int nMap, int* pointFeat,
                            const int* featFrame, const int* featFirst, int curFrame, double* mapPts, double* mapCov,
                            unsigned char* mapFlags, unsigned char* newPt, int* staticFrameNum, const int* firstFrame, double pixelVar,
                            int* numFalse) {
    return classify_core(nCams, N, nHist, Ks, iKs, histR, histT, histXY, trackSpan, featStatic, slot2map, nMap, pointFeat, featFrame, featFirst,
                         NULL, NULL, 0, NULL, curFrame, mapPts, mapCov, mapFlags, newPt, staticFrameNum, firstFrame, pixelVar, numFalse);
}
/* ... with the points' features as references (featRef [nMap][nCams][4] in / out, segPool [nCams][segCap][4], refStatic [nMap][nCams] in /
 * out or NULL: cls_feature_seg above): stale features are views of isStaticPoint (inside its window), isLittleMove and isStaticRemovable,
 * isStaticPoint's backward walks follow the linked segments, the view isStaticRemovable drops may be a stale one (its reference is
 * cleared), a point that returns to static sets the type of its stale features too.
 * PARITY: tests/cxx/ref_classify_test.cpp golden_relink -> tests/golden/classify_relink_golden.npz (the reference's own functions over
 * chains built with its classes). */
int opu_map_points_classify_ref(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                                const double* histXY, const int* trackSpan, unsigned char* featStatic, int* slot2map, int nMap, int* pointFeat,
                                int* featRef, const int* segPool, int segCap, unsigned char* refStatic, int curFrame, double* mapPts,
                                double* mapCov, unsigned char* mapFlags, unsigned char* newPt, int* staticFrameNum, const int* firstFrame,
                                double pixelVar, int* numFalse) {
    return classify_core(nCams, N, nHist, Ks, iKs, histR, histT, histXY, trackSpan, featStatic, slot2map, nMap, pointFeat, NULL, NULL, featRef,
                         segPool, segCap, refStatic, curFrame, mapPts, mapCov, mapFlags, newPt, staticFrameNum, firstFrame, pixelVar, numFalse);
}

/* ---- CoSLAM::checkUnify (/root/reference/src/app/SL_CoSLAM.cpp:561-665) ---------------------------------------------------------
 * What the registration loops ask when a map point's candidate feature already belongs to another map point (:795-829, every 50th
 * frame and after a group merge): can the two points be one?  The views of BOTH points -- per camera first point 1's feature and
 * its widest-parallax predecessor (angles at point 1's position), then point 2's likewise (angles at point 2's position) -- are
 * triangulated together (triangulateMultiView, getTriangulateCovMat) and every view must lie within Mahalanobis distance 1 of the
 * new point.  NOTE :657: the gate's getProjectionCovMat is handed `Rs + 3 * i` where `Rs + 9 * i` is meant, i.e. for view i >= 1 the
 * nine doubles starting at offset 3 i of the concatenated rotations -- rows of two different views' matrices; the projection itself
 * (:656) uses the right one.  Restated as written.  Features of this frame only (a pair's tables name the slot per camera, < 0 none);
 * the walk runs over the whole track (no window), bounded by the history.  Returns 1 / 0; M and cov are written in either case.
 * Pinned against the reference's own function (tests/cxx/ref_update_points_test.cpp: halves of one point's cameras, and different
 * points); the device counterpart is cs_check_unify_dev (coslam_amd/csrc/poseupdate.hip: k_check_unify). */
static int check_unify_core(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                            const double* histXY, const int* trackSpan, const int* pf1, const int* pf2, const int* ref1, const int* ref2,
                            const int* segPool, int segCap, int curFrame, int walkCap, const double* M1, const double* M2, double sigma,
                            int cmpAcos, double* M, double* cov) {
    const int byRef = ref1 != 0;
    const opu_chain_ctx X = {byRef ? curFrame : 0, nHist, byRef ? walkCap : nHist, segCap, N, cmpAcos, segPool};
    opu_view v[128];
    int slotv[128], nv = 0;
    for (int c = 0; c < nCams; c++) {
        for (int which = 0; which < 2; which++) {
            int ref[4];
            /* (a point's row of references / of slots: chain_ref's m = 0 row of a one-point table) */
            chain_ref(byRef ? (which ? ref2 : ref1) : 0, which ? pf2 : pf1, trackSpan, nCams, N, X.curFrame, 0, c, ref);
            const int s = ref[0], j0 = X.curFrame - ref[1];
            if (s < 0 || j0 >= nHist) continue;
            const double* Mold = which ? M2 : M1;
            const double* hR = histR + (size_t)c * nHist * 9;
            const double* hT = histT + (size_t)c * nHist * 3;
            v[nv].c = c, v[nv].j = j0, slotv[nv] = s, nv++;
            double C0[3];
            cam_center(hR + 9 * (size_t)j0, hT + 3 * (size_t)j0, C0);
            int best = -1, bs = s;
            if (chain_widest(&X, c, ref, hR, hT, Mold, C0, &bs, &best)) v[nv].c = c, v[nv].j = best, slotv[nv] = bs, nv++;
        }
    }
    opu_normal_eq E;
    memset(&E, 0, sizeof(E));
    double RsFlat[128 * 9 + 9];
    for (int i = 0; i < nv; i++) {
        const double* R = histR + ((size_t)v[i].c * nHist + v[i].j) * 9;
        const double* t = histT + ((size_t)v[i].c * nHist + v[i].j) * 3;
        const double* h = histXY + ((size_t)v[i].c * nHist + v[i].j) * 2 * N;
        memcpy(RsFlat + 9 * i, R, 72);
        ne_add_view(&E, iKs + 9 * v[i].c, R, t, h[slotv[i]], h[N + slotv[i]]);
    }
    double cf[6];
    const double det = sym33_cof(E.N, cf);
    M[0] = ((cf[0] * E.g[0] + cf[1] * E.g[1]) + cf[2] * E.g[2]) / det;
    M[1] = ((cf[1] * E.g[0] + cf[3] * E.g[1]) + cf[4] * E.g[2]) / det;
    M[2] = ((cf[2] * E.g[0] + cf[4] * E.g[1]) + cf[5] * E.g[2]) / det;
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nv; i++)
        cov_add_view(S, Ks + 9 * v[i].c, histR + ((size_t)v[i].c * nHist + v[i].j) * 9, histT + ((size_t)v[i].c * nHist + v[i].j) * 3, M);
    const double dS = sym33_cof(S, cf), s2 = sigma * sigma;
    cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
    cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
    cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
    for (int i = 0; i < nv; i++) {
        const double* R = histR + ((size_t)v[i].c * nHist + v[i].j) * 9;
        const double* t = histT + ((size_t)v[i].c * nHist + v[i].j) * 3;
        const double* h = histXY + ((size_t)v[i].c * nHist + v[i].j) * 2 * N;
        double rm[2], var[4], ivar[4];
        org_project(Ks + 9 * v[i].c, R, t, M, rm);                                  /* :656 */
        org_projection_cov(Ks + 9 * v[i].c, RsFlat + 3 * i, t, M, cov, var, sigma); /* :657: Rs + 3 * i */
        mat22_inv(var, ivar);
        if (maha_dist2(rm, h[slotv[i]], h[N + slotv[i]], ivar) > 1.0) return 0;
    }
    return 1;
}
int opu_check_unify(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                    const double* histXY, const int* trackSpan, const int* pf1, const int* pf2, const double* M1, const double* M2,
                    double sigma, int cmpAcos, double* M, double* cov) {
    return check_unify_core(nCams, N, nHist, Ks, iKs, histR, histT, histXY, trackSpan, pf1, pf2, 0, 0, 0, 0, 0, 0, M1, M2, sigma, cmpAcos, M, cov);
}
/* the same with the two points' features as references (ref1 / ref2 [nCams][4], segPool: above update_points_core) */
int opu_check_unify_ref(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                        const double* histXY, const int* ref1, const int* ref2, const int* segPool, int segCap, int curFrame, int walkCap,
                        const double* M1, const double* M2, double sigma, int cmpAcos, double* M, double* cov) {
    return check_unify_core(nCams, N, nHist, Ks, iKs, histR, histT, histXY, 0, 0, 0, ref1, ref2, segPool, segCap, curFrame, walkCap, M1, M2, sigma,
                            cmpAcos, M, cov);
}
