/*
 * oracle/posegraph_oracle.c -- CPU restatement of the pose-graph relaxation that moves the non-key frames after a bundle
 * adjustment.
 *
 * TEST INFRASTRUCTURE ONLY (see klt_oracle.h).  Follows
 *   GlobalPoseGraph::computeNewCameraRotations      /root/reference/src/slam/SL_GlobalPoseEstimation.cpp:52-219
 *   GlobalPoseGraph::computeNewCameraTranslations   /root/reference/src/slam/SL_GlobalPoseEstimation.cpp:220-359
 * as RobustBundleRTS::updateNonKeyCameraPoses (src/app/SL_CoSLAMRobustBA.cpp:230-247) calls them, one graph at a time:
 * every edge with at least one free end gives 9 (rotation entries) / 3 (translation) linear equations
 *       R_j = R_ij R_i          T_j - R_ij T_i = T_ij
 * in the unknown poses of the free nodes, fixed nodes moved to the right-hand side; the over-determined system is solved in
 * the least-squares sense; the 3x3 blocks of the rotation solution are projected back to rotations.  The dense 9-unknowns-
 * per-node system is built exactly like the reference's triplet list (same rows, columns, signs, right-hand sides), so the
 * restatement is pinned by tests/cxx/ref_posegraph_test.cpp, which runs the reference's own source compiled in place
 * (oracle/_ref/ref_posegraph_test, goldens in tests/golden/posegraph_golden.npz).
 *
 * Edges with uncertainScale (extra scale unknowns, only created by the camera-group merge, src/app/SL_MergeCameraGroup.cpp:
 * 972-1025, out of SURVEY 8's scope) are not restated.
 *
 * PARITY UNPINNED for the external LibVisualSLAM helpers (absent; only their calls are in the reference):
 *   sparseSolveLin(T, b, x)       least-squares solution of T x = b (dense Householder QR here)
 *   approxRotationMat(R, Rnew)    U V^T of the SVD of R (one-sided Jacobi here)
 *   getRigidTransFromTo           R = R2 R1^T, t = t2 - R t1 (src/app/SL_CoSLAMRobustBA.cpp:225)
 *   mat33AB / mat33Trans / mat33ProdVec
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "klt_oracle.h"

void opg_rigid_from_to(const double R1[9], const double t1[3], const double R2[9], const double t2[3], double R[9], double t[3]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = (R2[3 * r] * R1[3 * c] + R2[3 * r + 1] * R1[3 * c + 1]) + R2[3 * r + 2] * R1[3 * c + 2];
    for (int r = 0; r < 3; ++r) t[r] = t2[r] - ((R[3 * r] * t1[0] + R[3 * r + 1] * t1[1]) + R[3 * r + 2] * t1[2]);
}

/* min |A x - b| by Householder QR; A is m x n row-major (destroyed), m >= n, full column rank */
static int ls_solve(int m, int n, double* A, double* b, double* x) {
    double* v = (double*)malloc(sizeof(double) * (size_t)m);
    for (int j = 0; j < n; ++j) {
        double nrm = 0;
        for (int i = j; i < m; ++i) nrm += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        nrm = sqrt(nrm);
        if (nrm == 0) {
            free(v);
            return -1;
        }
        const double alpha = A[(size_t)j * n + j] > 0 ? -nrm : nrm;
        for (int i = j; i < m; ++i) v[i] = A[(size_t)i * n + j];
        v[j] -= alpha;
        double vtv = 0;
        for (int i = j; i < m; ++i) vtv += v[i] * v[i];
        for (int c = j; c < n; ++c) {
            double s = 0;
            for (int i = j; i < m; ++i) s += v[i] * A[(size_t)i * n + c];
            s = 2 * s / vtv;
            if (s != 0)
                for (int i = j; i < m; ++i) A[(size_t)i * n + c] -= s * v[i];
        }
        double s = 0;
        for (int i = j; i < m; ++i) s += v[i] * b[i];
        s = 2 * s / vtv;
        for (int i = j; i < m; ++i) b[i] -= s * v[i];
    }
    free(v);
    for (int j = n - 1; j >= 0; --j) {
        double s = b[j];
        for (int c = j + 1; c < n; ++c) s -= A[(size_t)j * n + c] * x[c];
        x[j] = s / A[(size_t)j * n + j];
    }
    return 0;
}

/* U V^T of R = U S V^T (one-sided Jacobi on the columns) */
void opg_approx_rotation(const double R[9], double Rnew[9]) {
    double W[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(W, R, sizeof(W));
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, bq = 0, g = 0;
                for (int i = 0; i < 3; ++i) {
                    a += W[3 * i + p] * W[3 * i + p];
                    bq += W[3 * i + q] * W[3 * i + q];
                    g += W[3 * i + p] * W[3 * i + q];
                }
                if (fabs(g) <= 1e-300 || fabs(g) <= 1e-17 * sqrt(a * bq)) continue;
                off = fmax(off, fabs(g) / sqrt(a * bq));
                const double zeta = (bq - a) / (2 * g), tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                const double c = 1 / sqrt(1 + tt * tt), s = c * tt;
                for (int i = 0; i < 3; ++i) {
                    const double wp = W[3 * i + p], wq = W[3 * i + q], vp = V[3 * i + p], vq = V[3 * i + q];
                    W[3 * i + p] = c * wp - s * wq, W[3 * i + q] = s * wp + c * wq;
                    V[3 * i + p] = c * vp - s * vq, V[3 * i + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-16) break;
    }
    double U[9];
    for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int i = 0; i < 3; ++i) s += W[3 * i + c] * W[3 * i + c];
        s = sqrt(s);
        for (int i = 0; i < 3; ++i) U[3 * i + c] = s > 0 ? W[3 * i + c] / s : (i == c ? 1.0 : 0.0);
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rnew[3 * r + c] = (U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1]) + U[3 * r + 2] * V[3 * c + 2];
}

/* one graph: node poses (R, t) with fixed flags, edges (id1 -> id2, R, t); newR / newT for every node.  Returns 0, or -1 when
 * the system is rank deficient (a free node no edge reaches). */
int opg_relax(int nNodes, int nEdges, const unsigned char* fixed, const double* nodeR, const double* nodeT, const int* id1,
              const int* id2, const double* edgeR, const double* edgeT, double* newR, double* newT) {
    int* indR = (int*)malloc(sizeof(int) * (size_t)(nEdges > 0 ? nEdges : 1));
    int* indC = (int*)malloc(sizeof(int) * (size_t)(nNodes > 0 ? nNodes : 1));
    int nValidEdge = 0, nValidNode = 0;
    for (int k = 0; k < nEdges; ++k) indR[k] = (!fixed[id1[k]] || !fixed[id2[k]]) ? nValidEdge++ : -1; /* :56-61 */
    for (int k = 0; k < nNodes; ++k) indC[k] = !fixed[k] ? nValidNode++ : -1;                           /* :66-70 */
    memcpy(newR, nodeR, sizeof(double) * 9 * (size_t)nNodes);                                            /* :213-214 */
    memcpy(newT, nodeT, sizeof(double) * 3 * (size_t)nNodes);
    int rc = 0;
    if (nValidNode > 0) {
        /* rotations (:72-197): unknown block c holds R_c^T row-major, i.e. 3 consecutive unknowns = one COLUMN of R_c */
        int m = 9 * nValidEdge, n = 9 * nValidNode;
        double* A = (double*)calloc((size_t)m * n, sizeof(double));
        double* b = (double*)calloc((size_t)m, sizeof(double));
        double* x = (double*)calloc((size_t)n, sizeof(double));
        for (int e = 0; e < nEdges; ++e) {
            if (indR[e] < 0) continue;
            const int r = indR[e], c1 = indC[id1[e]], c2 = indC[id2[e]];
            const double* R = edgeR + 9 * e;
            if (c2 >= 0)
                for (int q = 0; q < 9; ++q) A[(size_t)(9 * r + q) * n + 9 * c2 + q] += 1.0; /* :96-104 */
            if (c1 >= 0)
                for (int a = 0; a < 3; ++a)
                    for (int i = 0; i < 3; ++i)
                        for (int k = 0; k < 3; ++k) A[(size_t)(9 * r + 3 * a + i) * n + 9 * c1 + 3 * a + k] += -R[3 * i + k]; /* :106-134 */
            if (c1 < 0) { /* :148-150: b = (R_ij R_i)^T */
                const double* R1 = nodeR + 9 * id1[e];
                for (int i = 0; i < 3; ++i)
                    for (int a = 0; a < 3; ++a) b[9 * r + 3 * a + i] = (R[3 * i] * R1[a] + R[3 * i + 1] * R1[3 + a]) + R[3 * i + 2] * R1[6 + a];
            } else if (c2 < 0) { /* :183-193: b = -R_j^T */
                const double* R2 = nodeR + 9 * id2[e];
                for (int i = 0; i < 3; ++i)
                    for (int a = 0; a < 3; ++a) b[9 * r + 3 * a + i] = -R2[3 * i + a];
            }
        }
        if (m < n || ls_solve(m, n, A, b, x)) rc = -1;
        if (!rc)
            for (int k = 0; k < nNodes; ++k) { /* :205-212 */
                const int c = indC[k];
                if (c < 0) continue;
                double tmp[9];
                for (int i = 0; i < 3; ++i)
                    for (int a = 0; a < 3; ++a) tmp[3 * i + a] = x[9 * c + 3 * a + i];
                opg_approx_rotation(tmp, newR + 9 * k);
            }
        free(A), free(b), free(x);
        /* translations (:241-348), no uncertain-scale edges */
        m = 3 * nValidEdge, n = 3 * nValidNode;
        A = (double*)calloc((size_t)m * n, sizeof(double));
        b = (double*)calloc((size_t)m, sizeof(double));
        x = (double*)calloc((size_t)n, sizeof(double));
        for (int e = 0; e < nEdges; ++e) {
            if (indR[e] < 0) continue;
            const int r = indR[e], i = indC[id1[e]], j = indC[id2[e]];
            const double *R = edgeR + 9 * e, *t = edgeT + 3 * e;
            if (j >= 0)
                for (int q = 0; q < 3; ++q) A[(size_t)(3 * r + q) * n + 3 * j + q] += 1.0;
            if (i >= 0)
                for (int q = 0; q < 3; ++q)
                    for (int k = 0; k < 3; ++k) A[(size_t)(3 * r + q) * n + 3 * i + k] += -R[3 * q + k];
            if (i >= 0 && j >= 0) {
                memcpy(b + 3 * r, t, sizeof(double) * 3); /* :288 */
            } else if (j < 0) {
                const double* tj = nodeT + 3 * id2[e]; /* :313-315 */
                for (int q = 0; q < 3; ++q) b[3 * r + q] = t[q] - tj[q];
            } else {
                const double* ti = nodeT + 3 * id1[e]; /* :330: R t_i + t */
                for (int q = 0; q < 3; ++q) b[3 * r + q] = 1.0 * ((R[3 * q] * ti[0] + R[3 * q + 1] * ti[1]) + R[3 * q + 2] * ti[2]) + 1.0 * t[q];
            }
        }
        if (!rc && (m < n || ls_solve(m, n, A, b, x))) rc = -1;
        if (!rc)
            for (int k = 0; k < nNodes; ++k)
                if (indC[k] >= 0) memcpy(newT + 3 * k, x + 3 * indC[k], sizeof(double) * 3); /* :346 */
        free(A), free(b), free(x);
    }
    free(indR), free(indC);
    return rc;
}
