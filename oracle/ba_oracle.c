/*
 * oracle/ba_oracle.c -- CPU definition of bundleAdjustRobust for the hot path.
 * TEST INFRASTRUCTURE ONLY (see klt_oracle.h).
 *
 * PARITY UNPINNED.  bundleAdjustRobust lives in danping/LibVisualSLAM (geometry/SL_BundleAdjust.{h,cpp},
 * extern/sba-1.6), which is NOT vendored in the reference and has no pinned version (reference
 * CMakeLists.txt:7, cmake/Modules/FindVisualSLAM.cmake:1-18, README.md:36).  Only its call contract is
 * in the reference (src/app/SL_CoSLAMRobustBA.cpp:170-180,273-316; SL_InterCamPoseEstimator.cpp:92-95;
 * SL_MergeCameraGroup.cpp:646-647).  This file is OUR definition of an algorithm that honours that
 * contract, restated on the GPU by coslam_amd/csrc/ba.hip:
 *
 *   minimise  sum over inlier measurements |m_ij - pi(K_j (R_j M_i + t_j))|^2
 *   over cameras j >= nCamsCon (R_j <- R_j exp(w), t_j <- t_j + dt: the parametrisation of
 *   src/slam/SL_IntraCamPose.cpp:367-380) and points i >= nPtsCon (M_i <- M_i + dM), K fixed;
 *   outer loop (<= maxIter): Levenberg-Marquardt with the point blocks eliminated by Schur complement
 *   (<= innerMaxIter steps, lambda0 = 1e-3, x10 on reject / /10 on accept, additive damping lambda*I),
 *   (a free point with fewer than two inlier measurements is held for that run: its depth is unconstrained),
 *   then every measurement with reprojection error > maxErr is flagged outlier (Meas2D::outlier = 1,
 *   consumed at SL_CoSLAMRobustBA.cpp:298-306) and leaves the next round; stop when the flags no longer
 *   change.  Jacobians are analytic; everything is binary64.
 */
#include "ba_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static void so3_exp(const double w[3], double R[9]) { /* same map as SL_IntraCamPose.cpp:10-39 */
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (th == 0) {
        memset(R, 0, sizeof(double) * 9);
        R[0] = R[4] = R[8] = 1.0;
        return;
    }
    double h0 = w[0] / th, h1 = w[1] / th, h2 = w[2] / th;
    double st = sin(th), ct = 1 - cos(th);
    R[0] = -ct * h1 * h1 - ct * h2 * h2 + 1;
    R[1] = ct * h0 * h1 - st * h2;
    R[2] = st * h1 + ct * h0 * h2;
    R[3] = st * h2 + ct * h0 * h1;
    R[4] = -ct * h0 * h0 - ct * h2 * h2 + 1;
    R[5] = ct * h1 * h2 - st * h0;
    R[6] = ct * h0 * h2 - st * h1;
    R[7] = st * h0 + ct * h1 * h2;
    R[8] = -ct * h0 * h0 - ct * h1 * h1 + 1;
}

static void mat33AB(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, T, sizeof(T));
}

/* residual e = m - pi(K(RM+t)), and optionally Jc (2x6: [d/dw | d/dt]) and Jp (2x3). returns 0 if behind camera */
int oba_residual(const double* K, const double* R, const double* t, const double* M, const double* m, double* e,
                 double* Jc, double* Jp) {
    double X[3];
    for (int r = 0; r < 3; ++r) X[r] = R[3 * r] * M[0] + R[3 * r + 1] * M[1] + R[3 * r + 2] * M[2] + t[r];
    double u = K[0] * X[0] + K[1] * X[1] + K[2] * X[2];
    double v = K[3] * X[0] + K[4] * X[1] + K[5] * X[2];
    double w = K[6] * X[0] + K[7] * X[1] + K[8] * X[2];
    if (!(w > 1e-12)) {
        e[0] = e[1] = 1e150;
        if (Jc) memset(Jc, 0, sizeof(double) * 12);
        if (Jp) memset(Jp, 0, sizeof(double) * 6);
        return 0;
    }
    double mx = u / w, my = v / w;
    e[0] = m[0] - mx;
    e[1] = m[1] - my;
    if (Jc || Jp) {
        double A[6]; /* d pi / d X */
        for (int c = 0; c < 3; ++c) {
            A[c] = (K[c] - mx * K[6 + c]) / w;
            A[3 + c] = (K[3 + c] - my * K[6 + c]) / w;
        }
        /* dX/dw = -R [M]x */
        double Mx[9] = {0, -M[2], M[1], M[2], 0, -M[0], -M[1], M[0], 0};
        double RMx[9];
        mat33AB(R, Mx, RMx);
        if (Jc) {
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 3; ++c) {
                    Jc[6 * r + c] = -(A[3 * r] * RMx[c] + A[3 * r + 1] * RMx[3 + c] + A[3 * r + 2] * RMx[6 + c]);
                    Jc[6 * r + 3 + c] = A[3 * r + c];
                }
        }
        if (Jp) {
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 3; ++c)
                    Jp[3 * r + c] = A[3 * r] * R[c] + A[3 * r + 1] * R[3 + c] + A[3 * r + 2] * R[6 + c];
        }
    }
    return 1;
}

static int inv33(const double* V, double* Vi) {
    double a = V[0], b = V[1], c = V[2], d = V[3], e = V[4], f = V[5], g = V[6], h = V[7], i = V[8];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    if (!(fabs(det) > 0)) return 0;
    double r = 1.0 / det;
    Vi[0] = A * r;
    Vi[1] = -(b * i - c * h) * r;
    Vi[2] = (b * f - c * e) * r;
    Vi[3] = B * r;
    Vi[4] = (a * i - c * g) * r;
    Vi[5] = -(a * f - c * d) * r;
    Vi[6] = C * r;
    Vi[7] = -(a * h - b * g) * r;
    Vi[8] = (a * e - b * d) * r;
    return 1;
}

/* in-place lower Cholesky of the n x n SPD matrix S (row-major, lower triangle used) and solve S x = b */
int oba_cholesky_solve(int n, double* S, double* b) {
    for (int j = 0; j < n; ++j) {
        double d = S[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= S[(size_t)j * n + k] * S[(size_t)j * n + k];
        if (!(d > 0)) return 0;
        d = sqrt(d);
        S[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = S[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= S[(size_t)i * n + k] * S[(size_t)j * n + k];
            S[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= S[(size_t)i * n + k] * b[k];
        b[i] = s / S[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= S[(size_t)k * n + i] * b[k];
        b[i] = s / S[(size_t)i * n + i];
    }
    return 1;
}

typedef struct {
    int C, P, nObs, nCamsCon, nPtsCon;
    const double* Ks;
    const int* obs_ptr;
    const int* obs_cam;
    const double* obs_xy;
} prob_t;

static double cost_of(const prob_t* p, const double* Rs, const double* Ts, const double* pts, const int* outlier) {
    double c = 0;
    for (int i = 0; i < p->P; ++i)
        for (int o = p->obs_ptr[i]; o < p->obs_ptr[i + 1]; ++o) {
            if (outlier[o]) continue;
            int j = p->obs_cam[o];
            double e[2];
            oba_residual(p->Ks + 9 * j, Rs + 9 * j, Ts + 3 * j, pts + 3 * i, p->obs_xy + 2 * o, e, NULL, NULL);
            c += e[0] * e[0] + e[1] * e[1];
        }
    return c;
}

/* one LM run over the current inlier set; returns the number of iterations performed */
static int lm_run(const prob_t* p, double* Rs, double* Ts, double* pts, const int* outlier, int maxIter,
                  double* cost_out, double* lambda_io) {
    const int C = p->C, P = p->P, nc = C - p->nCamsCon, n = 6 * nc;
    double lambda = *lambda_io;
    double cost = cost_of(p, Rs, Ts, pts, outlier);
    double* U = (double*)malloc(sizeof(double) * 36 * (nc > 0 ? nc : 1));
    double* gc = (double*)malloc(sizeof(double) * 6 * (nc > 0 ? nc : 1));
    double* Vinv = (double*)malloc(sizeof(double) * 9 * P);
    double* gp = (double*)malloc(sizeof(double) * 3 * P);
    double* W = (double*)malloc(sizeof(double) * 18 * (p->nObs > 0 ? p->nObs : 1));
    double* S = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (n > 0 ? n : 1));
    double* rhs = (double*)malloc(sizeof(double) * (n > 0 ? n : 1));
    double* Rn = (double*)malloc(sizeof(double) * 9 * C);
    double* Tn = (double*)malloc(sizeof(double) * 3 * C);
    double* Mn = (double*)malloc(sizeof(double) * 3 * P);
    int it = 0;
    for (; it < maxIter; ++it) {
        memset(U, 0, sizeof(double) * 36 * (nc > 0 ? nc : 1));
        memset(gc, 0, sizeof(double) * 6 * (nc > 0 ? nc : 1));
        memset(S, 0, sizeof(double) * (size_t)(n > 0 ? n : 1) * (n > 0 ? n : 1));
        /* linearise: U_j, V_i, W_ij, gradients */
        for (int i = 0; i < P; ++i) {
            double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            int nIn = 0;
            for (int o = p->obs_ptr[i]; o < p->obs_ptr[i + 1]; ++o) nIn += outlier[o] ? 0 : 1;
            /* a point seen by fewer than two inlier measurements has no depth constraint: hold it */
            int freeP = (i >= p->nPtsCon) && (nIn >= 2);
            for (int o = p->obs_ptr[i]; o < p->obs_ptr[i + 1]; ++o) {
                memset(W + 18 * o, 0, sizeof(double) * 18);
                if (outlier[o]) continue;
                int j = p->obs_cam[o];
                int freeC = (j >= p->nCamsCon);
                double e[2], Jc[12], Jp[6];
                oba_residual(p->Ks + 9 * j, Rs + 9 * j, Ts + 3 * j, pts + 3 * i, p->obs_xy + 2 * o, e, Jc, Jp);
                if (freeC) {
                    double* Uj = U + 36 * (j - p->nCamsCon);
                    double* gj = gc + 6 * (j - p->nCamsCon);
                    for (int r = 0; r < 6; ++r) {
                        for (int c = 0; c < 6; ++c) Uj[6 * r + c] += Jc[r] * Jc[c] + Jc[6 + r] * Jc[6 + c];
                        gj[r] += Jc[r] * e[0] + Jc[6 + r] * e[1];
                    }
                }
                if (freeP) {
                    for (int r = 0; r < 3; ++r) {
                        for (int c = 0; c < 3; ++c) V[3 * r + c] += Jp[r] * Jp[c] + Jp[3 + r] * Jp[3 + c];
                        g[r] += Jp[r] * e[0] + Jp[3 + r] * e[1];
                    }
                }
                if (freeC && freeP)
                    for (int r = 0; r < 6; ++r)
                        for (int c = 0; c < 3; ++c) W[18 * o + 3 * r + c] = Jc[r] * Jp[c] + Jc[6 + r] * Jp[3 + c];
            }
            if (freeP) {
                V[0] += lambda;
                V[4] += lambda;
                V[8] += lambda;
                if (!inv33(V, Vinv + 9 * i)) memset(Vinv + 9 * i, 0, sizeof(double) * 9);
            } else {
                memset(Vinv + 9 * i, 0, sizeof(double) * 9);
            }
            memcpy(gp + 3 * i, g, sizeof(g));
        }
        /* reduced camera system S = U + lambda I - sum_i W V^-1 W^T ; rhs = gc - sum_i W V^-1 gp */
        for (int j = 0; j < nc; ++j) {
            for (int r = 0; r < 6; ++r) {
                for (int c = 0; c < 6; ++c) S[(size_t)(6 * j + r) * n + 6 * j + c] = U[36 * j + 6 * r + c];
                S[(size_t)(6 * j + r) * n + 6 * j + r] += lambda;
                rhs[6 * j + r] = gc[6 * j + r];
            }
        }
        for (int i = p->nPtsCon; i < P; ++i) {
            const double* Vi = Vinv + 9 * i;
            for (int oa = p->obs_ptr[i]; oa < p->obs_ptr[i + 1]; ++oa) {
                int ja = p->obs_cam[oa] - p->nCamsCon;
                if (ja < 0 || outlier[oa]) continue;
                double Y[18]; /* W_ia V_i^-1 */
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 3; ++c)
                        Y[3 * r + c] = W[18 * oa + 3 * r] * Vi[c] + W[18 * oa + 3 * r + 1] * Vi[3 + c] +
                                       W[18 * oa + 3 * r + 2] * Vi[6 + c];
                for (int r = 0; r < 6; ++r)
                    rhs[6 * ja + r] -= Y[3 * r] * gp[3 * i] + Y[3 * r + 1] * gp[3 * i + 1] + Y[3 * r + 2] * gp[3 * i + 2];
                for (int ob = p->obs_ptr[i]; ob < p->obs_ptr[i + 1]; ++ob) {
                    int jb = p->obs_cam[ob] - p->nCamsCon;
                    if (jb < 0 || outlier[ob]) continue;
                    for (int r = 0; r < 6; ++r)
                        for (int c = 0; c < 6; ++c)
                            S[(size_t)(6 * ja + r) * n + 6 * jb + c] -=
                                Y[3 * r] * W[18 * ob + 3 * c] + Y[3 * r + 1] * W[18 * ob + 3 * c + 1] +
                                Y[3 * r + 2] * W[18 * ob + 3 * c + 2];
                }
            }
        }
        int ok = (n == 0) ? 1 : oba_cholesky_solve(n, S, rhs);
        double cost_new = 1e300;
        double step2 = 0;
        if (ok) {
            /* back-substitute points, apply the step to copies */
            memcpy(Rn, Rs, sizeof(double) * 9 * C);
            memcpy(Tn, Ts, sizeof(double) * 3 * C);
            memcpy(Mn, pts, sizeof(double) * 3 * P);
            for (int j = 0; j < nc; ++j) {
                double dR[9];
                so3_exp(rhs + 6 * j, dR);
                mat33AB(Rs + 9 * (j + p->nCamsCon), dR, Rn + 9 * (j + p->nCamsCon));
                for (int r = 0; r < 3; ++r) Tn[3 * (j + p->nCamsCon) + r] += rhs[6 * j + 3 + r];
                for (int r = 0; r < 6; ++r) step2 += rhs[6 * j + r] * rhs[6 * j + r];
            }
            for (int i = p->nPtsCon; i < P; ++i) {
                double b[3] = {gp[3 * i], gp[3 * i + 1], gp[3 * i + 2]};
                for (int o = p->obs_ptr[i]; o < p->obs_ptr[i + 1]; ++o) {
                    int j = p->obs_cam[o] - p->nCamsCon;
                    if (j < 0 || outlier[o]) continue;
                    for (int c = 0; c < 3; ++c)
                        for (int r = 0; r < 6; ++r) b[c] -= W[18 * o + 3 * r + c] * rhs[6 * j + r];
                }
                const double* Vi = Vinv + 9 * i;
                for (int r = 0; r < 3; ++r) {
                    double d = Vi[3 * r] * b[0] + Vi[3 * r + 1] * b[1] + Vi[3 * r + 2] * b[2];
                    Mn[3 * i + r] += d;
                    step2 += d * d;
                }
            }
            cost_new = cost_of(p, Rn, Tn, Mn, outlier);
        }
        if (ok && cost_new <= cost) {
            double dec = cost - cost_new;
            memcpy(Rs, Rn, sizeof(double) * 9 * C);
            memcpy(Ts, Tn, sizeof(double) * 3 * C);
            memcpy(pts, Mn, sizeof(double) * 3 * P);
            cost = cost_new;
            lambda /= 10;
            if (dec < 1e-9 * cost + 1e-15 || step2 < 1e-20) {
                ++it;
                break;
            }
        } else {
            lambda *= 10;
            if (lambda > 1e12) {
                ++it;
                break;
            }
        }
    }
    free(U);
    free(gc);
    free(Vinv);
    free(gp);
    free(W);
    free(S);
    free(rhs);
    free(Rn);
    free(Tn);
    free(Mn);
    *cost_out = cost;
    *lambda_io = lambda;
    return it;
}

int oba_robust(int C, int P, int nObs, const double* Ks, double* Rs, double* Ts, double* pts, const int* obs_ptr,
               const int* obs_cam, const double* obs_xy, int nCamsCon, int nPtsCon, double maxErr, int maxIter,
               int innerMaxIter, int* outlier, oba_stats* st) {
    prob_t p = {C, P, nObs, nCamsCon, nPtsCon, Ks, obs_ptr, obs_cam, obs_xy};
    if (nCamsCon > C) p.nCamsCon = C;
    if (nPtsCon > P) p.nPtsCon = P;
    memset(outlier, 0, sizeof(int) * (nObs > 0 ? nObs : 0));
    double cost = 0;
    int totalIt = 0, outer = 0;
    if (st) st->cost0 = cost_of(&p, Rs, Ts, pts, outlier);
    for (; outer < maxIter; ++outer) {
        double lambda = 1e-3;
        totalIt += lm_run(&p, Rs, Ts, pts, outlier, innerMaxIter, &cost, &lambda);
        int changed = 0;
        const double thr2 = maxErr * maxErr;
        for (int i = 0; i < P; ++i)
            for (int o = obs_ptr[i]; o < obs_ptr[i + 1]; ++o) {
                int j = obs_cam[o];
                double e[2];
                oba_residual(Ks + 9 * j, Rs + 9 * j, Ts + 3 * j, pts + 3 * i, obs_xy + 2 * o, e, NULL, NULL);
                int out = (e[0] * e[0] + e[1] * e[1] > thr2) ? 1 : 0;
                if (out != outlier[o]) changed = 1;
                outlier[o] = out;
            }
        if (!changed) {
            ++outer;
            break;
        }
    }
    if (st) {
        st->cost = cost_of(&p, Rs, Ts, pts, outlier);
        st->nIterTotal = totalIt;
        st->nOuter = outer;
        int no = 0;
        for (int o = 0; o < nObs; ++o) no += outlier[o];
        st->nOutliers = no;
    }
    return 1;
}
