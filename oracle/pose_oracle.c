/*
 * oracle/pose_oracle.c -- CPU restatement of intraCamEstimate (reference src/slam/SL_IntraCamPose.cpp).
 * TEST INFRASTRUCTURE ONLY (see klt_oracle.h).
 *
 * PARITY PINNED against the reference's own code: oracle/Makefile compiles the reference's
 * SL_IntraCamPose.cpp in place into oracle/_ref/libintracam_ref.so (against ref_shim/, a stand-in for the
 * six un-vendored LibVisualSLAM helpers it calls) and tests/test_pose_oracle.py checks this restatement
 * against it on seeded problems; tests/golden/pose_golden.npz holds vectors generated from that binary.
 */
#include "pose_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void okp_option_default(okp_option* o) { /* SL_IntraCamPose.h:42-46 */
    memset(o, 0, sizeof(*o));
    o->maxIterLM = 100;
    o->maxIterRW = 5;
    o->epsErrorChangeLM = 1e-7;
    o->epsParamChangeLM = 1e-6;
    o->epsErrorChangeRW = 1e-6;
    o->lambda0 = 1e-3;
}

void okp_so3_exp(const double w[3], double R[9]) { /* SL_IntraCamPose.cpp:10-39 */
    double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (theta == 0) {
        memset(R, 0, sizeof(double) * 9);
        R[0] = R[4] = R[8] = 1.0;
        return;
    }
    double hw[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    double st = sin(theta);
    double ct = 1 - cos(theta);
    double hw0hw0 = hw[0] * hw[0], hw0hw1 = hw[0] * hw[1], hw0hw2 = hw[0] * hw[2];
    double hw1hw1 = hw[1] * hw[1], hw1hw2 = hw[1] * hw[2], hw2hw2 = hw[2] * hw[2];
    R[0] = -ct * hw1hw1 - ct * hw2hw2 + 1;
    R[1] = ct * hw0hw1 - st * hw[2];
    R[2] = st * hw[1] + ct * hw0hw2;
    R[3] = st * hw[2] + ct * hw0hw1;
    R[4] = -ct * hw0hw0 - ct * hw2hw2 + 1;
    R[5] = ct * hw1hw2 - st * hw[0];
    R[6] = ct * hw0hw2 - st * hw[1];
    R[7] = st * hw[0] + ct * hw1hw2;
    R[8] = -ct * hw0hw0 - ct * hw1hw1 + 1;
}

static void mat33AB(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, T, sizeof(T));
}

void okp_project(const double* K, const double* R, const double* t, const double* M, double* m) {
    /* LibVisualSLAM project(): m = pi(K (R M + t)) (semantics from SL_IntraCamPose.cpp:60,99,234) */
    double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
    double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
    double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
    double u = K[0] * X + K[1] * Y + K[2] * Z;
    double v = K[3] * X + K[4] * Y + K[5] * Z;
    double w = K[6] * X + K[7] * Y + K[8] * Z;
    m[0] = u / w;
    m[1] = v / w;
}

/* n x n inverse, Gauss-Jordan with partial pivoting on [A | I] (the reference calls LAPACK here) */
void okp_mat_inv(int n, const double* A, double* invA) {
    double M[2 * 12 * 12];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            M[i * 2 * n + j] = A[i * n + j];
            M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (fabs(M[r * 2 * n + c]) > fabs(M[piv * 2 * n + c])) piv = r;
        if (piv != c)
            for (int j = 0; j < 2 * n; ++j) {
                double tmp = M[c * 2 * n + j];
                M[c * 2 * n + j] = M[piv * 2 * n + j];
                M[piv * 2 * n + j] = tmp;
            }
        double d = M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) M[c * 2 * n + j] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) invA[i * n + j] = M[i * 2 * n + n + j];
}

/* forward-difference Jacobians, eps = 1e-8 (SL_IntraCamPose.cpp:43-117) */
static void jac_num(const double* K, const double* R, const double* t, const double* M, const double* rm0, double* Jw,
                    double* Jt) {
    const double eps = 1e-8;
    double rm[2], dR[9], R1[9];
    for (int a = 0; a < 3; ++a) {
        double w[3] = {0, 0, 0};
        w[a] = eps;
        okp_so3_exp(w, dR);
        mat33AB(R, dR, R1);
        okp_project(K, R1, t, M, rm);
        Jw[a] = (rm[0] - rm0[0]) / eps;
        Jw[3 + a] = (rm[1] - rm0[1]) / eps;
    }
    for (int a = 0; a < 3; ++a) {
        double t1[3] = {t[0], t[1], t[2]};
        t1[a] = t[a] + eps;
        okp_project(K, R, t1, M, rm);
        Jt[a] = (rm[0] - rm0[0]) / eps;
        Jt[3 + a] = (rm[1] - rm0[1]) / eps;
    }
}

/* SL_IntraCamPose.cpp:259-303 */
static void weighted_lm_step(const double* K, const double* R, const double* t, int npts, const double* Ws,
                             const double* Ms, const double* ms, double* param, double lambda) {
    double sA[36], invSA[36], sB[6], rm[2], Jw[6], Jt[6];
    memset(sA, 0, sizeof(sA));
    memset(sB, 0, sizeof(sB));
    for (int i = 0; i < npts; ++i) {
        const double* pM = Ms + 3 * i;
        const double* pm = ms + 2 * i;
        okp_project(K, R, t, pM, rm);
        jac_num(K, R, t, pM, rm, Jw, Jt);
        double w = Ws[i];
        double J[12] = {w * Jw[0], w * Jw[1], w * Jw[2], w * Jt[0], w * Jt[1], w * Jt[2],
                        w * Jw[3], w * Jw[4], w * Jw[5], w * Jt[3], w * Jt[4], w * Jt[5]};
        double rerr[2] = {(-rm[0] + pm[0]) * w, (-rm[1] + pm[1]) * w};
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) sA[6 * r + c] += J[r] * J[c] + J[6 + r] * J[6 + c]; /* matATB(2,6,2,6) */
            sB[r] += J[r] * rerr[0] + J[6 + r] * rerr[1];
        }
    }
    for (int d = 0; d < 6; ++d) sA[7 * d] += lambda;
    okp_mat_inv(6, sA, invSA);
    for (int r = 0; r < 6; ++r) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += invSA[6 * r + c] * sB[c];
        param[r] = s;
    }
}

static void update_pose(const double* R, const double* t, const double* p, double* Rn, double* tn) { /* :367-380 */
    double dR[9];
    okp_so3_exp(p, dR);
    mat33AB(R, dR, Rn);
    tn[0] = t[0] + p[3];
    tn[1] = t[1] + p[4];
    tn[2] = t[2] + p[5];
}

static double reproj_err2_weighted(const double* K, const double* R, const double* t, int npts, const double* Ws,
                                   const double* Ms, const double* ms) { /* :439-456 */
    double rm[2], err = 0;
    for (int i = 0; i < npts; ++i) {
        okp_project(K, R, t, Ms + 3 * i, rm);
        double dx = ms[2 * i] - rm[0], dy = ms[2 * i + 1] - rm[1];
        err += (dx * dx + dy * dy) * Ws[i];
    }
    return err;
}

/* SL_IntraCamPose.cpp:475-549 */
int okp_weighted_lm(const double* K, const double* R0, const double* t0, int npts, const double* Ws, const double* Ms,
                    const double* ms, double* R_opt, double* t_opt, okp_option* opt) {
    double param[6];
    opt->npts = npts;
    opt->lambda = opt->lambda0;
    opt->err0 = reproj_err2_weighted(K, R0, t0, npts, Ws, Ms, ms);
    opt->err = opt->err0;
    double R[9], t[3], R_tmp[9], t_tmp[3];
    memcpy(R, R0, sizeof(R));
    memcpy(t, t0, sizeof(t));
    /* the reference leaves R_tmp/t_tmp uninitialised until the first accepted step; we start them at R0/t0 */
    memcpy(R_tmp, R0, sizeof(R));
    memcpy(t_tmp, t0, sizeof(t));
    opt->retTypeLM = 1;
    int i = 0;
    double err = opt->err0;
    for (; i < opt->maxIterLM; ++i) {
        weighted_lm_step(K, R, t, npts, Ws, Ms, ms, param, opt->lambda);
        update_pose(R, t, param, R_opt, t_opt);
        double p2 = param[0] * param[0] + param[1] * param[1] + param[2] * param[2] + param[3] * param[3] +
                    param[4] * param[4] + param[5] * param[5];
        if (p2 < opt->epsParamChangeLM) {
            memcpy(R, R_opt, sizeof(R));
            memcpy(t, t_opt, sizeof(t));
            opt->retTypeLM = 0;
            break;
        }
        err = reproj_err2_weighted(K, R_opt, t_opt, npts, Ws, Ms, ms);
        if (fabs(err - opt->err) < opt->epsErrorChangeLM) {
            opt->retTypeLM = 0;
            break;
        }
        if (err <= opt->err) {
            memcpy(R, R_opt, sizeof(R));
            memcpy(t, t_opt, sizeof(t));
            memcpy(R_tmp, R_opt, sizeof(R));
            memcpy(t_tmp, t_opt, sizeof(t));
            opt->err = err;
            opt->lambda /= 10;
        } else {
            opt->lambda *= 10;
            if (opt->lambda > 1e+18) {
                opt->retTypeLM = -1;
                break;
            }
        }
    }
    if (opt->retTypeLM == -1) {
        memcpy(R_opt, R_tmp, sizeof(R));
        memcpy(t_opt, t_tmp, sizeof(t));
    }
    opt->err = err;
    opt->nIterLM = i;
    return opt->retTypeLM >= 0;
}

static double tukey(double e, double tau) { /* :646-653, :693-699 */
    if (e >= tau) return 0;
    e /= tau;
    e = 1 - e * e;
    return e * e;
}

/* SL_IntraCamPose.cpp:626-709 */
int okp_intracam_estimate(const double* K, const double* R0, const double* t0, int npts, const double* prevErrs,
                          const double* Ms, const double* ms, double tau, double* R_opt, double* t_opt,
                          okp_option* opt) {
    double* Ws = (double*)malloc(sizeof(double) * (npts > 0 ? npts : 1));
    for (int i = 0; i < npts; ++i) Ws[i] = prevErrs ? tukey(fabs(prevErrs[i]), tau) : 1.0;
    double R[9], t[3];
    memcpy(R, R0, sizeof(R));
    memcpy(t, t0, sizeof(t));
    int ret = 1, k = 0;
    opt->errRW = -1;
    for (; k < opt->maxIterRW; ++k) {
        if (!okp_weighted_lm(K, R, t, npts, Ws, Ms, ms, R_opt, t_opt, opt)) {
            ret = 0;
            break;
        }
        opt->lambda0 = opt->lambda;
        if (opt->errRW < 0) {
            opt->errRW = opt->err;
        } else {
            if (fabs(opt->err - opt->errRW) < opt->epsErrorChangeRW) {
                ret = 1;
                break;
            }
            opt->errRW = opt->err;
        }
        memcpy(R, R_opt, sizeof(R));
        memcpy(t, t_opt, sizeof(t));
        for (int i = 0; i < npts; ++i) {
            double rm[2];
            okp_project(K, R, t, Ms + 3 * i, rm);
            double dx = rm[0] - ms[2 * i], dy = rm[1] - ms[2 * i + 1];
            Ws[i] = tukey(sqrt(dx * dx + dy * dy), tau);
        }
    }
    opt->nIterRW = k;
    free(Ws);
    return ret;
}
