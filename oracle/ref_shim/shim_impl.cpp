// ref_shim/shim_impl.cpp -- definitions for the stand-in headers (TEST INFRASTRUCTURE, see SL_LinAlg.h).
#include "math/SL_LinAlg.h"
#include "geometry/SL_Geometry.h"

#include <cmath>
#include <vector>

void doubleArrCopy(double* dst, int off, const double* src, int n) { memcpy(dst + (size_t)off * n, src, sizeof(double) * n); }

void mat33AB(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, T, sizeof(T));
}

void matATB(int m, int n, int p, int q, const double* A, const double* B, double* C) {
    (void)p;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < q; ++j) {
            double s = 0;
            for (int k = 0; k < m; ++k) s += A[k * n + i] * B[k * q + j];
            C[i * q + j] = s;
        }
}

void matAB(int m, int n, int p, int q, const double* A, const double* B, double* C) {
    (void)p;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < q; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * q + j];
            C[i * q + j] = s;
        }
}

void matInv(int n, const double* A, double* invA) {
    std::vector<double> M((size_t)n * 2 * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            M[(size_t)i * 2 * n + j] = A[i * n + j];
            M[(size_t)i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (fabs(M[(size_t)r * 2 * n + c]) > fabs(M[(size_t)piv * 2 * n + c])) piv = r;
        if (piv != c)
            for (int j = 0; j < 2 * n; ++j) std::swap(M[(size_t)c * 2 * n + j], M[(size_t)piv * 2 * n + j]);
        double d = M[(size_t)c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) M[(size_t)c * 2 * n + j] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = M[(size_t)r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) M[(size_t)r * 2 * n + j] -= f * M[(size_t)c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) invA[i * n + j] = M[(size_t)i * 2 * n + n + j];
}

void mat22Inv(const double* A, double* invA) {
    const double det = A[0] * A[3] - A[1] * A[2];
    const double a = A[0], b = A[1], c = A[2], d = A[3];
    invA[0] = d / det;
    invA[1] = -b / det;
    invA[2] = -c / det;
    invA[3] = a / det;
}
void matScale(int m, int n, const double* A, double s, double* B) {
    for (int i = 0; i < m * n; ++i) B[i] = A[i] * s;
}
void mat33Inv(const double* A, double* invA) { matInv(3, A, invA); }

void project(const double* K, const double* R, const double* t, const double* M, double* m) {
    double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
    double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
    double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
    double u = K[0] * X + K[1] * Y + K[2] * Z;
    double v = K[3] * X + K[4] * Y + K[5] * Z;
    double w = K[6] * X + K[7] * Y + K[8] * Z;
    m[0] = u / w;
    m[1] = v / w;
}

double reprojError2(const double* K, const double* R, const double* t, int npts, const double* Ms, const double* ms) {
    double e = 0, rm[2];
    for (int i = 0; i < npts; ++i) {
        project(K, R, t, Ms + 3 * i, rm);
        double dx = ms[2 * i] - rm[0], dy = ms[2 * i + 1] - rm[1];
        e += dx * dx + dy * dy;
    }
    return e;
}

void getProjectionCovMat(const double* K, const double* R, const double* t, const double* M, const double* cov,
                         double* var, double sigma) {
    double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
    double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
    double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
    double KR[9];
    mat33AB(K, R, KR);
    double u = K[0] * X + K[1] * Y + K[2] * Z, v = K[3] * X + K[4] * Y + K[5] * Z, w = K[6] * X + K[7] * Y + K[8] * Z;
    double J[6];
    for (int j = 0; j < 3; ++j) {
        J[j] = (KR[j] * w - u * KR[6 + j]) / (w * w);
        J[3 + j] = (KR[3 + j] * w - v * KR[6 + j]) / (w * w);
    }
    double JC[6];
    matAB(2, 3, 3, 3, J, cov, JC);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += JC[3 * i + k] * J[3 * j + k];
            var[2 * i + j] = s + ((i == j) ? sigma * sigma : 0.0);
        }
}

double mahaDist2(const double* a, const double* b, const double* ivar) {
    double dx = a[0] - b[0], dy = a[1] - b[1];
    return dx * (ivar[0] * dx + ivar[1] * dy) + dy * (ivar[2] * dx + ivar[3] * dy);
}

void formEMat(const double* R1, const double* t1, const double* R2, const double* t2, double* E) {
    double R1t[9] = {R1[0], R1[3], R1[6], R1[1], R1[4], R1[7], R1[2], R1[5], R1[8]};
    double R[9];
    mat33AB(R2, R1t, R);
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = t2[i] - (R[3 * i] * t1[0] + R[3 * i + 1] * t1[1] + R[3 * i + 2] * t1[2]);
    double Tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    mat33AB(Tx, R, E);
}

void getFMat(const double* invK1, const double* invK2, const double* E, double* F) {
    double T[9];
    matATB(3, 3, 3, 3, invK2, E, T);
    mat33AB(T, invK1, F);
}

void computeEpipolarLine(const double* F, double x, double y, double* l) {
    for (int i = 0; i < 3; ++i) l[i] = F[3 * i] * x + F[3 * i + 1] * y + F[3 * i + 2];
}

double epipolarError(const double* F, const double* m2, const double* m1) {
    double l[3];
    computeEpipolarLine(F, m1[0], m1[1], l);
    double n = sqrt(l[0] * l[0] + l[1] * l[1]);
    return fabs(l[0] * m2[0] + l[1] * m2[1] + l[2]) / (n > 0 ? n : 1.0);
}
