// ref_shim/ref_glue_impl.cpp -- definitions for the stand-in headers the reference's tracker / BA callers need
// (TEST INFRASTRUCTURE, see math/SL_Matrix.h).
#include <cmath>

#include "SL_error.h"
#include "geometry/SL_Distortion.h"
#include "geometry/SL_Geometry.h"
#include "geometry/SL_RigidTransform.h"
#include "geometry/SL_Triangulate.h"
#include "math/SL_LinAlg.h"

void repErr(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw SL_Exception(buf);
}
void warn(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
void logInfo(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stdout, fmt, ap);
    va_end(ap);
}

// our definition (see SL_Distortion.h): the same operations in the same order as handback.hip / handback_oracle.c
void undistorPoint(const double* K, const double* kud, const double* in, double* out) {
    const double yn = (in[1] - K[5]) / K[4];
    const double xn = ((in[0] - K[2]) - K[1] * yn) / K[0];
    const double r2 = xn * xn + yn * yn;
    double f = kud[6];
    for (int i = 5; i >= 0; --i) f = f * r2 + kud[i];
    f = 1.0 + f * r2;
    const double xu = xn * f, yu = yn * f;
    out[0] = (K[0] * xu + K[1] * yu) + K[2];
    out[1] = K[4] * yu + K[5];
}

void mat33Trans(const double* A, double* At) {
    double T[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]};
    for (int i = 0; i < 9; ++i) At[i] = T[i];
}
double dist2(const double* a, const double* b) { return sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1])); }
void invRigidTransFromTo(const double* R, const double* t, double* iR, double* it) {
    mat33Trans(R, iR);
    for (int r = 0; r < 3; ++r) it[r] = -(iR[3 * r] * t[0] + iR[3 * r + 1] * t[1] + iR[3 * r + 2] * t[2]);
}
void getRigidTransFromTo(const double* R1, const double* t1, const double* R2, const double* t2, double* R, double* t) {
    double R1t[9];
    mat33Trans(R1, R1t);
    mat33AB(R2, R1t, R);
    for (int r = 0; r < 3; ++r) t[r] = t2[r] - (R[3 * r] * t1[0] + R[3 * r + 1] * t1[1] + R[3 * r + 2] * t1[2]);
}
// One Kalman / Gauss-Newton update of a map point and its covariance from a new measurement (our definition of the
// external helper; used by the Mahalanobis post-pass of InterCamPoseEstimator::apply, off the product's path).
void seqTriangulate(const double* K, const double* R, const double* t, const double* m, double* M, double* cov, double sigma) {
    double X[3], rm[2];
    for (int r = 0; r < 3; ++r) X[r] = R[3 * r] * M[0] + R[3 * r + 1] * M[1] + R[3 * r + 2] * M[2] + t[r];
    project(K, R, t, M, rm);
    const double w = K[6] * X[0] + K[7] * X[1] + K[8] * X[2];
    double J[6];  // d project / d M (2 x 3)
    for (int c = 0; c < 3; ++c) {
        const double du = K[0] * R[c] + K[1] * R[3 + c] + K[2] * R[6 + c], dv = K[3] * R[c] + K[4] * R[3 + c] + K[5] * R[6 + c];
        const double dw = K[6] * R[c] + K[7] * R[3 + c] + K[8] * R[6 + c];
        J[c] = (du - rm[0] * dw) / w;
        J[3 + c] = (dv - rm[1] * dw) / w;
    }
    double PJt[6], S[4], iS[4];  // cov J^T (3 x 2), S = J cov J^T + sigma^2 I
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 2; ++c) PJt[2 * r + c] = cov[3 * r] * J[3 * c] + cov[3 * r + 1] * J[3 * c + 1] + cov[3 * r + 2] * J[3 * c + 2];
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) S[2 * r + c] = J[3 * r] * PJt[c] + J[3 * r + 1] * PJt[2 + c] + J[3 * r + 2] * PJt[4 + c] + (r == c ? sigma * sigma : 0);
    mat22Inv(S, iS);
    double Kg[6];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 2; ++c) Kg[2 * r + c] = PJt[2 * r] * iS[c] + PJt[2 * r + 1] * iS[2 + c];
    const double e[2] = {m[0] - rm[0], m[1] - rm[1]};
    for (int r = 0; r < 3; ++r) M[r] += Kg[2 * r] * e[0] + Kg[2 * r + 1] * e[1];
    double nc[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) nc[3 * r + c] = cov[3 * r + c] - (Kg[2 * r] * PJt[2 * c] + Kg[2 * r + 1] * PJt[2 * c + 1]);
    for (int i = 0; i < 9; ++i) cov[i] = nc[i];
}
