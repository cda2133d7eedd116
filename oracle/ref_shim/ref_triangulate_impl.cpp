// ref_shim/ref_triangulate_impl.cpp -- TEST INFRASTRUCTURE: the LibVisualSLAM helpers src/slam/SL_CoSLAMHelper.cpp's
// updateStaticPointPosition / updateDynamicPointPosition call (:338-394, :455-484).  LibVisualSLAM is not vendored in the reference
// (only the calls are there), so these are OUR definitions -- the ones oracle/poseupdate_oracle.c and coslam_amd/csrc/
// poseupdate.hip use, in the same operation order (the golden vectors of tests/cxx/ref_update_points_test.cpp therefore pin the
// reference's LOOPS: which views are taken, in which order, which points are touched):
//   getInvK               inverse of an upper-triangular K
//   normPoint             the dehomogenised iK (m, 1)
//   getCameraCenter       C = -R^T t
//   getAbsRadiansBetween  the angle at M between C0 - M and C - M
//   triangulateMultiView  linear least squares of (r1 - x r3) M = x t3 - t1, (r2 - y r3) M = y t3 - t2 over the views, normal
//                         equations, symmetric 3x3 inverse by cofactors
//   getTriangulateCovMat  sigma^2 (sum_i J_i^T J_i)^-1 with J_i = d project_i / dM at M
//   isAtCameraBack, dist3 (isDynamicPoint, :251-312)
//   reprojErrorSingle     Euclidean pixel distance of m from the projection of M (NewMapPtsNCC::reconstructTracks)
#include <cmath>
#include <cstring>

#define REF_SHIM_TRIANGULATE_ON_PATH
#include "ref_not_on_path.h"

void getInvK(const double* K, double* iK) {
    const double fx = K[0], s = K[1], cx = K[2], fy = K[4], cy = K[5];
    iK[0] = 1.0 / fx, iK[1] = -s / (fx * fy), iK[2] = (s * cy - cx * fy) / (fx * fy);
    iK[3] = 0, iK[4] = 1.0 / fy, iK[5] = -cy / fy;
    iK[6] = 0, iK[7] = 0, iK[8] = 1;
}
void normPoint(const double* iK, const double* m, double* nm) {
    const double w = (iK[6] * m[0] + iK[7] * m[1]) + iK[8];
    nm[0] = ((iK[0] * m[0] + iK[1] * m[1]) + iK[2]) / w;
    nm[1] = ((iK[3] * m[0] + iK[4] * m[1]) + iK[5]) / w;
}
void getCameraCenter(const double* R, const double* t, double* C) {
    for (int i = 0; i < 3; ++i) C[i] = -((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]);
}
double getAbsRadiansBetween(const double* M, const double* C0, const double* C) {
    const double a[3] = {C0[0] - M[0], C0[1] - M[1], C0[2] - M[2]}, b[3] = {C[0] - M[0], C[1] - M[1], C[2] - M[2]};
    const double d = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    const double na = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2], nb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    return fabs(acos(d / sqrt(na * nb)));
}
static double sym33Cof(const double* N, double* c) {
    c[0] = N[3] * N[5] - N[4] * N[4];
    c[1] = N[2] * N[4] - N[1] * N[5];
    c[2] = N[1] * N[4] - N[2] * N[3];
    c[3] = N[0] * N[5] - N[2] * N[2];
    c[4] = N[1] * N[2] - N[0] * N[4];
    c[5] = N[0] * N[3] - N[1] * N[1];
    return (N[0] * c[0] + N[1] * c[1]) + N[2] * c[2];
}
void triangulateMultiView(int nView, const double* Rs, const double* ts, const double* nms, double* M) {
    static const int I[6] = {0, 0, 0, 1, 1, 2}, J[6] = {0, 1, 2, 1, 2, 2};
    double N[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, c[6];
    for (int v = 0; v < nView; ++v) {
        const double* R = Rs + 9 * v;
        const double* t = ts + 3 * v;
        const double x = nms[2 * v], y = nms[2 * v + 1];
        const double a0[3] = {R[0] - x * R[6], R[1] - x * R[7], R[2] - x * R[8]}, a1[3] = {R[3] - y * R[6], R[4] - y * R[7], R[5] - y * R[8]};
        const double b0 = x * t[2] - t[0], b1 = y * t[2] - t[1];
        for (int q = 0; q < 6; ++q) N[q] = N[q] + (a0[I[q]] * a0[J[q]] + a1[I[q]] * a1[J[q]]);
        for (int q = 0; q < 3; ++q) g[q] = g[q] + (a0[q] * b0 + a1[q] * b1);
    }
    const double det = sym33Cof(N, c);
    M[0] = ((c[0] * g[0] + c[1] * g[1]) + c[2] * g[2]) / det;
    M[1] = ((c[1] * g[0] + c[3] * g[1]) + c[4] * g[2]) / det;
    M[2] = ((c[2] * g[0] + c[4] * g[1]) + c[5] * g[2]) / det;
}
void getTriangulateCovMat(int nView, const double* Ks, const double* Rs, const double* ts, const double* M, double* cov, double sigma) {
    static const int I[6] = {0, 0, 0, 1, 1, 2}, J[6] = {0, 1, 2, 1, 2, 2};
    double S[6] = {0, 0, 0, 0, 0, 0}, c[6];
    for (int v = 0; v < nView; ++v) {
        const double* K = Ks + 9 * v;
        const double* R = Rs + 9 * v;
        const double* t = ts + 3 * v;
        const double X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0];
        const double Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1];
        const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
        double KR[9], Jm[6];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) KR[3 * i + j] = (K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j]) + K[3 * i + 2] * R[6 + j];
        const double u = (K[0] * X + K[1] * Y) + K[2] * Z, w2 = (K[3] * X + K[4] * Y) + K[5] * Z, w = (K[6] * X + K[7] * Y) + K[8] * Z;
        const double ww = w * w;
        for (int j = 0; j < 3; ++j) {
            Jm[j] = (KR[j] * w - u * KR[6 + j]) / ww;
            Jm[3 + j] = (KR[3 + j] * w - w2 * KR[6 + j]) / ww;
        }
        for (int q = 0; q < 6; ++q) S[q] = S[q] + (Jm[I[q]] * Jm[J[q]] + Jm[3 + I[q]] * Jm[3 + J[q]]);
    }
    const double dS = sym33Cof(S, c), s2 = sigma * sigma;
    cov[0] = (c[0] / dS) * s2, cov[1] = (c[1] / dS) * s2, cov[2] = (c[2] / dS) * s2;
    cov[3] = cov[1], cov[4] = (c[3] / dS) * s2, cov[5] = (c[4] / dS) * s2;
    cov[6] = cov[2], cov[7] = cov[5], cov[8] = (c[5] / dS) * s2;
}
// isAtCameraBack(R, t, M) = (R M + t).z < 0 (as register.hip / register_oracle.c define it); dist3 = Euclidean distance
bool isAtCameraBack(const double* R, const double* t, const double* M) { return ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2] < 0; }
double dist3(const double* a, const double* b) {
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return sqrt((dx * dx + dy * dy) + dz * dz);
}

void project(const double* K, const double* R, const double* t, const double* M, double* m);   // shim_impl.cpp
double reprojErrorSingle(const double* K, const double* R, const double* t, const double* M, const double* m) {
    double rm[2];
    project(K, R, t, M, rm);
    const double dx = m[0] - rm[0], dy = m[1] - rm[1];
    return sqrt(dx * dx + dy * dy);
}

// the two-view forms SingleSLAM::newMapPoints names (src/app/SL_SingleSLAM.cpp:950, :957): the multi-view definitions above over the two
// views in the order given (first the track's oldest static feature, then the current one)
void binTriangulate(const double* R1, const double* t1, const double* R2, const double* t2, const double* m1, const double* m2, double* M) {
    double Rs[18], ts[6], nms[4];
    memcpy(Rs, R1, 72), memcpy(Rs + 9, R2, 72), memcpy(ts, t1, 24), memcpy(ts + 3, t2, 24);
    nms[0] = m1[0], nms[1] = m1[1], nms[2] = m2[0], nms[3] = m2[1];
    triangulateMultiView(2, Rs, ts, nms, M);
}
void getBinTriangulateCovMat(const double* K1, const double* R1, const double* t1, const double* K2, const double* R2, const double* t2,
                             const double* M, double* cov, double sigma) {
    double Ks[18], Rs[18], ts[6];
    memcpy(Ks, K1, 72), memcpy(Ks + 9, K2, 72), memcpy(Rs, R1, 72), memcpy(Rs + 9, R2, 72), memcpy(ts, t1, 24), memcpy(ts + 3, t2, 24);
    getTriangulateCovMat(2, Ks, Rs, ts, M, cov, sigma);
}
