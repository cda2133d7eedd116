// ref_shim/ref_posegraph_impl.cpp -- the LibVisualSLAM helpers the reference's own src/slam/SL_GlobalPoseEstimation.cpp calls
// on the computeNewCameraRotations / computeNewCameraTranslations path (lines 52-359), restated from their call sites and the
// published algorithms they name (LibVisualSLAM itself is absent: CMakeLists.txt:7, version unpinned):
//   sparseSolveLin(T, b, x)   least-squares solution of the over-determined system -- dense Householder QR here
//   approxRotationMat(R, Rn)  U V^T of the SVD of R -- one-sided Jacobi SVD here
//   mat33ProdVec              y = a R x + b t
// TEST INFRASTRUCTURE (see math/SL_Matrix.h): linked only into oracle/_ref/ binaries.
#include <cmath>
#include <cstring>
#include <vector>

#include "geometry/SL_RigidTransform.h"
#include "math/SL_SparseLinearSystem.h"

void sparseSolveLin(const Triplets& T, const double* b, double* x) {
    const int m = T.m, n = T.n;
    std::vector<double> A((size_t)m * n, 0.0), rhs(b, b + m);
    for (size_t k = 0; k < T.val.size(); ++k) A[(size_t)T.ri[k] * n + T.ci[k]] += T.val[k];
    std::vector<double> v(m);
    for (int j = 0; j < n; ++j) {  // Householder reflection that zeroes column j below the diagonal
        double nrm = 0;
        for (int i = j; i < m; ++i) nrm += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        nrm = sqrt(nrm);
        if (nrm == 0) continue;
        const double ajj = A[(size_t)j * n + j], alpha = ajj > 0 ? -nrm : nrm;
        for (int i = j; i < m; ++i) v[i] = A[(size_t)i * n + j];
        v[j] -= alpha;
        double vtv = 0;
        for (int i = j; i < m; ++i) vtv += v[i] * v[i];
        if (vtv == 0) continue;
        for (int c = j; c < n; ++c) {
            double s = 0;
            for (int i = j; i < m; ++i) s += v[i] * A[(size_t)i * n + c];
            s = 2 * s / vtv;
            for (int i = j; i < m; ++i) A[(size_t)i * n + c] -= s * v[i];
        }
        double s = 0;
        for (int i = j; i < m; ++i) s += v[i] * rhs[i];
        s = 2 * s / vtv;
        for (int i = j; i < m; ++i) rhs[i] -= s * v[i];
    }
    for (int j = n - 1; j >= 0; --j) {  // R x = Q^T b
        double s = rhs[j];
        for (int c = j + 1; c < n; ++c) s -= A[(size_t)j * n + c] * x[c];
        x[j] = s / A[(size_t)j * n + j];
    }
}

void approxRotationMat(const double* R, double* Rnew) {
    // one-sided Jacobi: rotate column pairs of W = R V until they are orthogonal; then W = U S, and U V^T is the answer
    double W[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(W, R, sizeof(W));
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, bq = 0, g = 0;
                for (int i = 0; i < 3; ++i) a += W[3 * i + p] * W[3 * i + p], bq += W[3 * i + q] * W[3 * i + q], g += W[3 * i + p] * W[3 * i + q];
                if (fabs(g) <= 1e-300 || fabs(g) <= 1e-17 * sqrt(a * bq)) continue;
                off = fmax(off, fabs(g) / sqrt(a * bq));
                const double zeta = (bq - a) / (2 * g), t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                const double c = 1 / sqrt(1 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double wp = W[3 * i + p], wq = W[3 * i + q];
                    W[3 * i + p] = c * wp - s * wq, W[3 * i + q] = s * wp + c * wq;
                    const double vp = V[3 * i + p], vq = V[3 * i + q];
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
        for (int c = 0; c < 3; ++c) Rnew[3 * r + c] = U[3 * r] * V[3 * c] + U[3 * r + 1] * V[3 * c + 1] + U[3 * r + 2] * V[3 * c + 2];
}

void mat33ProdVec(const double* R, const double* x, const double* t, double* y, double a, double b) {
    for (int r = 0; r < 3; ++r) y[r] = a * (R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2]) + b * t[r];
}
