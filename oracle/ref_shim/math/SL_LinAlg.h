/* ref_shim/math/SL_LinAlg.h -- stand-in for the un-vendored LibVisualSLAM header of the same name
 * (reference CMakeLists.txt:7, cmake/Modules/FindVisualSLAM.cmake:1-18; version unpinned).
 * TEST INFRASTRUCTURE: lets oracle/Makefile compile the reference's own src/slam/SL_IntraCamPose.cpp
 * in place.  Semantics are those evident from the call sites in that file (row-major doubles).
 * On the intraCamEstimate path only doubleArrCopy, mat33AB, matATB, matAB, matInv are exercised. */
#ifndef REF_SHIM_SL_LINALG_H
#define REF_SHIM_SL_LINALG_H
#include <cstring>
#include <cmath>
#include "ref_not_on_path.h"

/* dst[off*n .. off*n+n) = src[0..n)  (always called with off == 0 in SL_IntraCamPose.cpp) */
void doubleArrCopy(double* dst, int off, const double* src, int n);
/* C(3x3) = A(3x3) B(3x3) */
void mat33AB(const double* A, const double* B, double* C);
/* C(n x q) = A(m x n)^T B(p x q), m == p */
void matATB(int m, int n, int p, int q, const double* A, const double* B, double* C);
/* C(m x q) = A(m x n) B(p x q), n == p */
void matAB(int m, int n, int p, int q, const double* A, const double* B, double* C);
/* invA = A^-1, n x n (LibVisualSLAM: LAPACK dgetrf/dgetri; here LU with partial pivoting) */
void matInv(int n, const double* A, double* invA);
/* adjugate / determinant (the closed form; searchMahaNearestFeatPt, src/app/SL_SingleSLAM.cpp:1148) */
void mat22Inv(const double* A, double* invA);
/* B(m x n) = s A(m x n) (src/app/SL_SingleSLAM.cpp:1149: matScale(2, 2, ivar, 1 / maxDist, ivar)) */
void matScale(int m, int n, const double* A, double s, double* B);
void mat33Inv(const double* A, double* invA);
void mat33Trans(const double* A, double* At);
/* Euclidean distance of two 2-vectors (src/app/SL_SingleSLAM.cpp:658: dist2(m, fp->m)) */
double dist2(const double* a, const double* b);
/* declared for src/slam/SL_SLAMHelper.cpp (solvePnPRansac, getCameraCenterAxes: off every driver's path, no definition) */
void randChoose(int n, int* idx, int k);
void mat33TransProdVec(const double* A, const double* v, double* r);
#endif
