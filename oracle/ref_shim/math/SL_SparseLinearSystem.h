/* ref_shim/math/SL_SparseLinearSystem.h -- stand-in (see math/SL_SparseMat.h).  sparseSolveLin(T, b, x): the least-squares
 * solution of the over-determined sparse system T x = b (call sites src/slam/SL_GlobalPoseEstimation.cpp:197,337: 9 or 3
 * equations per valid edge, fewer unknowns than equations).  LibVisualSLAM evidently hands this to a sparse QR
 * (SuiteSparse; version unpinned, absent); the stand-in is the dense Householder-QR least-squares solution of the same system,
 * which is the same x wherever the system has full column rank.  TEST INFRASTRUCTURE. */
#ifndef REF_SHIM_SL_SPARSELINEARSYSTEM_H
#define REF_SHIM_SL_SPARSELINEARSYSTEM_H
#include "math/SL_SparseMat.h"
void sparseSolveLin(const Triplets& T, const double* b, double* x);
void sparseSolveLin(const SparseMat& A1, const SparseMat& A2, const double* b, double* x, double* y);
#endif
