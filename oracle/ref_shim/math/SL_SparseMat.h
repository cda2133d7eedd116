/* ref_shim/math/SL_SparseMat.h -- stand-in for the un-vendored LibVisualSLAM header of the same name (see math/SL_Matrix.h).
 * TEST INFRASTRUCTURE: lets oracle/Makefile compile the reference's own src/slam/SL_GlobalPoseEstimation.cpp in place.
 * Only what that file names: a triplet list (reserve / add, evident from src/slam/SL_GlobalPoseEstimation.cpp:71-78,96) and
 * declarations for the constrained variants (computeNewCamera*2/3/4, off the path: never defined, dropped by --gc-sections). */
#ifndef REF_SHIM_SL_SPARSEMAT_H
#define REF_SHIM_SL_SPARSEMAT_H
#include <vector>
#include "math/SL_Matrix.h"
#include "math/SL_LinAlg.h"

class Triplets {
public:
    int m, n;  // rows (constraints), columns (unknowns)
    std::vector<int> ri, ci;
    std::vector<double> val;
    Triplets() : m(0), n(0) {}
    void reserve(int rows, int cols, int nnz) {
        m = rows;
        n = cols;
        ri.clear(), ci.clear(), val.clear();
        ri.reserve(nnz), ci.reserve(nnz), val.reserve(nnz);
    }
    void add(int r, int c, double v) { ri.push_back(r), ci.push_back(c), val.push_back(v); }
};
class SparseMat {
public:
    int m, n;
    SparseMat() : m(0), n(0) {}
};
void triplets2Sparse(const Triplets& T, SparseMat& A);
void tripletsSplitCol(const Triplets& T, int col, Triplets& T1, Triplets& T2);
void dense2Sparse(const Mat_d& D, SparseMat& A);
void sparseMatMul(const SparseMat& A, const SparseMat& B, SparseMat& C);
void sparseSplitCol(const SparseMat& A, int col, SparseMat& B, bool first);
void print(const Mat_d& M);
void writeMat(int m, int n, const double* A, const char* path);
#endif
