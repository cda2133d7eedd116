/* ref_shim/math/SL_LinAlgWarper.h -- stand-in (see math/SL_Matrix.h): declarations only, the callers
 * (GlobalPoseGraph::computeNewCamera*2/3, src/slam/SL_GlobalPoseEstimation.cpp:769-791) are off the path and dropped by
 * --gc-sections. */
#ifndef REF_SHIM_SL_LINALGWARPER_H
#define REF_SHIM_SL_LINALGWARPER_H
#include "math/SL_Matrix.h"
void matTrans(const Mat_d& A, Mat_d& At);
void matQR(const Mat_d& A, Mat_d& Q, Mat_d& R);
void matAx(int m, int n, const double* A, const double* x, double* y);
#endif
