/* ref_shim/geometry/SL_Triangulate.h -- SL_IntraCamPose.cpp includes it but calls nothing from it. */
#ifndef REF_SHIM_SL_TRIANGULATE_H
#define REF_SHIM_SL_TRIANGULATE_H
/* sequential (Kalman) refinement of a map point and its covariance from one more measurement
 * (src/app/SL_InterCamPoseEstimator.cpp:119, src/app/SL_SingleSLAM.cpp:690) */
void seqTriangulate(const double* K, const double* R, const double* t, const double* m, double* M, double* cov, double sigma);
#endif
