/* ref_shim/geometry/SL_RigidTransform.h -- stand-in (see math/SL_Matrix.h); only declarations the compiled files name. */
#ifndef REF_SHIM_SL_RIGIDTRANSFORM_H
#define REF_SHIM_SL_RIGIDTRANSFORM_H
/* inverse of the rigid transform (R, t): iR = R^T, it = -R^T t (src/app/SL_MergeCameraGroup.h:77-80) */
void invRigidTransFromTo(const double* R, const double* t, double* iR, double* it);
/* relative transform that takes camera-1 coordinates to camera-2 coordinates: R = R2 R1^T, t = t2 - R t1
 * (src/app/SL_CoSLAMRobustBA.cpp:225) */
void getRigidTransFromTo(const double* R1, const double* t1, const double* R2, const double* t2, double* R, double* t);
/* nearest orthogonal matrix in the Frobenius norm, U V^T of R = U S V^T (src/slam/SL_GlobalPoseEstimation.cpp:210: turns the
 * least-squares 3x3 blocks back into rotations) */
void approxRotationMat(const double* R, double* Rnew);
/* y = a R x + b t (src/slam/SL_GlobalPoseEstimation.cpp:330, commented `T_j = T_{ij} + R_{ij} T_i`) */
void mat33ProdVec(const double* R, const double* x, const double* t, double* y, double a, double b);
#endif
