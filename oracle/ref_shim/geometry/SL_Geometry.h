/* ref_shim/geometry/SL_Geometry.h -- stand-in for LibVisualSLAM geometry helpers (see math/SL_LinAlg.h).
 * project() is on the intraCamEstimate path; the covariance / epipolar helpers are only reached by the
 * other estimators in SL_IntraCamPose.cpp and follow their textbook definitions. */
#ifndef REF_SHIM_SL_GEOMETRY_H
#define REF_SHIM_SL_GEOMETRY_H
/* m = pi(K (R M + t)) */
void project(const double* K, const double* R, const double* t, const double* M, double* m);
/* sum_i |ms_i - project(M_i)|^2 */
double reprojError2(const double* K, const double* R, const double* t, int npts, const double* Ms, const double* ms);
/* var(2x2) = J cov J^T + sigma^2 I, J = d project / d M */
void getProjectionCovMat(const double* K, const double* R, const double* t, const double* M, const double* cov,
                         double* var, double sigma);
/* (a-b)^T ivar (a-b) */
double mahaDist2(const double* a, const double* b, const double* ivar);
/* E = [t]x R with R = R2 R1^T, t = t2 - R t1 */
void formEMat(const double* R1, const double* t1, const double* R2, const double* t2, double* E);
/* F = invK2^T E invK1 */
void getFMat(const double* invK1, const double* invK2, const double* E, double* F);
/* distance of m2 to the epipolar line F m1 */
double epipolarError(const double* F, const double* m2, const double* m1);
void computeEpipolarLine(const double* F, double x, double y, double* l);
/* declared for src/slam/SL_SLAMHelper.cpp's RANSAC pose helpers (off every driver's path: no definition, dropped by --gc-sections) */
void project(const double* K, const double* R, const double* t, int npts, const double* Ms, double* ms);
#endif
