/* oracle/ba_oracle.h -- CPU definition of bundleAdjustRobust (TEST INFRASTRUCTURE, PARITY UNPINNED; see ba_oracle.c) */
#ifndef COSLAM_BA_ORACLE_H
#define COSLAM_BA_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oba_stats {
    double cost0, cost; /* sum of squared inlier residuals before / after */
    int nIterTotal, nOuter, nOutliers, pad;
} oba_stats;

/* Flat form of bundleAdjustRobust(nCamsCon,Ks,Rs,Ts,nPtsCon,pts,meas,maxErr,maxIter,innerMaxIter)
 * (call sites: reference src/app/SL_CoSLAMRobustBA.cpp:174, SL_InterCamPoseEstimator.cpp:95).
 * Measurements are grouped by point (CSR): point i owns obs_ptr[i]..obs_ptr[i+1]; obs_cam = Meas2D::viewId,
 * obs_xy = (Meas2D::x, y).  Rs (C x 9), Ts (C x 3), pts (P x 3) are updated in place; outlier[nObs] = Meas2D::outlier. */
int oba_robust(int C, int P, int nObs, const double* Ks, double* Rs, double* Ts, double* pts, const int* obs_ptr,
               const int* obs_cam, const double* obs_xy, int nCamsCon, int nPtsCon, double maxErr, int maxIter,
               int innerMaxIter, int* outlier, oba_stats* st);
int oba_residual(const double* K, const double* R, const double* t, const double* M, const double* m, double* e,
                 double* Jc, double* Jp);
int oba_cholesky_solve(int n, double* S, double* b);
#ifdef __cplusplus
}
#endif
#endif
