// include/shim/geometry/SL_BundleAdjust.h -- bundleAdjustRobust with the call signature CoSLAM uses
// (src/app/SL_CoSLAMRobustBA.cpp:174, SL_InterCamPoseEstimator.cpp:95, SL_MergeCameraGroup.cpp:646-647):
//
//   bundleAdjustRobust(int nCamsCon, vector<Mat_d>& Ks, vector<Mat_d>& Rs, vector<Mat_d>& Ts, int nPtsCon,
//                      vector<Point3d>& pt3Ds, vector<vector<Meas2D>>& meas2Ds, double maxErr, int maxIter,
//                      int innerMaxIter)
//
// The LibVisualSLAM types are not redefined here: the function is a template that binds to them by the members
// CoSLAM itself uses (Mat_d::data row-major doubles, Point3d::M[3], Meas2D::{viewId,x,y,outlier};
// SL_CoSLAMRobustBA.cpp:90-92,128,153-154,283-303), so it works with the real headers and with stand-ins.
#ifndef COSLAM_SHIM_SL_BUNDLEADJUST_H
#define COSLAM_SHIM_SL_BUNDLEADJUST_H

#include <stdexcept>
#include <string>
#include <vector>

#include "coslam_hip.h"

#ifndef COSLAM_HIP_DEVICE
#define COSLAM_HIP_DEVICE 0
#endif

template <class MatD, class Pt3, class Meas>
inline void bundleAdjustRobust(int nCamsCon, std::vector<MatD>& Ks, std::vector<MatD>& Rs, std::vector<MatD>& Ts,
                               int nPtsCon, std::vector<Pt3>& pt3Ds, std::vector<std::vector<Meas> >& meas2Ds,
                               double maxErr, int maxIter = 5, int innerMaxIter = 10) {
    const int C = (int)Rs.size(), P = (int)pt3Ds.size();
    std::vector<double> K(9 * (size_t)C), R(9 * (size_t)C), T(3 * (size_t)C), M(3 * (size_t)(P > 0 ? P : 1));
    for (int j = 0; j < C; ++j) {
        for (int q = 0; q < 9; ++q) {
            K[9 * j + q] = Ks[j].data[q];
            R[9 * j + q] = Rs[j].data[q];
        }
        for (int q = 0; q < 3; ++q) T[3 * j + q] = Ts[j].data[q];
    }
    std::vector<int> ptr(P + 1, 0), cam;
    std::vector<double> xy;
    for (int i = 0; i < P; ++i) {
        for (int q = 0; q < 3; ++q) M[3 * i + q] = pt3Ds[i].M[q];
        for (size_t m = 0; m < meas2Ds[i].size(); ++m) {
            cam.push_back(meas2Ds[i][m].viewId);
            xy.push_back(meas2Ds[i][m].x);
            xy.push_back(meas2Ds[i][m].y);
        }
        ptr[i + 1] = (int)cam.size();
    }
    const int nObs = (int)cam.size();
    std::vector<int> outlier(nObs > 0 ? nObs : 1, 0);
    int rc = cs_ba_robust(C, P, nObs, K.data(), R.data(), T.data(), M.data(), ptr.data(), cam.data(), xy.data(), nCamsCon,
                          nPtsCon, maxErr, maxIter, innerMaxIter, outlier.data(), 0, COSLAM_HIP_DEVICE);
    if (rc != CS_OK) throw std::runtime_error(std::string("bundleAdjustRobust: ") + cs_last_error());
    for (int j = 0; j < C; ++j) {
        for (int q = 0; q < 9; ++q) Rs[j].data[q] = R[9 * j + q];
        for (int q = 0; q < 3; ++q) Ts[j].data[q] = T[3 * j + q];
    }
    for (int i = 0, o = 0; i < P; ++i) {
        for (int q = 0; q < 3; ++q) pt3Ds[i].M[q] = M[3 * i + q];
        for (size_t m = 0; m < meas2Ds[i].size(); ++m, ++o) meas2Ds[i][m].outlier = outlier[o];
    }
}

#endif
