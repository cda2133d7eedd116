// include/shim/slam/SL_IntraCamPose.h -- header-compatible replacement of the reference's
// src/slam/SL_IntraCamPose.h for the function on the hot path: the same IntraCamPoseOption class (:19-57) and the
// same bool intraCamEstimate(...) signature (:92-95), running on an MI355X through libcoslam_hip.so.
// The epipolar / covariance variants of the reference file are not on the per-frame path and are not provided.
#ifndef SL_INTRACAMPOSE_H_
#define SL_INTRACAMPOSE_H_

#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "coslam_hip.h"

class IntraCamPoseOption {
public:
    int maxIterLM;
    int maxIterRW;
    double epsErrorChangeLM;
    double epsParamChangeLM;
    double epsErrorChangeRW;
    int verboseLM;
    int verboseRW;

public:
    double lambda0;
    double lambda;

    double err0;
    double err;
    double errRW;

    int retTypeLM;
    int npts;

    int nIterLM;
    int nIterRW;

public:
    IntraCamPoseOption()
        : maxIterLM(100), maxIterRW(5), epsErrorChangeLM(1e-7), epsParamChangeLM(1e-6), epsErrorChangeRW(1e-6),
          verboseLM(0), verboseRW(0), lambda0(1e-3), lambda(0), err0(0), err(0), errRW(0), retTypeLM(0), npts(0),
          nIterLM(0), nIterRW(0) {}
    void printLM() {
        printf("lambda:%lf -> %lf\n", lambda0, lambda);
        printf("ssd: %lf -> %lf\n", err0, err);
        printf("err: %lf -> %lf\n", sqrt(err0 / npts), sqrt(err / npts));
        printf("npts:%d\n", npts);
        printf("return type:%d\n", retTypeLM);
        printf("number of LM interation:%d\n", nIterLM);
    }
    void printRW() {}
};

static_assert(sizeof(IntraCamPoseOption) == sizeof(cs_pose_option), "IntraCamPoseOption layout");

#ifndef COSLAM_HIP_DEVICE
#define COSLAM_HIP_DEVICE 0
#endif

/**
 * intra-camera pose estimation using M-estimator (Tukey's estimator)
 * R0, t0 : initial camera pose
 * errs : reprojection error from the previous frame
 */
inline bool intraCamEstimate(const double K[9], const double R0[9], const double t0[3], int npts,
                             const double prevErrs[], const double Ms[], const double ms[], const double tau,
                             double R_opt[9], double t_opt[3], IntraCamPoseOption* opt) {
    int rc = cs_pose_intracam(K, R0, t0, npts, prevErrs, Ms, ms, tau, R_opt, t_opt,
                              reinterpret_cast<cs_pose_option*>(opt), COSLAM_HIP_DEVICE);
    if (rc < 0) throw std::runtime_error(std::string("intraCamEstimate: ") + cs_last_error());
    return rc == 1;
}

#endif /* SL_INTRACAMPOSE_H_ */
