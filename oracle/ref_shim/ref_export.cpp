// ref_shim/ref_export.cpp -- extern "C" doorway into the reference's own intraCamEstimate
// (src/slam/SL_IntraCamPose.cpp:626-709), compiled in place by oracle/Makefile into oracle/_ref/.
#include "SL_IntraCamPose.h"

extern "C" int ref_intraCamEstimate(const double* K, const double* R0, const double* t0, int npts,
                                    const double* prevErrs, const double* Ms, const double* ms, double tau,
                                    double* R_opt, double* t_opt, double* stats /* [8] */) {
    IntraCamPoseOption opt;
    bool ok = intraCamEstimate(K, R0, t0, npts, prevErrs, Ms, ms, tau, R_opt, t_opt, &opt);
    if (stats) {
        stats[0] = opt.err;
        stats[1] = opt.errRW;
        stats[2] = opt.lambda;
        stats[3] = (double)opt.nIterLM;
        stats[4] = (double)opt.nIterRW;
        stats[5] = (double)opt.retTypeLM;
        stats[6] = opt.err0;
        stats[7] = opt.lambda0;
    }
    return ok ? 1 : 0;
}

extern "C" void ref_getSO3ExpMap(const double* w, double* R) { getSO3ExpMap(w, R); }
