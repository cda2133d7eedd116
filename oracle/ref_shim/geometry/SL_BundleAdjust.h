/* ref_shim/geometry/SL_BundleAdjust.h -- what the reference's callers see when they include the external
 * "geometry/SL_BundleAdjust.h": the measurement type they construct (Meas2D(viewId, x, y), read back through .outlier:
 * src/app/SL_CoSLAMRobustBA.cpp:153-154,298-306, src/app/SL_InterCamPoseEstimator.cpp:46-49) as a stand-in for
 * LibVisualSLAM's, and then THE PRODUCT'S header-compatible bundleAdjustRobust (include/shim/geometry/SL_BundleAdjust.h)
 * over libcoslam_hip.so.  TEST INFRASTRUCTURE (see math/SL_Matrix.h). */
#ifndef REF_SHIM_SL_BUNDLEADJUST_H
#define REF_SHIM_SL_BUNDLEADJUST_H
#include "geometry/SL_Point.h"
#include "math/SL_Matrix.h"
class Meas2D {
public:
    int viewId;
    double x, y;
    int outlier;
    Meas2D() : viewId(-1), x(0), y(0), outlier(0) {}
    Meas2D(int v, double a, double b) : viewId(v), x(a), y(b), outlier(0) {}
};
#include "../../../include/shim/geometry/SL_BundleAdjust.h"
#endif
