/* ref_shim: stand-in for the external sba-1.6 header (see math/SL_Matrix.h); the abandoned sba estimators
 * (src/slam/SL_IntraCamPoseEstimator.h:16-18 "abandoned") are not on the call path -- only the array-size constants
 * their class declarations use. */
#ifndef REF_SHIM_SBA_H
#define REF_SHIM_SBA_H
#define SBA_OPTSSZ 5
#define SBA_INFOSZ 10
#define SBA_INIT_MU 1e-03
#define SBA_STOP_THRESH 1e-12
#endif
