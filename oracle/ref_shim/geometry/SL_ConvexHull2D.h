/* ref_shim/geometry/SL_ConvexHull2D.h -- stand-in (see math/SL_Matrix.h): src/slam/SL_CoSLAMHelper.cpp includes it for
 * get2DConvexHull, which is not on the hot path; the declaration lets the file compile in place, --gc-sections drops the
 * caller.  TEST INFRASTRUCTURE. */
#ifndef REF_SHIM_SL_CONVEXHULL2D_H
#define REF_SHIM_SL_CONVEXHULL2D_H
#include "ref_not_on_path.h"
#endif
