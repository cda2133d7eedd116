/* ref_shim/matching/SL_StereoMatcherHelper.h -- stand-in (see math/SL_Matrix.h): src/slam/SL_FeatureMatching.h includes it
 * for epipolarError (declared in geometry/SL_Geometry.h here). */
#ifndef REF_SHIM_SL_STEREOMATCHERHELPER_H
#define REF_SHIM_SL_STEREOMATCHERHELPER_H
#include <cassert>
#include "math/SL_Matrix.h"
#include "geometry/SL_Geometry.h"
#endif
