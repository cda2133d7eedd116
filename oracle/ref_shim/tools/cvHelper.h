/* ref_shim/tools/cvHelper.h -- stand-in (see math/SL_Matrix.h): OpenCV helpers are not on the call path. */
#ifndef REF_SHIM_CVHELPER_H
#define REF_SHIM_CVHELPER_H
#include "matching/SL_Matching.h"
#include "SL_error.h"
#include <vector>
/* the OpenCV key-point / match lists SL_InitMap.h holds as members (src/app/SL_InitMap.h:42,125-128): never touched */
struct RefShimKeyPoint {};
struct RefShimDMatch {};
typedef std::vector<RefShimKeyPoint> KpVec;
typedef std::vector<RefShimDMatch> DMatchVec;
#endif
