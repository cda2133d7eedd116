/* ref_shim/tools/cvHelper.h -- stand-in (see math/SL_Matrix.h): OpenCV helpers are not on the call path. */
#ifndef REF_SHIM_CVHELPER_H
#define REF_SHIM_CVHELPER_H
#include "matching/SL_Matching.h"
#include "SL_error.h"
#endif
