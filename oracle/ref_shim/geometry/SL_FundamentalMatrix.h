/* ref_shim/geometry/SL_FundamentalMatrix.h -- included by SL_IntraCamPose.cpp:800; see SL_5point.h. */
#ifndef REF_SHIM_SL_FUNDAMENTALMATRIX_H
#define REF_SHIM_SL_FUNDAMENTALMATRIX_H
#include "geometry/SL_Geometry.h"
#endif
