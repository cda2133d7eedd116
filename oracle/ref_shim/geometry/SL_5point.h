/* ref_shim/geometry/SL_5point.h -- included by SL_IntraCamPose.cpp:799; the essential-matrix helpers it
 * uses (formEMat, getFMat, epipolarError, computeEpipolarLine) are declared in SL_Geometry.h. */
#ifndef REF_SHIM_SL_5POINT_H
#define REF_SHIM_SL_5POINT_H
#include "geometry/SL_Geometry.h"
#endif
