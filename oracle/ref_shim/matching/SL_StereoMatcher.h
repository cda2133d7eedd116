/* ref_shim/matching/SL_StereoMatcher.h -- stand-in (see math/SL_Matrix.h): named by src/app/SL_NewMapPointsInterCam.cpp's include list only;
 * nothing of it is used by the NCC path. */
#ifndef REF_SHIM_SL_StereoMatcher_H
#define REF_SHIM_SL_StereoMatcher_H
#endif
