/* ref_shim/matching/SL_GuidedNCCMatcher.h -- stand-in (see math/SL_Matrix.h): the greedy matchers NewMapPtsNCC::matchBetween calls
 * (src/app/SL_NewMapPointsInterCam.cpp:295-316).  LibVisualSLAM is not vendored: the definitions this repository uses are stated in
 * coslam_amd/csrc/newpts.hip and DESIGN.md 3.6; here they are prototypes only (the drivers that compile SL_NewMapPointsInterCam.cpp
 * in place give the matches, not the score matrices). */
#ifndef REF_SHIM_SL_GUIDEDNCCMATCHER_H
#define REF_SHIM_SL_GUIDEDNCCMATCHER_H
#include "math/SL_Matrix.h"
#include "matching/SL_Matching.h"
void getDisparityMat(const Mat_d& pts1, const Mat_d& pts2, const Mat_d& seeds1, const Mat_d& seeds2, double maxDisp, Mat_d& dispMat);
int greedyGuidedNCCMatch(const Mat_d& nccMat, const Mat_d& dispMat, Matching& matches);
int greedyNCCMatch(const Mat_d& nccMat, Matching& matches);
#endif
