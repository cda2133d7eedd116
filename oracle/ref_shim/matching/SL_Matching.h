/* ref_shim/matching/SL_Matching.h -- stand-in (see math/SL_Matrix.h): feature matching results are named by
 * src/app/SL_MergeCameraGroup.h only; not on the call path. */
#ifndef REF_SHIM_SL_MATCHING_H
#define REF_SHIM_SL_MATCHING_H
class Matching {
public:
    int num;
    Matching() : num(0) {}
};
#endif
