/* ref_shim/tools/SL_Tictoc.h -- stand-in (see math/SL_Matrix.h): the timer CoSLAM's stages wrap themselves in
 * (src/app/SL_CoSLAM.cpp:120-132); nothing compiled for the tests reads the times. */
#ifndef REF_SHIM_SL_TICTOC_H
#define REF_SHIM_SL_TICTOC_H
#include <cfloat>
#include <limits>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
class TimeMeasurer {
public:
    void tic() {}
    double toc() { return 0.0; }
};
#endif
