/* ref_shim/calibration/SL_CalibTwoCam.h -- stand-in (see math/SL_Matrix.h): declarations only; the two-view initialisation
 * (CoSLAM::initMapSingleCam, src/app/SL_CoSLAM.cpp:140-230) is off the call path and dropped by --gc-sections. */
#ifndef REF_SHIM_SL_CALIBTWOCAM_H
#define REF_SHIM_SL_CALIBTWOCAM_H
#include <vector>
#include "math/SL_Matrix.h"
class CalibTwoCam {
public:
    void setIntrinParam(const double* K1, const double* K2);
    void setDistorParam(int W, int H, const Mat_d& kc1, const Mat_d& kc2);
    void setMatchedPoints(const Mat_d& p1, const Mat_d& p2);
    void estimateEMat(double thres = 2.0);
    void getInlierInd(std::vector<int>& ind);
    void outputInlierNormPoints(Mat_d& p1, Mat_d& p2);
    void outputRTs(Mat_d& R1, Mat_d& t1, Mat_d& R2, Mat_d& t2);
};
void binTriangulatePoints(const double* R1, const double* t1, const double* R2, const double* t2, int npts, const double* pts1,
                          const double* pts2, double* Ms);
void readIntrinDistParam(const char* path, Mat_d& K, Mat_d& kc);
void invDistorParam(int W, int H, const double* iK, const double* kc, double* kud);
#endif
