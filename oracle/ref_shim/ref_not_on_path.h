/* ref_shim/ref_not_on_path.h -- declarations only, NO definitions: external (LibVisualSLAM / OpenCV) helpers that the
 * reference's translation units name in functions which are NOT on the tracker / pose / BA call path (new-map-point
 * triangulation, image scaling, the epipolar pose variant ...).  They let src/app/SL_SingleSLAM.cpp compile in place;
 * the objects are built with -ffunction-sections and linked with --gc-sections, so the functions that would need them
 * are discarded and nothing here is ever resolved.  TEST INFRASTRUCTURE (see math/SL_Matrix.h). */
#ifndef REF_SHIM_NOT_ON_PATH_H
#define REF_SHIM_NOT_ON_PATH_H
template <class... A> void scaleDownAvg(const A&...);
template <class... A> int searchNearestPoint(const A&...);
template <class... A> bool intraCamEstimateEpi(const A&...);
template <class... A> double getCameraDistance(const A&...);
#ifdef REF_SHIM_TRIANGULATE_ON_PATH
/* src/slam/SL_CoSLAMHelper.cpp compiled in place for updateStaticPointPosition / updateDynamicPointPosition (:338-394, :455-484):
 * there these helpers ARE on the path -- real prototypes, OUR definitions in ref_triangulate_impl.cpp (un-vendored LibVisualSLAM) */
void normPoint(const double* iK, const double* m, double* nm);
void getCameraCenter(const double* R, const double* t, double* C);
void triangulateMultiView(int nView, const double* Rs, const double* ts, const double* nms, double* M);
void getTriangulateCovMat(int nView, const double* Ks, const double* Rs, const double* ts, const double* M, double* cov, double sigma);
void getInvK(const double* K, double* iK);
double getAbsRadiansBetween(const double* M, const double* C0, const double* C);
bool isAtCameraBack(const double* R, const double* t, const double* M);   /* isDynamicPoint (:283) */
double dist3(const double* a, const double* b);                           /* isDynamicPoint (:290) */
/* NewMapPtsNCC::reconstructTracks (src/app/SL_NewMapPointsInterCam.cpp:247): the pixel distance of m from the projection of M */
double reprojErrorSingle(const double* K, const double* R, const double* t, const double* M, const double* m);
/* SingleSLAM::newMapPoints (src/app/SL_SingleSLAM.cpp:950, :957): the two-view forms -- OUR definitions: triangulateMultiView /
 * getTriangulateCovMat over the two views in the order given (ref_triangulate_impl.cpp) */
void binTriangulate(const double* R1, const double* t1, const double* R2, const double* t2, const double* m1, const double* m2, double* M);
void getBinTriangulateCovMat(const double* K1, const double* R1, const double* t1, const double* K2, const double* R2, const double* t2,
                             const double* M, double* cov, double sigma);
#else
template <class... A> void getBinTriangulateCovMat(const A&...);
template <class... A> void binTriangulate(const A&...);
template <class... A> double reprojErrorSingle(const A&...);
template <class... A> bool isAtCameraBack(const A&...);
template <class... A> double dist3(const A&...);
template <class... A> void normPoint(const A&...);
template <class... A> void getCameraCenter(const A&...);
template <class... A> void triangulateMultiView(const A&...);
template <class... A> void getTriangulateCovMat(const A&...);
template <class... A> void getInvK(const A&...);
template <class... A> double getAbsRadiansBetween(const A&...);
#endif
#define CV_8UC1 0
namespace cv {
struct Size {
    Size();
    Size(int, int);
};
struct Point2d {
    Point2d(double, double);
};
struct Mat {
    unsigned char* data;
    Mat();
    Mat(int, int, int, void*);
};
void resize(Mat&, Mat&, Size);
void resize(Mat&, Mat&, Size, double, double);
void getRectSubPix(Mat&, Size, Point2d, Mat&);
}  // namespace cv
#endif
