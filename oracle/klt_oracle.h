/*
 * oracle/klt_oracle.h -- CPU restatement of CoSLAM's GPU-KLT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under coslam_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported baseline.
 *
 * KLT PARITY: arithmetic pinned to the reference's shaders, sampling model and pass schedule ours.  The reference
 * (danping/CoSLAM) runs this path as Nvidia Cg fragment shaders through OpenGL FBOs and ships no tests, golden vectors
 * or fixtures; Cg / GL cannot run in this image.  The shader BODIES can: oracle/build_cgref.sh
 * compiles each .cg file where it lies under /root/reference (piped through four syntactic sed rewrites, over
 * ref_shim/cg/cg_shim.h) into oracle/_ref/libcgklt_ref.so, ref_shim/cg/cgklt_driver.cpp rasterises the passes in the
 * host's order, and this file -- in its serial summation mode -- reproduces every pass BIT FOR BIT
 * (tests/test_cgklt_cpu.py; tests/golden/cgklt_golden.npz holds the shaders' outputs).  This file restates, line by
 * line, what the shaders and their host scheduling compute; each function cites the reference file:line it follows
 * (paths relative to the reference root).  The frame logic (KLT_SequenceTracker, GL-bound host C++) and the GL texture
 * model below remain restatements.
 *
 * Numeric model (the places where OpenGL leaves bits to the hardware and we
 * had to pick; all are stated in DESIGN.md):
 *   - pyramid texels are IEEE binary16, round-to-nearest-even, subnormals kept;
 *   - all shader arithmetic is IEEE binary32, no FMA contraction, evaluated in
 *     the order written in the .cg source;
 *   - NEAREST taps that land exactly on a texel boundary (the 2x decimation,
 *     v3d_gpupyramid.cpp:407-418) resolve with floor() as the GL spec says
 *     ("centered" = 0); centered = 1 selects the geometrically centred taps;
 *   - bilinear weights are full binary32 (real texture units quantise them to
 *     8 fractional bits; not modelled).
 */
#ifndef COSLAM_KLT_ORACLE_H
#define COSLAM_KLT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OKL_MAX_LEVELS 12

/* v3d_gpuklt.h:166-176 */
typedef struct okl_tracked_feature {
    int status; /* 0 tracked, 1 new, -1 dead */
    float pos[2];
    float gain;
    int fed;
} okl_tracked_feature;

/* v3d_gpuklt.h:180-199 (same field order, same defaults via okl_config_default) */
typedef struct okl_config {
    int nIterations, nLevels, levelSkip, windowWidth;
    float trackBorderMargin, convergenceThreshold, SSD_Threshold;
    int trackWithGain;
    int minDistance;
    float minCornerness, detectBorderMargin;
} okl_config;

void okl_config_default(okl_config* c);

/* ---- pyramid layout: level l is (W>>l) x (H>>l) texels of 4 halfs (I,Ix,Iy,0),
 *      level starts aligned to 64 texels.  Returns the total texel count. */
size_t okl_pyr_layout(int W, int H, int nLevels, int64_t* off_texels);

uint16_t okl_f32_to_f16(float f);
float okl_f16_to_f32(uint16_t h);

/* v3d_gpupyramid.cpp:376-429 + pyramid_with_derivative_pass1v/1h/pass2.cg */
void okl_pyramid_build(const uint8_t* img, int W, int H, int nLevels, int centered, uint16_t* pyr);

/* bilinear fetch of (I,Ix,Iy) at normalized (s,t) on one level (GL_LINEAR, CLAMP_TO_EDGE) */
void okl_sample(const uint16_t* lvl, int Wl, int Hl, float s, float t, float out[3]);

/* klt_tracker.cg:24-132 scheduled by v3d_gpuklt.cpp:99-161; N features, in/out N x 3 floats */
void okl_set_nogain_sum_mode(int mode); /* 0: the shader's serial window sums; 1: the HIP kernel's lane / tree order */
void okl_track_nogain(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int nLevels, int levelSkip,
                      int halfWidth, int nIterShader, float margin, float convThr, float ssdThr, int N,
                      const float* featIn, float* featOut);

/* one launch of klt_tracker_with_gain.cg:42-148 on level `level` */
void okl_track_gain_pass(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int nLevels, int level, int fw,
                         int fh, int halfWidth, const float* feat0, const float* featIn, float* featOut,
                         float sqrConvThr, float ssdThr, const float validRegion[4], float lambda, float delta);

/* the same pass with the window sums in the HIP tracker's fixed order (rows serially, rows folded in a fixed tree, the
 * neighbour term added once): bit-for-bit what coslam_amd/csrc/klt_track_rows.hip computes.  hw in 1..7. */
void okl_track_gain_pass_tree(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int nLevels, int level, int fw,
                              int fh, int halfWidth, const float* feat0, const float* featIn, float* featOut,
                              float sqrConvThr, float ssdThr, const float validRegion[4], float lambda, float delta);

/* test diagnostic: per-slot minimum relative distance to a validity threshold over the gain passes that follow
 * (caller-owned N floats, preset to a large value; NULL turns it off) */
void okl_set_threshold_margin_buffer(float* perSlot);

/* 7x7 structure tensor -> min eigenvalue: klt_detector_pass1.cg, klt_detector_pass2.cg,
 * scheduled by v3d_gpuklt.cpp:457-473 */
void okl_cornerness(const uint16_t* lvl0, int W, int H, float minCornerness, float margin, float* out);
/* v3d_gpuklt.cpp:475-500 */
void okl_suppress_present(float* corner, int W, int H, int nPresent, const float* present3);
/* klt_detector_nonmax.cg x2, v3d_gpuklt.cpp:502-512 */
void okl_nonmax(float* corner, int W, int H, int minDist);
/* discriminator + histopyramid + traversal (v3d_gpuklt.cpp:514-588): survivors (value>0) in
 * HistoPyramid order (= Morton order of (x,y), x minor), at most maxOut; returns the TOTAL count */
int okl_extract(const float* corner, int W, int H, int maxOut, float* list3);

/* ---- KLT_SequenceTracker restated (v3d_gpuklt.cpp:592-889, v3d_gpuklt.h:202-263) ---- */
typedef struct okl_seq okl_seq;
okl_seq* okl_seq_create(const okl_config* cfg, int centered);
void okl_seq_destroy(okl_seq* s);
okl_seq* okl_seq_clone(const okl_seq* s); /* deep copy of the whole tracker state */
void okl_seq_allocate(okl_seq* s, int W, int H, int nLevels, int fw, int fh, int plw, int plh);
void okl_seq_detect(okl_seq* s, const uint8_t* img, int* nDetected, okl_tracked_feature* dest);
void okl_seq_detect_present(okl_seq* s, const uint8_t* img, int* nDetected, okl_tracked_feature* dest, int nPresent,
                            const float* present3);
void okl_seq_redetect(okl_seq* s, const uint8_t* img, int* nNew, okl_tracked_feature* dest);
void okl_seq_track(okl_seq* s, const uint8_t* img, int* nPresent, okl_tracked_feature* dest);
void okl_seq_feed(okl_seq* s, int npts, const float* featPts, int* trackIds, int* nFed);
void okl_seq_advance(okl_seq* s);
void okl_seq_set_border_margin(okl_seq* s, float m);
void okl_seq_set_convergence_threshold(okl_seq* s, float t);
void okl_seq_set_ssd_threshold(okl_seq* s, float t);
/* 0 (default): window sums serially as the shader writes them; 1: okl_track_gain_pass_tree for the gain tracker */
void okl_seq_set_sum_mode(okl_seq* s, int mode);
/* test access: pyramid of the frame most recently built (pyrCreator1) and the cornerness map */
const uint16_t* okl_seq_cur_pyramid(const okl_seq* s);
const float* okl_seq_cornerness(const okl_seq* s);
/* copy of the feature buffer that readFeatures() would return (N x 3) */
void okl_seq_read_features(const okl_seq* s, float* out3);

/* ---- host glue between tracker and pose solve restated (handback_oracle.c): GPUKLT::addToFeaturePoints,
 * SingleSLAM::chooseStaticFeatPts and the Ms / ms packing of SingleSLAM::poseUpdate3D ---- */
void ohb_undistort_point(const double K[9], const double kud[7], const double in[2], double out[2]);
int ohb_handback(int N, int W, int H, int frame, const okl_tracked_feature* features, const double K[9], const double kud[7],
                 const double* mapPts, const unsigned char* isStatic, int* slot2map, int* trackSpan, double* xy, int* state,
                 int nColBlk, int nRowBlk, int* selBlk, int ptsStride, double* Ms, double* ms, int* sel);

void ohb_point_features(int N, const int* state, const int* slot2map, int P, int stride, int* pointFeat);

/* ---- search step of CoSLAM's map-point registration restated (register_oracle.c): projection, projected covariance,
 * searchMahaNearestFeatPt, the candidate's own mergability term ---- */
int org_is_at_camera_back(const double R[9], const double t[3], const double M[3]);
void org_project(const double K[9], const double R[9], const double t[3], const double M[3], double m[2]);
void org_projection_cov(const double K[9], const double R[9], const double t[3], const double M[3], const double cov[9],
                        double var[4], double sigma);
int org_search_maha_nearest(int N, const double* xy, const int* state, const double m[2], const double var[4], double maxDist,
                            double* dmin);
void org_register_search(int nCams, int N, int W, int H, const double* Ks, const double* Rs, const double* ts,
                         const double* const* xy, const int* const* state, const int* const* slot2map,
                         const unsigned char* const* isDynamic, int P, const double* Ms, const double* covs,
                         const int* pointFeat, double sigmaSearch, double maxDist, double sigmaMerge, int* slot, double* m_out,
                         double* var_out, double* dist, int* flags);

int org_static_check_mergability(const double K[9], int nHist, const double* histR, const double* histT, const double* histXY, int N,
                                 int slot, int len, const double M[3], const double cov[9], double pixelVar);

void org_register_mergability_cam(const double K[9], int nHist, const double* histR, const double* histT, const double* histXY, int N,
                                  const int* trackSpan, int P, const double* Ms, const double* covs, const int* slot, int slotStride,
                                  double pixelVar, unsigned char* out);

/* ---- what a frame does with a camera's new pose: poseUpdate3D's gate + seqTriangulate loop, detectDynamicFeaturePoints
 * (poseupdate_oracle.c) ---- */
void opu_seq_triangulate(const double K[9], const double R[9], const double t[3], const double m[2], double M[3], double cov[9],
                         double sigma);
int opu_gate_camera(const double K[9], const double R[9], const double t[3], int N, const double* xy, const int* state,
                    const int* slot2map, int nMap, double* mapPts, double* mapCov, unsigned char* mapFlags, int largeErr,
                    double sigma, double* reprojErr, int* numOut);
void opu_form_emat(const double* R1, const double* t1, const double* R2, const double* t2, double* E);
void opu_get_fmat(const double* iK1, const double* iK2, const double* E, double* F);
double opu_epipolar_error(const double* F, double ax, double ay, double bx, double by);
int opu_detect_dynamic_camera(const double iK[9], int N, int H, int nHist, const double* histR, const double* histT,
                              const double* histXY, const int* state, const int* slot2map, const int* trackSpan, int nMap,
                              const unsigned char* mapFlags, int maxLen, int minLen, int minOutNum, double maxEpiErr,
                              unsigned char* isStatic);
int opu_update_new_poses_points(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR,
                                const double* histT, const double* histXY, const int* trackSpan, const unsigned char* featStatic,
                                int nMap, const int* pointFeat, const int* lastFrame, const unsigned char* isCurrent,
                                int firstKeyFrame, double* mapPts, double* mapCov, const unsigned char* mapFlags, double sigma,
                                int cmpAcos, int* chosen, int* nStat, int* nDyn);
int opu_map_points_classify(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                            const double* histXY, const int* trackSpan, unsigned char* featStatic, int* slot2map, int nMap, int* pointFeat,
                            const int* featFrame, const int* featFirst, int curFrame, double* mapPts, double* mapCov,
                            unsigned char* mapFlags, unsigned char* newPt, int* staticFrameNum, const int* firstFrame, double pixelVar,
                            int* numFalse);
int opu_map_points_classify_ref(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                                const double* histXY, const int* trackSpan, unsigned char* featStatic, int* slot2map, int nMap, int* pointFeat,
                                int* featRef, const int* segPool, int segCap, unsigned char* refStatic, int curFrame, double* mapPts,
                                double* mapCov, unsigned char* mapFlags, unsigned char* newPt, int* staticFrameNum, const int* firstFrame,
                                double pixelVar, int* numFalse);
int opu_check_unify(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                    const double* histXY, const int* trackSpan, const int* pf1, const int* pf2, const double* M1, const double* M2,
                    double sigma, int cmpAcos, double* M, double* cov);
int opu_refine_map_points(int nCams, int N, int nHist, const double* Ks, const double* iKs, const double* histR, const double* histT,
                          const double* histXY, const int* trackSpan, int nMap, const int* pointFeat, const unsigned char* select,
                          double* mapPts, double* mapCov, double sigma, int cmpAcos);

/* ---- NCC blocks and the epipolar / NCC matrices of the inter-camera matching restated (ncc_oracle.c) ---- */
int onc_block_compute(const unsigned char* img, int W, int H, double x, double y, double scale, unsigned char* I, double* abc);
double onc_match(const unsigned char* I1, const double* abc1, const unsigned char* I2, const double* abc2);
/* getNCCBlocks' block cutter (SL_NCCBlock.cpp:79-155): OpenCV's resize (INTER_LINEAR, 8-bit) and getRectSubPix (8u -> 8u) restated */
void onc_resize_dims(int W, int H, double fx, double fy, int* Wd, int* Hd);
void onc_resize_linear_u8(const unsigned char* src, int W, int H, double fx, double fy, unsigned char* dst);
void onc_get_rect_sub_pix_u8(const unsigned char* img, int W, int H, double cx, double cy, int win, unsigned char* patch);
void onc_get_ncc_blocks(const unsigned char* img, int W, int H, int n, const double* x, const double* y, double scale,
                        unsigned char* small, unsigned char* blocks, double* abc);
void onc_epi_ncc_mat(const double* F, int M, const double* x1, const double* y1, const unsigned char* blk1, const double* abc1,
                     const int* valid1, int N, const double* x2, const double* y2, const unsigned char* blk2, const double* abc2,
                     const int* valid2, double epiMax, double nccMin, double wNone, double* epiMat, double* nccMat);

/* ---- pose-graph relaxation of the non-key frames after a BA restated (posegraph_oracle.c) ---- */
void opg_rigid_from_to(const double R1[9], const double t1[3], const double R2[9], const double t2[3], double R[9], double t[3]);
void opg_approx_rotation(const double R[9], double Rnew[9]);
int opg_relax(int nNodes, int nEdges, const unsigned char* fixed, const double* nodeR, const double* nodeT, const int* id1,
              const int* id2, const double* edgeR, const double* edgeT, double* newR, double* newT);

#ifdef __cplusplus
}
#endif
#endif
