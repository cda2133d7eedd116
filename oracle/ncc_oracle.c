/*
 * oracle/ncc_oracle.c -- CPU restatement of the NCC-block stage of CoSLAM's inter-camera matching.
 *
 * TEST INFRASTRUCTURE ONLY (see klt_oracle.h).  Follows, statement by statement:
 *   NCCBlock::compute / computeScaled   /root/reference/src/slam/SL_NCCBlock.cpp:15-54   (the 11 x 11 block at the truncated
 *                                       position, avgI, A = sum I, B = sum I^2, C = 1 / sqrt(121 B - A^2))
 *   matchNCCBlock                       /root/reference/src/slam/SL_NCCBlock.cpp:258-264
 *   getEpiNccMat                        /root/reference/src/slam/SL_FeatureMatching.cpp:3-46 (the M x N epipolar-error and NCC
 *                                       matrices NewMapPtsNCC::matchBetween, src/app/SL_NewMapPointsInterCam.cpp:273-317,
 *                                       hands to the greedy matcher)
 * All three are in the reference tree and pinned: tests/cxx/ref_ncc_test.cpp compiles them in place and
 * tests/golden/ncc_golden.npz holds their outputs.
 *
 * PARITY UNPINNED for epipolarError (un-vendored LibVisualSLAM; only its calls are in the reference).  Definition used here,
 * in oracle/ref_shim/shim_impl.cpp and in coslam_amd/csrc/ncc.hip: epipolarError(F, a, b) = distance of a from the line
 * l = F (b, 1): |l0 a.x + l1 a.y + l2| / sqrt(l0^2 + l1^2).  How the blocks of matchBetween are cut (cv::getRectSubPix on a
 * cv::resize'd image, SL_NCCBlock.cpp:111-155) is OpenCV and not restated: the in-tree NCCBlock::compute is.
 */
#include <math.h>
#include <string.h>

#include "klt_oracle.h"

#define ONC_HW 5
#define ONC_LEN 121

/* NCCBlock::computeScaled(img, scale, x0, y0) = compute(img, x0 * scale, y0 * scale), SL_NCCBlock.cpp:51-54,15-49.
 * I: 121 bytes out; abc: A, B, C, avgI.  Returns 1 (true) or 0 (the block would leave the image: nothing is written). */
int onc_block_compute(const unsigned char* img, int W, int H, double x, double y, double scale, unsigned char* I, double* abc) {
    const double xs = x * scale, ys = y * scale;
    const int bw = 2 * ONC_HW + 1;
    const int x0 = (int)xs, y0 = (int)ys; /* :20-21 */
    if (x0 - ONC_HW < 0 || x0 + ONC_HW >= W || y0 - ONC_HW < 0 || y0 + ONC_HW >= H) return 0; /* :24-26 */
    double avgI = 0;
    for (int j = 0; j < ONC_LEN; ++j) { /* :29-39 */
        const int yy = j / bw, xx = j - yy * bw;
        I[j] = img[(size_t)(y0 + yy - ONC_HW) * W + (x0 + xx - ONC_HW)];
        avgI += I[j];
    }
    avgI /= ONC_LEN; /* :40 */
    double a = 0, b = 0;
    for (int j = 0; j < ONC_LEN; ++j) { /* :42-46 */
        a += I[j];
        b += (double)I[j] * I[j];
    }
    abc[0] = a;
    abc[1] = b;
    abc[2] = 1 / sqrt(ONC_LEN * b - a * a); /* :49 */
    abc[3] = avgI;
    return 1;
}

/* matchNCCBlock, SL_NCCBlock.cpp:258-264 */
double onc_match(const unsigned char* I1, const double* abc1, const unsigned char* I2, const double* abc2) {
    double d = 0;
    for (int i = 0; i < ONC_LEN; i++) d += (double)I1[i] * I2[i];
    return (ONC_LEN * d - abc1[0] * abc2[0]) * abc1[2] * abc2[2];
}

static double epipolar_error(const double* F, double ax, double ay, double bx, double by) {
    const double l0 = (F[0] * bx + F[1] * by) + F[2];
    const double l1 = (F[3] * bx + F[4] * by) + F[5];
    const double l2 = (F[6] * bx + F[7] * by) + F[8];
    const double n = sqrt(l0 * l0 + l1 * l1);
    return fabs((l0 * ax + l1 * ay) + l2) / (n > 0 ? n : 1.0);
}

/* getEpiNccMat, SL_FeatureMatching.cpp:3-46.  Points as x[], y[]; blocks as n x 128 bytes (121 used), abc n x 4; valid: the
 * block exists (compute returned true) -- a pair with a missing block is reported as wNone. */
void onc_epi_ncc_mat(const double* F, int M, const double* x1, const double* y1, const unsigned char* blk1, const double* abc1,
                     const int* valid1, int N, const double* x2, const double* y2, const unsigned char* blk2, const double* abc2,
                     const int* valid2, double epiMax, double nccMin, double wNone, double* epiMat, double* nccMat) {
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            const size_t o = (size_t)i * N + j;
            epiMat[o] = wNone;
            nccMat[o] = wNone;
            const double epiErr = epipolar_error(F, x1[i], y1[i], x2[j], y2[j]); /* :24-25 */
            if (epiErr <= epiMax && valid1[i] && valid2[j]) {                    /* :26 */
                const double ncc = onc_match(blk1 + 128 * (size_t)i, abc1 + 4 * (size_t)i, blk2 + 128 * (size_t)j, abc2 + 4 * (size_t)j);
                if (ncc >= nccMin) { /* :29-31 */
                    epiMat[o] = epiErr;
                    nccMat[o] = ncc;
                }
            }
        }
}
