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

/* ------------------------------------------------------------------------------------------------------------------------
 * How NewMapPtsNCC::matchBetween really cuts its blocks: getNCCBlocks (SL_NCCBlock.cpp:79-180, called at
 * src/app/SL_NewMapPointsInterCam.cpp:280-282 with the camera's FULL image and blockScale = 0.3) =
 *     cv::resize(img, small, cv::Size(), scale, scale)        [default interpolation INTER_LINEAR, CV_8UC1]       (:125-127)
 *     per point: cv::getRectSubPix(small, 11 x 11, cv::Point2d(x * scale, y * scale), patch)                       (:131-134)
 *     A = sum I, B = sum I^2, C = 1 / sqrt(121 B - A^2), avgI                                                      (:137-151)
 * (scale == 1.0: getRectSubPix on the image itself, :93-121).
 *
 * THIRD-PARTY, ABSENT FROM THIS IMAGE, VERSION UNPINNED (CMakeLists.txt: find_package(OpenCV REQUIRED), no version): OpenCV.
 * PARITY UNPINNED for the two functions below.  They restate the published generic (non-IPP, non-OpenCL, non-SIMD) C++ paths:
 *   cv::resize INTER_LINEAR on 8-bit: modules/imgproc/src/resize.cpp -- cv::resize's table set-up, HResizeLinear<uchar, int,
 *       short, INTER_RESIZE_COEF_SCALE = 2048>, VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>> (the same
 *       arithmetic from OpenCV 2.4 to 4.x);
 *   cv::getRectSubPix 8u -> 8u: modules/imgproc/src/samplers.cpp getRectSubPix_Cn_<uchar, uchar, int, scale_fixpt, cast_8u> +
 *       adjustRect (OpenCV 3.x / 4.x: 16-bit fixed-point weights, replicated border; 2.4's C path used float weights and
 *       cvRound and can differ by one grey level).
 * Known-answer tests in tests/test_oracle_cpu.py (integer centres copy pixels, half-pixel centres average four with the
 * fixed-point rounding, centres outside replicate the border, constant and ramp images under resize).
 * ------------------------------------------------------------------------------------------------------------------------ */
#include <stdint.h>
#include <stdlib.h>

static int onc_cv_round(double v) { return (int)lrint(v); }    /* cvRound: round half to even */
static int onc_cv_roundf(float v) { return (int)lrintf(v); }
static int onc_cv_floorf(float v) {                              /* cvFloor */
    int i = (int)v;
    return i - (i > v);
}
static int onc_clip(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; } /* resize.cpp clip() */
static short onc_sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* Size dsize(saturate_cast<int>(ssize.width * inv_scale_x), saturate_cast<int>(ssize.height * inv_scale_y)) */
void onc_resize_dims(int W, int H, double fx, double fy, int* Wd, int* Hd) {
    *Wd = onc_cv_round(W * fx);
    *Hd = onc_cv_round(H * fy);
}

void onc_resize_linear_u8(const unsigned char* src, int W, int H, double fx, double fy, unsigned char* dst) {
    int Wd, Hd;
    onc_resize_dims(W, H, fx, fy, &Wd, &Hd);
    const double scale_x = 1. / fx, scale_y = 1. / fy;
    int* xofs = (int*)malloc(sizeof(int) * Wd);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * Wd);
    int xmax = Wd;
    for (int dx = 0; dx < Wd; dx++) {
        float f = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = onc_cv_floorf(f);
        f -= sx;
        if (sx < 0) f = 0, sx = 0; /* ksize2 - 1 = 0 */
        if (sx + 1 >= W) {
            xmax = xmax < dx ? xmax : dx;
            if (sx >= W - 1) f = 0, sx = W - 1;
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = onc_sat_short(onc_cv_roundf((1.f - f) * 2048.f));
        ialpha[2 * dx + 1] = onc_sat_short(onc_cv_roundf(f * 2048.f));
    }
    int* row0 = (int*)malloc(sizeof(int) * Wd);
    int* row1 = (int*)malloc(sizeof(int) * Wd);
    for (int dy = 0; dy < Hd; dy++) {
        float f = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = onc_cv_floorf(f);
        f -= sy;
        const short b0 = onc_sat_short(onc_cv_roundf((1.f - f) * 2048.f)), b1 = onc_sat_short(onc_cv_roundf(f * 2048.f));
        const unsigned char* S0 = src + (size_t)onc_clip(sy, 0, H) * W;
        const unsigned char* S1 = src + (size_t)onc_clip(sy + 1, 0, H) * W;
        for (int dx = 0; dx < Wd; dx++) { /* HResizeLinear */
            const int sx = xofs[dx];
            if (dx < xmax) {
                row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx + 1] * ialpha[2 * dx + 1];
                row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx + 1] * ialpha[2 * dx + 1];
            } else {
                row0[dx] = S0[sx] * 2048;
                row1[dx] = S1[sx] * 2048;
            }
        }
        unsigned char* D = dst + (size_t)dy * Wd;
        for (int x = 0; x < Wd; x++) /* VResizeLinear, uchar */
            D[x] = (unsigned char)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs);
    free(ialpha);
    free(row0);
    free(row1);
}

/* cv::getRectSubPix(img (W x H, 8UC1), Size(win, win), Point2f((float)cx, (float)cy), patch 8UC1), row-major win x win out.
 * Pointer arithmetic of the original kept as (row, column) indices. */
void onc_get_rect_sub_pix_u8(const unsigned char* img, int W, int H, double cx, double cy, int win, unsigned char* patch) {
    float fxc = (float)cx, fyc = (float)cy; /* Point2d -> Point2f */
    fxc -= (win - 1) * 0.5f;
    fyc -= (win - 1) * 0.5f;
    const int ipx = onc_cv_floorf(fxc), ipy = onc_cv_floorf(fyc);
    const float a = fxc - ipx, b = fyc - ipy;
#define ONC_FIX(v) onc_cv_roundf((v) * 65536.f)               /* scale_fixpt */
#define ONC_CAST(t) ((unsigned char)(((t) + (1 << 15)) >> 16)) /* cast_8u */
    const int a11 = ONC_FIX((1.f - a) * (1.f - b)), a12 = ONC_FIX(a * (1.f - b)), a21 = ONC_FIX((1.f - a) * b), a22 = ONC_FIX(a * b);
    const int b1 = ONC_FIX(1.f - b), b2 = ONC_FIX(b);
    if (0 <= ipx && ipx < W - win && 0 <= ipy && ipy < H - win) { /* totally inside */
        for (int i = 0; i < win; i++) {
            const unsigned char* s = img + (size_t)(ipy + i) * W + ipx;
            for (int j = 0; j < win; j++) {
                const int t = s[j] * a11 + s[j + 1] * a12 + s[j + W] * a21 + s[j + W + 1] * a22;
                patch[i * win + j] = ONC_CAST(t);
            }
        }
        return;
    }
    /* adjustRect */
    int rx, rw, ry, rh;
    int col0; /* image column that src[rect.x] refers to: the returned pointer is src - rect.x */
    int row = 0;
    if (ipx >= 0) {
        col0 = ipx;
        rx = 0;
    } else {
        col0 = 0;
        rx = -ipx;
        if (rx > win) rx = win;
    }
    if (ipx < W - win) {
        rw = win;
    } else {
        rw = W - ipx - 1;
        if (rw < 0) {
            col0 += rw;
            rw = 0;
        }
    }
    if (ipy >= 0) {
        row = ipy;
        ry = 0;
    } else {
        ry = -ipy;
    }
    if (ipy < H - win) {
        rh = win;
    } else {
        rh = H - ipy - 1;
        if (rh < 0) {
            row += rh;
            rh = 0;
        }
    }
    /* src[j] (j in window columns) = img[row][col0 + j - rx] */
    for (int i = 0; i < win; i++) {
        int row2 = row + 1;
        if (i < ry || i >= rh) row2 -= 1;
        const unsigned char* s = img + (size_t)row * W + (col0 - rx);
        const unsigned char* s2 = img + (size_t)row2 * W + (col0 - rx);
        int t = s[rx] * b1 + s2[rx] * b2;
        for (int j = 0; j < rx; j++) patch[i * win + j] = ONC_CAST(t);
        t = s[rw] * b1 + s2[rw] * b2;
        for (int j = rw; j < win; j++) patch[i * win + j] = ONC_CAST(t);
        for (int j = rx; j < rw; j++) {
            const int u = s[j] * a11 + s[j + 1] * a12 + s2[j] * a21 + s2[j + 1] * a22;
            patch[i * win + j] = ONC_CAST(u);
        }
        if (i < rh) row = row2;
    }
#undef ONC_FIX
#undef ONC_CAST
}

/* getNCCBlocks(img, pts, blocks, scale), SL_NCCBlock.cpp:79-155: blocks n x 128 (121 used, padding 0x80), abc n x 4
 * (A, B, C, avgI).  small: caller's scratch of onc_resize_dims bytes (unused when scale == 1.0). */
void onc_get_ncc_blocks(const unsigned char* img, int W, int H, int n, const double* x, const double* y, double scale,
                        unsigned char* small, unsigned char* blocks, double* abc) {
    const unsigned char* im = img;
    int Ws = W, Hs = H;
    if (scale != 1.0) { /* :123-127 */
        onc_resize_dims(W, H, scale, scale, &Ws, &Hs);
        onc_resize_linear_u8(img, W, H, scale, scale, small);
        im = small;
    }
    for (int i = 0; i < n; i++) {
        unsigned char* I = blocks + 128 * (size_t)i;
        memset(I, 0x80, 128);
        const double px = (scale != 1.0) ? x[i] * scale : x[i], py = (scale != 1.0) ? y[i] * scale : y[i]; /* :131-132 / :99-100 */
        onc_get_rect_sub_pix_u8(im, Ws, Hs, px, py, 2 * ONC_HW + 1, I);
        double a = 0, b = 0, avg = 0;
        for (int j = 0; j < ONC_LEN; ++j) { /* :142-150 */
            a += I[j];
            b += (double)I[j] * I[j];
            avg += I[j];
        }
        abc[4 * i] = a;
        abc[4 * i + 1] = b;
        abc[4 * i + 2] = 1 / sqrt(ONC_LEN * b - a * a);
        abc[4 * i + 3] = avg / ONC_LEN;
    }
}

/* ---- NewMapPtsNCC::run + output behind getEpiNccMat, in C (bench.py's CPU baseline: a Python loop would not be a fair CPU figure) ---------
 * oracle.new_map_points_from_pairs (oracle/__init__.py: the line-cited restatement of /root/reference/src/app/SL_NewMapPointsInterCam.cpp:
 * 97-127, 150-161, 163-192, 194-270, 25-91, 295-316, 631-690, itself pinned to the reference's own featTracksFromMatches / reconstructTracks
 * / decidePointType compiled in place) operation for operation; tests/test_oracle_cpu.py holds the two against each other.
 * nC cameras with N slots each; consecutive pairs (a, a + 1); pairs[a] = nPairs[a] candidates {i, j, epi, ncc} of the NCC stage (pairStride
 * doubles apart: 4 doubles each).  Ks / iKs / Rs [nC][9], ts [nC][3]; xy [nC][2N]; state / slot2map / isStatic [nC][N] (slot2map is written);
 * the map arrays (mapPts [cap][3], mapCov [cap][9], mapFlags, newPt, firstFrame [cap], pointFeat [cap][nC]) are appended behind *mapCount.
 * match [nC - 1][N] out.  Returns the number of new points.  TEST INFRASTRUCTURE. */
typedef struct {
    double negNcc;
    int i, j;
} onc_cand;
static int onc_cand_cmp(const void* a, const void* b) {
    const onc_cand *x = (const onc_cand*)a, *y = (const onc_cand*)b;
    if (x->negNcc != y->negNcc) return x->negNcc < y->negNcc ? -1 : 1;
    if (x->i != y->i) return x->i < y->i ? -1 : 1;
    return x->j < y->j ? -1 : (x->j > y->j ? 1 : 0);
}
static double onc_sym33_cof(const double* S, double* c) {
    c[0] = S[3] * S[5] - S[4] * S[4], c[1] = S[2] * S[4] - S[1] * S[5], c[2] = S[1] * S[4] - S[2] * S[3];
    c[3] = S[0] * S[5] - S[2] * S[2], c[4] = S[1] * S[2] - S[0] * S[4], c[5] = S[0] * S[3] - S[1] * S[1];
    return (S[0] * c[0] + S[1] * c[1]) + S[2] * c[2];
}
int onc_new_points_from_pairs(int nC, int N, const double* const* pairs, const int* nPairs, const double* Ks, const double* iKs, const double* Rs,
                              const double* ts, const double* xy, const int* state, int* slot2map, const unsigned char* isStatic, double* mapPts,
                              double* mapCov, unsigned char* mapFlags, unsigned char* newPt, int* firstFrame, int* pointFeat, int cap,
                              int* mapCount, int curFrame, double maxDisp, double maxRpErr, double sigma, int minLen, int maxSeeds, int W, int H,
                              int* match) {
    unsigned char* hasIn = (unsigned char*)calloc((size_t)nC * N, 1);
    unsigned char *rowUsed = (unsigned char*)malloc(N), *colUsed = (unsigned char*)malloc(N);
    double* seeds = (double*)malloc(sizeof(double) * 4 * (maxSeeds > 0 ? maxSeeds : 1));
    int count = *mapCount, nNew = 0;
    int* newIdx = (int*)malloc(sizeof(int) * (size_t)(nC > 1 ? (nC - 1) : 1) * N);
    for (int q = 0; q < (nC - 1) * N; q++) match[q] = -1;
    for (int a = 0; a + 1 < nC; a++) {
        const int b = a + 1;
        const double *xa = xy + (size_t)a * 2 * N, *xb = xy + (size_t)b * 2 * N;
        int nSeeds = 0; /* getSeedsBetween (:97-127), map order */
        for (int m = 0; m < (count < cap ? count : cap) && nSeeds < maxSeeds; m++) {
            if (mapFlags[m] & 6) continue;
            const int s1 = pointFeat[(size_t)m * nC + a], s2 = pointFeat[(size_t)m * nC + b];
            if (s1 >= 0 && s2 >= 0) {
                double* sd = seeds + 4 * nSeeds++;
                sd[0] = xa[s1], sd[1] = xa[N + s1], sd[2] = xb[s2] - sd[0], sd[3] = xb[N + s2] - sd[1];
            }
        }
        onc_cand* cand = (onc_cand*)malloc(sizeof(onc_cand) * (size_t)(nPairs[a] > 0 ? nPairs[a] : 1));
        int nCand = 0;
        for (int q = 0; q < nPairs[a]; q++) {
            const double* p = pairs[a] + 4 * (size_t)q;
            const int i = (int)p[0], j = (int)p[1];
            if (nSeeds) { /* getDisparityMat + the guide */
                const double x1 = xa[i], y1 = xa[N + i];
                double best = 1.0e300;
                int bk = 0;
                for (int k = 0; k < nSeeds; k++) {
                    const double dx = seeds[4 * k] - x1, dy = seeds[4 * k + 1] - y1, d2 = dx * dx + dy * dy;
                    if (d2 < best) best = d2, bk = k;
                }
                const double ex = (xb[j] - x1) - seeds[4 * bk + 2], ey = (xb[N + j] - y1) - seeds[4 * bk + 3];
                if (!(sqrt(ex * ex + ey * ey) <= maxDisp)) continue;
            }
            cand[nCand].negNcc = -p[3], cand[nCand].i = i, cand[nCand].j = j, nCand++;
        }
        qsort(cand, nCand, sizeof(onc_cand), onc_cand_cmp); /* falling score, then rising row, then rising column */
        memset(rowUsed, 0, N), memset(colUsed, 0, N);
        for (int q = 0; q < nCand; q++) {
            if (rowUsed[cand[q].i] || colUsed[cand[q].j]) continue;
            rowUsed[cand[q].i] = colUsed[cand[q].j] = 1;
            match[(size_t)a * N + cand[q].i] = cand[q].j;
            hasIn[(size_t)b * N + cand[q].j] = 1;
        }
        free(cand);
    }
    int tkC[64], tkS[64];
    for (int a = 0; a + 1 < nC; a++) /* featTracksFromMatches (:631-690): numbered by (pair, feature) */
        for (int i0 = 0; i0 < N; i0++) {
            if (match[(size_t)a * N + i0] < 0 || (a > 0 && hasIn[(size_t)a * N + i0])) continue;
            int len = 0, c = a, s = i0;
            tkC[len] = c, tkS[len] = s, len++;
            while (c < nC - 1 && match[(size_t)c * N + s] >= 0) {
                s = match[(size_t)c * N + s];
                c++;
                tkC[len] = c, tkS[len] = s, len++;
            }
            if (len < minLen) continue; /* reconstructTracks (:194-270) */
            double Nn[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            for (int v = 0; v < len; v++) {
                const double *iK = iKs + 9 * tkC[v], *R = Rs + 9 * tkC[v], *t = ts + 3 * tkC[v];
                const double mx = xy[(size_t)tkC[v] * 2 * N + tkS[v]], my = xy[(size_t)tkC[v] * 2 * N + N + tkS[v]];
                const double w = (iK[6] * mx + iK[7] * my) + iK[8];
                const double x = ((iK[0] * mx + iK[1] * my) + iK[2]) / w, y = ((iK[3] * mx + iK[4] * my) + iK[5]) / w;
                const double a0[3] = {R[0] - x * R[6], R[1] - x * R[7], R[2] - x * R[8]}, a1[3] = {R[3] - y * R[6], R[4] - y * R[7], R[5] - y * R[8]};
                const double b0 = x * t[2] - t[0], b1 = y * t[2] - t[1];
                Nn[0] = Nn[0] + (a0[0] * a0[0] + a1[0] * a1[0]), Nn[1] = Nn[1] + (a0[0] * a0[1] + a1[0] * a1[1]);
                Nn[2] = Nn[2] + (a0[0] * a0[2] + a1[0] * a1[2]), Nn[3] = Nn[3] + (a0[1] * a0[1] + a1[1] * a1[1]);
                Nn[4] = Nn[4] + (a0[1] * a0[2] + a1[1] * a1[2]), Nn[5] = Nn[5] + (a0[2] * a0[2] + a1[2] * a1[2]);
                for (int q = 0; q < 3; q++) g[q] = g[q] + (a0[q] * b0 + a1[q] * b1);
            }
            double cf[6];
            const double det = onc_sym33_cof(Nn, cf);
            const double M[3] = {((cf[0] * g[0] + cf[1] * g[1]) + cf[2] * g[2]) / det, ((cf[1] * g[0] + cf[3] * g[1]) + cf[4] * g[2]) / det,
                                 ((cf[2] * g[0] + cf[4] * g[1]) + cf[5] * g[2]) / det};
            int outlier = 0;
            double S[6] = {0, 0, 0, 0, 0, 0};
            for (int v = 0; v < len; v++) {
                const double *K = Ks + 9 * tkC[v], *R = Rs + 9 * tkC[v], *t = ts + 3 * tkC[v];
                const double X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0], Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1];
                const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
                const double u = (K[0] * X + K[1] * Y) + K[2] * Z, vv = (K[3] * X + K[4] * Y) + K[5] * Z, w = (K[6] * X + K[7] * Y) + K[8] * Z;
                const double dx = xy[(size_t)tkC[v] * 2 * N + tkS[v]] - u / w, dy = xy[(size_t)tkC[v] * 2 * N + N + tkS[v]] - vv / w;
                const double e = sqrt(dx * dx + dy * dy);
                if (e > maxRpErr || Z < 0) outlier = 1;
                double KR[9], J[6];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) KR[3 * i + j] = (K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j]) + K[3 * i + 2] * R[6 + j];
                const double ww = w * w;
                for (int j = 0; j < 3; j++) J[j] = (KR[j] * w - u * KR[6 + j]) / ww, J[3 + j] = (KR[3 + j] * w - vv * KR[6 + j]) / ww;
                S[0] = S[0] + (J[0] * J[0] + J[3] * J[3]), S[1] = S[1] + (J[0] * J[1] + J[3] * J[4]), S[2] = S[2] + (J[0] * J[2] + J[3] * J[5]);
                S[3] = S[3] + (J[1] * J[1] + J[4] * J[4]), S[4] = S[4] + (J[1] * J[2] + J[4] * J[5]), S[5] = S[5] + (J[2] * J[2] + J[5] * J[5]);
            }
            if (outlier || count >= cap) continue;
            const double dS = onc_sym33_cof(S, cf), s2 = sigma * sigma;
            const int m = count++;
            double* cov = mapCov + 9 * (size_t)m;
            cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
            cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
            cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
            memcpy(mapPts + 3 * (size_t)m, M, 24);
            int nDyn = 0;
            for (int v = 0; v < len; v++) nDyn += isStatic && !isStatic[(size_t)tkC[v] * N + tkS[v]];
            mapFlags[m] = nDyn > 1 ? 1 : 4; /* setLocalDynamic / setUncertain (:253-264) */
            newPt[m] = 1, firstFrame[m] = curFrame;
            for (int c2 = 0; c2 < nC; c2++) pointFeat[(size_t)m * nC + c2] = -1;
            for (int v = 0; v < len; v++) pointFeat[(size_t)m * nC + tkC[v]] = tkS[v], slot2map[(size_t)tkC[v] * N + tkS[v]] = m;
            newIdx[nNew++] = m;
        }
    /* decidePointType (:25-91) */
    int anyUncertain = 0;
    for (int q = 0; q < nNew; q++) anyUncertain |= mapFlags[newIdx[q]] == 4;
    if (anyUncertain) {
        int* dynXY = (int*)malloc(sizeof(int) * 2 * (size_t)nC * N);
        int* nDynC = (int*)calloc(nC, sizeof(int));
        for (int c = 0; c < nC; c++)
            for (int s = 0; s < N; s++) {
                const int m = slot2map[(size_t)c * N + s], st = state[(size_t)c * N + s];
                if ((st == 0 || st == 1) && m >= 0 && m < cap && mapFlags[m] == 1) {
                    int* o = dynXY + 2 * ((size_t)c * N + nDynC[c]++);
                    o[0] = (int)(xy[(size_t)c * 2 * N + s] + 0.5), o[1] = (int)(xy[(size_t)c * 2 * N + N + s] + 0.5);
                }
            }
        for (int q = 0; q < nNew; q++) {
            const int m = newIdx[q];
            if (mapFlags[m] != 4) continue;
            int isStat = 1;
            for (int c = 0; c < nC && isStat; c++) {
                const int s = pointFeat[(size_t)m * nC + c];
                if (s < 0) continue;
                const int x = (int)(xy[(size_t)c * 2 * N + s] + 0.5), y = (int)(xy[(size_t)c * 2 * N + N + s] + 0.5);
                if (!(x >= 0 && x < W && y >= 0 && y < H)) continue;
                for (int k = 0; k < nDynC[c]; k++) {
                    const int* o = dynXY + 2 * ((size_t)c * N + k);
                    if (abs(x - o[0]) <= 20 && abs(y - o[1]) <= 20) {
                        isStat = 0;
                        break;
                    }
                }
            }
            if (isStat) mapFlags[m] = 0;
        }
        free(dynXY), free(nDynC);
    }
    *mapCount = count;
    free(hasIn), free(rowUsed), free(colUsed), free(seeds), free(newIdx);
    return nNew;
}
