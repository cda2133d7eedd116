// handback.hip -- on-device hand-back of the tracker's output to the pose stage, gfx950.
//
// Replaces, for every camera of a group in ONE launch:
//   GPUKLT::addToFeaturePoints            src/tracking/GPUKLT.cpp:36-60        normalised position -> pixel -> undistorted
//                                                                             pixel, the `out >= W | H` drop rule, track
//                                                                             bookkeeping (extend / restart / clear);
//   SingleSLAM::chooseStaticFeatPts       src/app/SL_SingleSLAM.cpp:345-397    one track per 40 x 40 px block (mapped
//                                                                             first, else the longest);
//   the 3D-2D packing of poseUpdate3D     src/app/SL_SingleSLAM.cpp:620-640    Ms / ms of intraCamEstimate.
// The reference heap-allocates one FeaturePoint per feature per frame and walks pointer lists on the host; here the
// tracker's dest[] never leaves HBM: a structure-of-arrays record per slot (undistorted pixel, state, track length,
// slot -> map point) and the packed correspondences the batched pose kernel (pose.hip) consumes directly.
//
// undistorPoint(K, k_ud, in, out) lives in un-vendored LibVisualSLAM; only its call (GPUKLT.cpp:45) and the 7-vector
// k_ud (src/tracking/GPUKLT.h:44-47) are in the reference.  Our definition: normalise with K, scale by
// 1 + sum_{i=0..6} k_ud[i] r^(2 (i + 1)), map back with K; k_ud = 0 is the identity (what the synthetic sequences use).
#include "cs_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int HB_MAX_CAMS = 16;
constexpr int HB_MAX_BLOCKS = 1024;

struct HbArgs {
    int N, W, H, nColBlk, nRowBlk, blkW, blkH, ptsStride, nCams, frame;
    cs_handback_cam cam[HB_MAX_CAMS];
};

__device__ __forceinline__ void undistort_point(const double* __restrict__ K, const double* __restrict__ kud, double x,
                                                double y, double& ox, double& oy) {
    const double yn = (y - K[5]) / K[4];
    const double xn = ((x - K[2]) - K[1] * yn) / K[0];
    const double r2 = xn * xn + yn * yn;
    double f = kud[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) f = f * r2 + kud[i];
    f = 1.0 + f * r2;
    const double xu = xn * f, yu = yn * f;
    ox = (K[0] * xu + K[1] * yu) + K[2];
    oy = K[4] * yu + K[5];
}

__global__ __launch_bounds__(1024) void k_handback(HbArgs A) {
    __shared__ unsigned long long key[HB_MAX_BLOCKS];
    __shared__ int flag[HB_MAX_BLOCKS];
    CS_POSE_STREAM_PRIO();
    const cs_handback_cam& C = A.cam[blockIdx.x];
    const int tid = threadIdx.x, N = A.N;
    const int nBlk = A.nColBlk * A.nRowBlk;
    for (int b = tid; b < nBlk; b += 1024) key[b] = 0ull;
    if (C.pointFeat)
        for (int q = tid; q < C.nPointFeat; q += 1024) C.pointFeat[(size_t)q * C.pointFeatStride] = -1;
    __syncthreads();
    // ---- GPUKLT.cpp:36-60 per slot, then the block vote of SL_SingleSLAM.cpp:353-384 ------------------------------
    for (int i = tid; i < N; i += 1024) {
        const cs_klt_feature f = C.dest[i];
        // Track2D keeps the frame span [f1, f2] of the slot's track; its length() is f2 - f1 + 1 (src/tracking/SL_Track2D.h:63-65)
        int st = f.status, f1 = C.trackSpan[i], f2 = C.trackSpan[N + i], mp = C.slot2map[i];
        double x = C.xy[i], y = C.xy[N + i];
        if (st >= 0) {
            const double inx = (double)(f.pos[0] * (float)A.W), iny = (double)(f.pos[1] * (float)A.H);  // :43-44
            double ox, oy;
            undistort_point(C.K, C.kud, inx, iny, ox, oy);
            if (ox >= (double)A.W || oy >= (double)A.H) {  // :46-47 `continue`: the track is neither extended nor cleared
                st = -2;
            } else {
                x = ox;
                y = oy;
                if (st == 0 && f1 >= 0) {
                    f2 = A.frame;  // :50-52 m_tks[i].add(p): the span grows to this frame
                } else {
                    f1 = f2 = A.frame;  // :53-57 newly detected (or first point of an empty track): the track restarts
                    if (st != 0) mp = -1;
                }
            }
        } else {
            f1 = f2 = -1;  // :59 m_tks[i].clear()
            mp = -1;
        }
        const int len = (f1 >= 0) ? f2 - f1 + 1 : 0;
        C.xy[i] = x;
        C.xy[N + i] = y;
        C.state[i] = st;
        C.trackSpan[i] = f1;
        C.trackSpan[N + i] = f2;
        C.slot2map[i] = mp;
        // MapPoint::pFeatures[iCam] of the current frame: the (highest) slot in this frame's list that carries the point
        if (C.pointFeat && mp >= 0 && mp < C.nPointFeat && (st == 0 || st == 1)) atomicMax(&C.pointFeat[(size_t)mp * C.pointFeatStride], i);
        if (len > 0) {  // !tk->empty(); candidates are the static ones: here every mapped slot and, when the caller
                        // supplies the classification, every slot it marks static
            // (a track born in this frame has no predecessor to take a type from -- propagateFeatureStates, SL_SingleSLAM.cpp:40-42 --
            // and keeps the constructor's TYPE_FEATPOINT_STATIC, whatever the slot's previous track left in isStatic[])
            const bool isStatic = (mp >= 0) || (C.isStatic && (C.isStatic[i] || f1 == A.frame));
            if (isStatic) {
                const int bx = (int)(x / (double)A.blkW), by = (int)(y / (double)A.blkH);
                if (bx < A.nColBlk && by < A.nRowBlk && bx >= 0 && by >= 0) {
                    // a mapped track is never displaced (first one in slot order wins); otherwise the longest, first on ties
                    const unsigned long long order = (unsigned long long)(N - 1 - i);
                    const unsigned long long k = (mp >= 0) ? ((1ull << 62) | order) : (((unsigned long long)len << 24) | order);
                    atomicMax(&key[by * A.nColBlk + bx], k);
                }
            }
        }
    }
    __syncthreads();
    // ---- featPts in block order (:386-394), then the mapped ones packed for intraCamEstimate (:620-640) ----------
    for (int b = tid; b < nBlk; b += 1024) {
        const unsigned long long k = key[b];
        const int slot = k ? (N - 1 - (int)(k & 0xFFFFFFull)) : -1;
        flag[b] = (k >> 62) ? 1 : 0;
        if (C.selBlk) C.selBlk[b] = slot;
    }
    __syncthreads();
    for (int b = tid; b < nBlk; b += 1024) {
        if (!flag[b]) continue;
        int r = 0;
        for (int q = 0; q < b; ++q) r += flag[q];
        if (r >= A.ptsStride) continue;
        const int slot = N - 1 - (int)(key[b] & 0xFFFFFFull);
        const int mp = C.slot2map[slot];
        C.sel[r] = slot;
        C.Ms[3 * r] = C.mapPts[3 * (size_t)mp];
        C.Ms[3 * r + 1] = C.mapPts[3 * (size_t)mp + 1];
        C.Ms[3 * r + 2] = C.mapPts[3 * (size_t)mp + 2];
        C.ms[2 * r] = C.xy[slot];
        C.ms[2 * r + 1] = C.xy[N + slot];
    }
    if (tid == 0) {
        int n = 0;
        for (int q = 0; q < nBlk; ++q) n += flag[q];
        *C.npts = n < A.ptsStride ? n : A.ptsStride;
        if (C.opt) {  // IntraCamPoseOption(), src/slam/SL_IntraCamPose.h:42-46: the pose kernel's in/out block, reset per frame
            cs_pose_option o;
            o.maxIterLM = 100;
            o.maxIterRW = 5;
            o.epsErrorChangeLM = 1e-7;
            o.epsParamChangeLM = 1e-6;
            o.epsErrorChangeRW = 1e-6;
            o.verboseLM = o.verboseRW = 0;
            o.lambda0 = 1e-3;
            o.lambda = 0;
            o.err0 = o.err = o.errRW = 0;
            o.retTypeLM = o.npts = o.nIterLM = o.nIterRW = 0;
            *C.opt = o;
        }
    }
}

}  // namespace

extern "C" int cs_klt_handback_dev(int device, void* hip_stream, int nCams, const cs_handback_cam* cams, int N, int W, int H,
                                   int nColBlk, int nRowBlk, int ptsStride, int frame) {
    if (nCams < 1 || nCams > HB_MAX_CAMS || !cams || N < 1 || N >= (1 << 24) || W < 1 || H < 1 || frame < 0 || nColBlk < 1 || nRowBlk < 1 ||
        nColBlk * nRowBlk > HB_MAX_BLOCKS || ptsStride < 1 || W / nColBlk < 1 || H / nRowBlk < 1) {
        cs_set_error("cs_klt_handback_dev: bad arguments (1..%d cameras, <= %d blocks)", HB_MAX_CAMS, HB_MAX_BLOCKS);
        return CS_ERR_INVALID;
    }
    HbArgs A;
    memset(&A, 0, sizeof(A));
    A.N = N;
    A.W = W;
    A.H = H;
    A.nColBlk = nColBlk;
    A.nRowBlk = nRowBlk;
    A.blkW = W / nColBlk;  // src/app/SL_SingleSLAM.cpp:270-271 (integer division)
    A.blkH = H / nRowBlk;
    A.ptsStride = ptsStride;
    A.nCams = nCams;
    A.frame = frame;
    for (int c = 0; c < nCams; ++c) {
        const cs_handback_cam& q = cams[c];
        if (!q.dest || !q.K || !q.kud || !q.mapPts || !q.slot2map || !q.trackSpan || !q.xy || !q.state || !q.Ms || !q.ms ||
            !q.sel || !q.npts) {
            cs_set_error("cs_klt_handback_dev: null pointer in camera %d", c);
            return CS_ERR_INVALID;
        }
        if (q.pointFeat && (q.pointFeatStride < 1 || q.nPointFeat < 0)) {
            cs_set_error("cs_klt_handback_dev: bad pointFeat stride / count in camera %d", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = q;
    }
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_handback, dim3(nCams), dim3(1024), 0, (hipStream_t)hip_stream, A);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
