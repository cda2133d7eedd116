// keyframe.hip -- the key-frame decision of CoSLAM::genNewMapPoints, per frame, for every camera in one launch (gfx950).
//
// Replaces CoSLAM::IsReadyForKeyFrame (src/app/SL_CoSLAM.cpp:1269-1279) with getCurMapCenterViewFrom (:1224-1247),
// IsMappedPtsDecreaseBelow (:1249-1268), SingleSLAM::getNumMappedStaticPts (src/app/SL_SingleSLAM.cpp:121-136),
// getViewAngleChangeSelf / getCameraTranslationSelf (:825-834; getCamDist / getViewAngleChange, src/slam/SL_SLAMHelper.cpp:201-217), the
// loop over the cameras in genNewMapPoints (:1298-1309: nReady, `decrease`) and -- on request -- what addKeyFrame (:1280-1293) does to
// each camera's key-pose state when one camera's mapped points have decreased (SingleSLAM::addKeyPose, SL_SingleSLAM.cpp:835-862).
//
// The reference walks a camera's FeaturePoint list of the frame three times on the host; here a workgroup per camera reads the
// hand-back's records (state, slot2map) once: the count of certainly static mapped features, the count of mapped features whose point
// is older than the last key pose, the mean of the mapped (not false) points -- the two loops of :1233 and :1258 stop BEFORE the frame's
// last feature (`fp && fp != pTail`), getNumMappedStaticPts does not: restated as written.  The mean is a tree sum (the reference adds in
// list order): it feeds one comparison against 5 degrees.  Angles are compared by cosine (cos is monotone on [0, pi]; the reference's
// "PI = 3.14" is kept: the threshold is cos(5 * 3.14 / 180) computed on the host).
#include "cs_common.h"

namespace {

constexpr int KF_MAX_CAMS = 16;
struct KfArgs {
    int nCams, N, nMap, curFrame, add;
    const double* mapPts;
    const unsigned char* mapFlags;
    const int* firstFrame;
    double ratio, cosMinAngle, minTranslation;
    int* ready;      // [nCams + 2]: the cameras' codes, nReady, decrease
    int* mapped;     // [2 nCams]: m_nMappedStaticPts, then IsMappedPtsDecreaseBelow's num
    double* center;  // [nCams][3]
    int* stats;      // [5] or null, accumulated: frames with nReady > 0, frames with decrease, cameras saying 1 / 2 / 3
    cs_keyframe_cam cam[KF_MAX_CAMS];
};

__device__ __forceinline__ void kf_center(const double* R, const double* t, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i) C[i] = -((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]);
}

__global__ __launch_bounds__(256) void k_keyframe_ready(KfArgs A) {
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const cs_keyframe_cam& C = A.cam[c];
    const int keyFrame = *C.keyFrame;
    __shared__ int sLast[4], sCnt[4][3];
    __shared__ double sSum[4][3];
    // the frame's last feature in list (= slot) order
    int last = -1;
    for (int s = tid; s < A.N; s += 256) {
        const int st = C.state[s];
        if (st == 0 || st == 1) last = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    if (lane == 0) sLast[wv] = last;
    __syncthreads();
    last = max(max(sLast[0], sLast[1]), max(sLast[2], sLast[3]));
    int nStatic = 0, num = 0, nCen = 0;
    double sx = 0, sy = 0, sz = 0;
    for (int s = tid; s < A.N; s += 256) {
        const int st = C.state[s];
        if (st != 0 && st != 1) continue;
        const int m = C.slot2map[s];
        if (m < 0 || m >= A.nMap) continue;
        const unsigned char fl = A.mapFlags[m];
        if ((fl & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN)) == 0) ++nStatic;   // isCertainStatic(), SL_SingleSLAM.cpp:131
        if (s == last) continue;                                                          // :1233, :1258
        if (A.firstFrame[m] <= keyFrame) ++num;                                           // :1259
        if (!(fl & CS_MAP_FALSE)) {                                                       // :1234
            sx += A.mapPts[3 * (size_t)m], sy += A.mapPts[3 * (size_t)m + 1], sz += A.mapPts[3 * (size_t)m + 2];
            ++nCen;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nStatic += __shfl_xor(nStatic, o, 64), num += __shfl_xor(num, o, 64), nCen += __shfl_xor(nCen, o, 64);
        sx += __shfl_xor(sx, o, 64), sy += __shfl_xor(sy, o, 64), sz += __shfl_xor(sz, o, 64);
    }
    if (lane == 0) sCnt[wv][0] = nStatic, sCnt[wv][1] = num, sCnt[wv][2] = nCen, sSum[wv][0] = sx, sSum[wv][1] = sy, sSum[wv][2] = sz;
    __syncthreads();
    if (tid != 0) return;
    nStatic = (sCnt[0][0] + sCnt[1][0]) + (sCnt[2][0] + sCnt[3][0]);
    num = (sCnt[0][1] + sCnt[1][1]) + (sCnt[2][1] + sCnt[3][1]);
    nCen = (sCnt[0][2] + sCnt[1][2]) + (sCnt[2][2] + sCnt[3][2]);
    double cen[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cen[k] = ((sSum[0][k] + sSum[1][k]) + (sSum[2][k] + sSum[3][k])) / (double)nCen;   // (no mapped point: NaN -- the reference pause()s)
    int code = 0;
    const int lastNum = *C.keyMapped;
    if ((double)num < (double)lastNum * A.ratio || num < 30)   // :1262-1266
        code = 1;
    else {
        double C0[3], C1[3];
        kf_center(C.selfR, C.selfT, C0), kf_center(C.R, C.t, C1);
        const double a[3] = {C0[0] - cen[0], C0[1] - cen[1], C0[2] - cen[2]}, b[3] = {C1[0] - cen[0], C1[1] - cen[1], C1[2] - cen[2]};
        const double d = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
        const double na = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2], nb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
        const double e0 = C0[0] - C1[0], e1 = C0[1] - C1[1], e2 = C0[2] - C1[2];
        if (d / sqrt(na * nb) < A.cosMinAngle)   // getViewAngleChange > m_minViewAngleChange (:1274)
            code = 2;
        else if (sqrt((e0 * e0 + e1 * e1) + e2 * e2) > A.minTranslation)   // :1276
            code = 3;
    }
    A.ready[c] = code;
    A.mapped[c] = nStatic, A.mapped[A.nCams + c] = num;
#pragma unroll
    for (int k = 0; k < 3; ++k) A.center[3 * c + k] = cen[k];
}

// genNewMapPoints' summary (:1298-1309) and, with `add`, addKeyFrame's effect on the cameras' key-pose state when `decrease` holds
__global__ __launch_bounds__(64) void k_keyframe_sum(KfArgs A) {
    const int c = threadIdx.x;
    const int code = c < A.nCams ? A.ready[c] : 0;
    const unsigned long long any = __builtin_amdgcn_ballot_w64(code > 0), dec = __builtin_amdgcn_ballot_w64(code == 1);
    const unsigned long long ang = __builtin_amdgcn_ballot_w64(code == 2), tra = __builtin_amdgcn_ballot_w64(code == 3);
    if (c == 0) {
        A.ready[A.nCams] = __popcll(any), A.ready[A.nCams + 1] = dec ? 1 : 0;
        if (A.stats) {
            if (any) atomicAdd(A.stats, 1);
            if (dec) atomicAdd(A.stats + 1, 1);
            atomicAdd(A.stats + 2, __popcll(dec));
            atomicAdd(A.stats + 3, __popcll(ang));
            atomicAdd(A.stats + 4, __popcll(tra));
        }
    }
    if (A.add && dec && c < A.nCams) {   // addKeyFrame -> slam[i].addKeyPose(readyForKeyFrame[i] > 0) for EVERY camera
        const cs_keyframe_cam& C = A.cam[c];
        *C.keyFrame = A.curFrame, *C.keyMapped = A.mapped[c];
        if (code > 0) {   // bSelfMotion: the pose joins m_selfKeyPose
            for (int k = 0; k < 9; ++k) C.selfR[k] = C.R[k];
            for (int k = 0; k < 3; ++k) C.selfT[k] = C.t[k];
        }
    }
}

// what a key frame's push would read of this frame, and the decision word, into one slot of a caller's ring: ONE launch (cs_keyframe_snapshot_dev)
__global__ __launch_bounds__(256) void k_keyframe_snapshot(int nXY, int nSt, int nR, int nT, const double* __restrict__ xy, const int* __restrict__ st,
                                                           const int* __restrict__ s2m, const double* __restrict__ R, const double* __restrict__ t,
                                                           const int* __restrict__ word, double* __restrict__ oxy, int* __restrict__ ost,
                                                           int* __restrict__ os2m, double* __restrict__ oR, double* __restrict__ ot, int* hostWord) {
    const int q = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int i = q; i < nXY; i += stride) oxy[i] = xy[i];
    for (int i = q; i < nSt; i += stride) ost[i] = st[i], os2m[i] = s2m[i];
    for (int i = q; i < nR; i += stride) oR[i] = R[i];
    for (int i = q; i < nT; i += stride) ot[i] = t[i];
    if (q == 0) __hip_atomic_store(hostWord, *word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (pinned host memory: seen behind the stream's next event)
}

}  // namespace

extern "C" int cs_keyframe_snapshot_dev(int device, void* hip_stream, int nCams, int N, const double* d_xy, const int* d_state, const int* d_slot2map,
                                        const double* d_R, const double* d_t, const int* d_word, double* d_xyOut, int* d_stateOut, int* d_slot2mapOut,
                                        double* d_ROut, double* d_tOut, int* h_word) {
    if (nCams < 1 || N < 1 || !d_xy || !d_state || !d_slot2map || !d_R || !d_t || !d_word || !d_xyOut || !d_stateOut || !d_slot2mapOut || !d_ROut ||
        !d_tOut || !h_word) {
        cs_set_error("cs_keyframe_snapshot_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(device));
    const int nXY = nCams * 2 * N;
    hipLaunchKernelGGL(k_keyframe_snapshot, dim3((nXY + 255) / 256 < 64 ? (nXY + 255) / 256 : 64), dim3(256), 0, (hipStream_t)hip_stream, nXY, nCams * N,
                       9 * nCams, 3 * nCams, d_xy, d_state, d_slot2map, d_R, d_t, d_word, d_xyOut, d_stateOut, d_slot2mapOut, d_ROut, d_tOut, h_word);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_keyframe_ready_dev(int device, void* hip_stream, int nCams, int N, const cs_keyframe_cam* cams, int nMap, const double* d_mapPts,
                                     const unsigned char* d_mapFlags, const int* d_firstFrame, int curFrame, double ratio,
                                     double minViewAngleDeg, double minTranslation, int addKeyFrame, int* d_ready, int* d_mapped,
                                     double* d_center, int* d_stats) {
    if (!cams || nCams < 1 || nCams > KF_MAX_CAMS || N < 1 || nMap < 0 || !d_mapPts || !d_mapFlags || !d_firstFrame || !d_ready || !d_mapped ||
        !d_center) {
        cs_set_error("cs_keyframe_ready_dev: bad arguments (1..%d cameras)", KF_MAX_CAMS);
        return CS_ERR_INVALID;
    }
    KfArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = nCams, A.N = N, A.nMap = nMap, A.curFrame = curFrame, A.add = addKeyFrame ? 1 : 0;
    A.mapPts = d_mapPts, A.mapFlags = d_mapFlags, A.firstFrame = d_firstFrame;
    A.ratio = ratio, A.cosMinAngle = cos(minViewAngleDeg * 3.14 / 180.0), A.minTranslation = minTranslation;   // ("const double PI = 3.14", SL_SLAMHelper.cpp:210)
    A.ready = d_ready, A.mapped = d_mapped, A.center = d_center, A.stats = d_stats;
    for (int c = 0; c < nCams; ++c) {
        const cs_keyframe_cam& q = cams[c];
        if (!q.state || !q.slot2map || !q.R || !q.t || !q.keyFrame || !q.keyMapped || !q.selfR || !q.selfT) {
            cs_set_error("cs_keyframe_ready_dev: null pointer in camera %d", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = q;
    }
    CS_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(k_keyframe_ready, dim3(nCams), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_keyframe_sum, dim3(1), dim3(64), 0, s, A);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
