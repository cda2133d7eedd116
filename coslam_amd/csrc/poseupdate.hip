// poseupdate.hip -- what a frame does with the cameras' new poses, on the device (second half), gfx950.
//
// Replaces, for every camera of a group in ONE launch behind the batched intraCamEstimate:
//   SingleSLAM::poseUpdate3D, second half   src/app/SL_SingleSLAM.cpp:672-708  per static mapped track node: project,
//       getProjectionCovMat, mat22Inv, mahaDist2 against 2.0 (6.0 with largeErr); inlier: FeaturePoint::reprojErr = that
//       distance and seqTriangulate updates the map point and its covariance in place; outlier: reprojErr = the pixel
//       distance, MapPoint::setUncertain()
//   SingleSLAM::getStaticMappedTrackNodes   :60-75
//   SingleSLAM::detectDynamicFeaturePoints  :784-824   per track of length >= minLen whose tail feature is unmapped or on a
//       certain-dynamic map point: walk the track backwards, count the past positions whose epipolar error against the current
//       one (F from the two frames' poses) is >= maxEpiErr; more than minOutNum -> TYPE_FEATPOINT_DYNAMIC, else an unmapped
//       feature -> TYPE_FEATPOINT_STATIC
//   SingleSLAM::getUnMappedAndDynamicTrackNodes :91-105
//   the type / reprojErr hand-down of propagateFeatureStates (:40-42, :54) = the persistence of the two per-slot arrays.
// The reference walks pointer lists (Track2DNode -> FeaturePoint -> MapPoint / CamPoseItem, FeaturePoint::preFrame) per camera
// on one host core.  Here:
//   gate role      one LANE PER MAP POINT walks the point's features over the cameras IN CAMERA ORDER (the hand-back's
//                  pointFeat table: MapPoint::pFeatures[iCam] of this frame): CoSLAM::parallelPoseUpdate runs the cameras one
//                  after the other (src/app/SL_CoSLAM.cpp:398-410), so a point seen by several cameras is updated in camera
//                  order and a point one camera made uncertain is no node of the next -- the lane reproduces exactly that,
//                  with the point and its covariance in registers between the cameras.  (One feature per camera and point, as
//                  MapPoint::pFeatures[iCam] holds one; of two slots carrying the same point the higher one counts.)
//   dynamic role   eight lanes per (camera, slot) striding the walk.  The track's past positions come from a ring of the last H frames' hand-back
//                  pixels (cs_track_history), the fundamental matrices from the ring of the camera's poses: F_j of "j frames
//                  back" is computed once per workgroup into LDS (formEMat + getFMat), the walk then costs two coalesced loads
//                  and ~25 flops a step.  NOTE src/app/SL_SingleSLAM.cpp:799: the reference's loop counter `f` is never
//                  advanced, so its `f < maxLen` never ends the walk (it runs to the head of the track or until the count
//                  exceeds minOutNum); maxLen is accepted here and, as there, has no effect -- the walk is bounded by the ring
//                  depth H instead (choose H >= the longest track that matters; H x nCams x 16 N bytes of HBM).
// Both roles are workgroups of the same launch (the gate only ever marks STATIC points uncertain, the dynamic role only reads
// the certain-DYNAMIC bit pattern: no ordering between them is needed).
// What is NOT identical to the serial reference when all cameras' poses come from ONE batched intraCamEstimate launch: camera c
// > 0 estimated its pose from the map as it stood before cameras < c refined it (the reference's own _parallelPoseUpdate,
// SL_CoSLAM.cpp:390-397, has the same property).  cam0 / nCamsRun let a caller run camera by camera for the serial order.
//
// project, getProjectionCovMat, mat22Inv, mahaDist2, dist2, seqTriangulate, formEMat, getFMat, epipolarError are un-vendored
// LibVisualSLAM (only their calls are in the reference): definitions in DESIGN.md, the same as register.hip / ncc.hip use;
// seqTriangulate = one Kalman update of (M, cov) from the measurement with noise sigma^2 I.
#include <cstdlib>
#include <vector>

#include "cs_common.h"
#include "small_ops.h"

#pragma clang fp contract(off)

namespace {

constexpr int PU_MAX_CAMS = 16;
constexpr int PU_MAX_HIST = 512;
constexpr int PU_MAX_STORE = 1 << 16;
constexpr int PU_LPS = 8;  // lanes per slot in the dynamic-point test

struct PuArgs {
    int nCams, N, nMap, cam0, nCamsRun;
    int gateBlocks, dynBlocksPerCam;
    // gate
    const int* pointFeat;  // [nMap][nCams]
    const double* R;       // [nCams][9] the new poses
    const double* t;       // [nCams][3]
    double* mapPts;
    double* mapCov;
    unsigned char* mapFlags;
    double errThres, sigma;
    int* numNodes;  // [nCams] or null (zeroed by the caller)
    int* numOut;
    // dynamic
    int H, head, nHist;  // ring depth, ring slot of this frame, frames held (this one included)
    double* histXY;      // [nCams][H][2N]
    double* histR;       // [nCams][H][9]
    double* histT;       // [nCams][H][3]
    int minLen, minOutNum;
    double maxEpiErr;
    int* numDyn;  // [nCams] or null
    // fused behind the gate (cs_pose_update_classify_frame_dev): mapPointsClassify's worklist -- what k_classify_select builds -- by the
    // gate's own lane of a map point, and the camera centres by walk depth in blocks of their own (k_ring_centres); null / 0: not fused
    int* clsList;              // [2 + nMap]: counters of the two parities, then the list
    int clsPar, clsCurFrame;
    int* clsCounts;            // mapPointsClassify's counts [2] or null: [1] is reset here (the worker adds to it)
    const int* clsFeatFrame;   // [nMap][nCams] or null
    double* cenOut;            // [nCams][nHist][3] or null
    int cenBlocks;
    cs_poseupdate_cam cam[PU_MAX_CAMS];
};

__device__ __forceinline__ void pu_mat22_inv(const double A[4], double iA[4]) {
    const double det = A[0] * A[3] - A[1] * A[2];
    iA[0] = A[3] / det;
    iA[1] = -A[1] / det;
    iA[2] = -A[2] / det;
    iA[3] = A[0] / det;
}

// everything one (camera, point) pair needs of the projection: u, v, w and J = d project / dM
struct PuProj {
    double u, v, w, J[6];
};
__device__ __forceinline__ PuProj pu_project(const double* __restrict__ K, const double* __restrict__ R, const double* __restrict__ t,
                                             const double M[3]) {
    PuProj q;
    const double X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0];
    const double Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1];
    const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
    double KR[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) KR[3 * i + j] = (K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j]) + K[3 * i + 2] * R[6 + j];
    q.u = (K[0] * X + K[1] * Y) + K[2] * Z;
    q.v = (K[3] * X + K[4] * Y) + K[5] * Z;
    q.w = (K[6] * X + K[7] * Y) + K[8] * Z;
    const double ww = q.w * q.w;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        q.J[j] = (KR[j] * q.w - q.u * KR[6 + j]) / ww;
        q.J[3 + j] = (KR[3 + j] * q.w - q.v * KR[6 + j]) / ww;
    }
    return q;
}

__device__ __forceinline__ void pu_gate_point(const PuArgs& A, int m) {
    unsigned char fl = A.mapFlags[m];
    if (fl & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN)) return;  // !isCertainStatic()
    double M[3], cov[9];
    bool loaded = false, dirty = false;
    for (int c = A.cam0; c < A.cam0 + A.nCamsRun; ++c) {
        const int s = A.pointFeat[(size_t)m * A.nCams + c];
        if (s < 0) continue;
        const cs_poseupdate_cam& C = A.cam[c];
        if (!loaded) {
#pragma unroll
            for (int q = 0; q < 3; ++q) M[q] = A.mapPts[3 * (size_t)m + q];
#pragma unroll
            for (int q = 0; q < 9; ++q) cov[q] = A.mapCov[9 * (size_t)m + q];
            loaded = true;
        }
        const double* K = C.K;
        const double* R = A.R + 9 * c;
        const double* t = A.t + 3 * c;
        const double mx = C.xy[s], my = C.xy[A.N + s];
        const PuProj q = pu_project(K, R, t, M);
        const double rm0 = q.u / q.w, rm1 = q.v / q.w;  // project (:678)
        // getProjectionCovMat (:679): var = J cov J^T + sigma^2 I
        double JC[6], var[4], ivar[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) JC[3 * i + j] = (q.J[3 * i] * cov[j] + q.J[3 * i + 1] * cov[3 + j]) + q.J[3 * i + 2] * cov[6 + j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double sv = (JC[3 * i] * q.J[3 * j] + JC[3 * i + 1] * q.J[3 * j + 1]) + JC[3 * i + 2] * q.J[3 * j + 2];
                var[2 * i + j] = (i == j) ? sv + A.sigma * A.sigma : sv;
            }
        pu_mat22_inv(var, ivar);
        const double dx = rm0 - mx, dy = rm1 - my;
        const double err = dx * (ivar[0] * dx + ivar[1] * dy) + dy * (ivar[2] * dx + ivar[3] * dy);  // mahaDist2 (:681)
        if (A.numNodes) atomicAdd(A.numNodes + c, 1);
        if (err < A.errThres) {
            C.reprojErr[s] = err;  // :683
            // seqTriangulate (:684-685): S = J (cov J^T) + sigma^2 I, G = (cov J^T) S^-1, M += G (m - rm), cov -= G (cov J^T)^T
            double PJt[6], S[4], iS[4], G[6];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 2; ++k) PJt[2 * r + k] = (cov[3 * r] * q.J[3 * k] + cov[3 * r + 1] * q.J[3 * k + 1]) + cov[3 * r + 2] * q.J[3 * k + 2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double sv = (q.J[3 * r] * PJt[k] + q.J[3 * r + 1] * PJt[2 + k]) + q.J[3 * r + 2] * PJt[4 + k];
                    S[2 * r + k] = (r == k) ? sv + A.sigma * A.sigma : sv;
                }
            pu_mat22_inv(S, iS);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 2; ++k) G[2 * r + k] = PJt[2 * r] * iS[k] + PJt[2 * r + 1] * iS[2 + k];
            const double e0 = mx - rm0, e1 = my - rm1;
            double nc[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) nc[3 * r + k] = cov[3 * r + k] - (G[2 * r] * PJt[2 * k] + G[2 * r + 1] * PJt[2 * k + 1]);
#pragma unroll
            for (int r = 0; r < 3; ++r) M[r] = M[r] + (G[2 * r] * e0 + G[2 * r + 1] * e1);
#pragma unroll
            for (int r = 0; r < 9; ++r) cov[r] = nc[r];
            dirty = true;
        } else {
            if (A.numOut) atomicAdd(A.numOut + c, 1);
            C.reprojErr[s] = sqrt(dx * dx + dy * dy);  // :701-702 dist2
            fl |= CS_MAP_UNCERTAIN;                      // :704: no node of the cameras that follow
            A.mapFlags[m] = fl;
            break;
        }
    }
    if (dirty) {
#pragma unroll
        for (int q = 0; q < 3; ++q) A.mapPts[3 * (size_t)m + q] = M[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) A.mapCov[9 * (size_t)m + q] = cov[q];
    }
}

__device__ __forceinline__ void pu_mat33_ab(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// F of (the frame j steps back, this frame): formEMat(R1, t1, R0, t0, E), getFMat(iK, iK, E, F)   (:806-807)
__device__ __forceinline__ void pu_fmat(const double* __restrict__ iK, const double* __restrict__ R1, const double* __restrict__ t1,
                                        const double* __restrict__ R0, const double* __restrict__ t0, double* F) {
    const double R1t[9] = {R1[0], R1[3], R1[6], R1[1], R1[4], R1[7], R1[2], R1[5], R1[8]};
    double R[9], t[3], E[9], T[9];
    pu_mat33_ab(R0, R1t, R);
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = t0[i] - (R[3 * i] * t1[0] + R[3 * i + 1] * t1[1] + R[3 * i + 2] * t1[2]);
    const double Tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    pu_mat33_ab(Tx, R, E);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += iK[3 * k + i] * E[3 * k + j];
            T[3 * i + j] = s;
        }
    pu_mat33_ab(T, iK, F);
}

__device__ __forceinline__ void up_cam_center(const double* __restrict__ R, const double* __restrict__ t, double* C);

__global__ __launch_bounds__(256) void k_pose_update(PuArgs A) {
    extern __shared__ double Fs[];  // dynamic role: [nHist][9]
    CS_POSE_STREAM_PRIO();
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < A.gateBlocks) {
        const int m = blockIdx.x * 256 + tid;
        if (m < A.nMap) pu_gate_point(A, m);
        if (A.clsList) {   // k_classify_select's test, by the lane that has just finished the point's gate (every camera's, in camera order)
            if (m == 0 && A.clsCounts) A.clsCounts[1] = 0;
            if (m < A.nMap) {
                const unsigned char fl = A.mapFlags[m];
                if ((fl & CS_MAP_UNCERTAIN) || (fl & (CS_MAP_DYNAMIC | CS_MAP_FALSE)) == CS_MAP_DYNAMIC) {  // SL_CoSLAM.cpp:431
                    bool vis = false;
                    for (int c = 0; c < A.nCams && !vis; ++c) {
                        const int sl = A.pointFeat[(size_t)m * A.nCams + c];
                        vis = sl >= 0 && (A.clsFeatFrame ? A.clsFeatFrame[(size_t)m * A.nCams + c] : A.clsCurFrame) == A.clsCurFrame;
                    }
                    if (vis) A.clsList[2 + atomicAdd(A.clsList + A.clsPar, 1)] = m;
                }
            }
        }
        return;
    }
    if ((int)blockIdx.x >= (int)gridDim.x - A.cenBlocks) {   // the camera centres by walk depth; depth 0 = this frame's pose, given
        const int q = (blockIdx.x - (gridDim.x - A.cenBlocks)) * 256 + tid;
        if (q < A.nCams * A.nHist) {
            const int c = q / A.nHist, j = q - c * A.nHist, rs = (A.head - j + A.H) % A.H;
            if (j == 0) up_cam_center(A.R + 9 * c, A.t + 3 * c, A.cenOut + 3 * (size_t)q);
            else up_cam_center(A.histR + ((size_t)c * A.H + rs) * 9, A.histT + ((size_t)c * A.H + rs) * 3, A.cenOut + 3 * (size_t)q);
        }
        return;
    }
    const int b = blockIdx.x - A.gateBlocks;
    const int c = A.cam0 + b / A.dynBlocksPerCam, blk = b % A.dynBlocksPerCam;
    const cs_poseupdate_cam& C = A.cam[c];
    const int N = A.N, H = A.H;
    double* hR = A.histR + (size_t)c * H * 9;
    double* hT = A.histT + (size_t)c * H * 3;
    double* hXY = A.histXY + (size_t)c * H * 2 * N;
    const double* R0 = A.R + 9 * c;
    const double* t0 = A.t + 3 * c;
    // this frame's pose into the ring (FeaturePoint::cam of this frame's features, updateCamParamForFeatPts :333-344)
    if (blk == 0 && tid < 12) {
        if (tid < 9)
            hR[(size_t)A.head * 9 + tid] = R0[tid];
        else
            hT[(size_t)A.head * 3 + (tid - 9)] = t0[tid - 9];
    }
    for (int j = tid; j < A.nHist; j += 256) {
        const int rs = (A.head - j + H) % H;
        const double* R1 = j ? hR + (size_t)rs * 9 : R0;  // (j = 0: this frame itself, possibly not in the ring yet)
        const double* t1 = j ? hT + (size_t)rs * 3 : t0;
        pu_fmat(C.iK, R1, t1, R0, t0, Fs + 9 * j);
    }
    __syncthreads();
    // PU_LPS lanes per slot: lane r of the group takes the frames j = r, r + PU_LPS, ... of the walk.  The reference stops
    // walking once the count exceeds minOutNum; the verdict -- count > minOutNum -- is the same over the whole depth, so the
    // frames can be tested in any order and in parallel: a 64-frame walk is 8 steps deep.
    const int i = blk * (256 / PU_LPS) + tid / PU_LPS, r = tid % PU_LPS;
    if (i >= N) return;  // (whole groups leave together: the shuffles below stay inside a group)
    const int st = C.state[i];
    const double m0x = C.xy[i], m0y = C.xy[N + i];
    if (r == 0) {
        hXY[(size_t)A.head * 2 * N + i] = m0x;  // (every slot: a dead slot's entry is never read, its track is empty)
        hXY[(size_t)A.head * 2 * N + N + i] = m0y;
    }
    if (!(st == 0 || st == 1)) return;
    unsigned char type = C.isStatic[i];
    if (st == 1) type = 1;  // a new FeaturePoint: type(0) = TYPE_FEATPOINT_STATIC (src/slam/SL_FeaturePoint.cpp:23)
    const int f1 = C.trackSpan[i], f2 = C.trackSpan[N + i];
    const int len = f1 >= 0 ? f2 - f1 + 1 : 0;
    const int mp = C.slot2map[i];
    const bool mapped = mp >= 0 && mp < A.nMap;
    bool examine = len >= A.minLen;  // :96
    if (examine && mapped) {
        const unsigned char fl = A.mapFlags[mp];
        examine = (fl & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN)) == CS_MAP_DYNAMIC;  // isCertainDynamic() (:99)
    }
    if (examine) {
        int nOut = 0;
        const int depth = len < A.nHist ? len : A.nHist;
        for (int j = r; j < depth; j += PU_LPS) {  // :799 (`f` never advances: maxLen does not bound the walk)
            double bx = m0x, by = m0y;
            if (j) {
                const int rs = (A.head - j + H) % H;
                bx = hXY[(size_t)rs * 2 * N + i];
                by = hXY[(size_t)rs * 2 * N + N + i];
            }
            const double* F = Fs + 9 * j;
            // epipolarError(F, m0, m1): distance of m0 from the line F (m1, 1)   (:809)
            const double l0 = F[0] * bx + F[1] * by + F[2], l1 = F[3] * bx + F[4] * by + F[5], l2 = F[6] * bx + F[7] * by + F[8];
            const double n = sqrt(l0 * l0 + l1 * l1);
            const double err = fabs(l0 * m0x + l1 * m0y + l2) / (n > 0 ? n : 1.0);
            if (err >= A.maxEpiErr) ++nOut;
        }
#pragma unroll
        for (int off = 1; off < PU_LPS; off <<= 1) nOut += __shfl_xor(nOut, off, 64);  // (examine is uniform over the group)
        if (nOut > A.minOutNum) {
            type = 0;  // TYPE_FEATPOINT_DYNAMIC (:814-816)
            if (A.numDyn && r == 0) atomicAdd(A.numDyn + c, 1);
        } else if (!mapped) {
            type = 1;  // :817-818
        }
    }
    if (r == 0) C.isStatic[i] = type;
}

// ---- CoSLAM::staticCheckMergability (src/app/SL_CoSLAM.cpp:714-729) for every candidate of a registration search --------------
// The candidate feature and every earlier feature of its track (fp, fp->preFrame, ...) must lie within Mahalanobis distance 1 of
// the map point's projection under the pose of its own frame (FeaturePoint::cam), covariance J cov J^T + pixelVar^2 I; the walk
// stops at the first failure.  The reference runs it per candidate inside the registration loops on the host, following
// pointers; here a group of lanes = one (map point, camera) candidate, blockIdx.y = camera, the camera's ring of poses in LDS,
// the track's past pixels from the history ring (one 16-byte gather per step).
struct MgArgs {
    int cam0;  // cameras cam0 .. cam0 + gridDim.y - 1
    int nCams, N, P, H, head, nHist;
    double sigma;
    const double* M;
    const double* cov;
    const int* slot;
    unsigned char* out;
    const double* histXY;
    const double* histR;
    const double* histT;
    cs_poseupdate_cam cam[PU_MAX_CAMS];
};
// MG_LPC lanes per candidate: lane r of the group takes the frames j = r, r + MG_LPC, ... of the walk (the verdict is the AND over
// the frames: the order in which they are tested does not enter it), so a candidate's 64-frame walk is 8 steps deep and the
// 12 k candidates of a pass are 1500 waves -- the whole chip -- instead of 190 waves walking 64 dependent steps each (68 us).
constexpr int MG_LPC = 8;
// one frame's term of the walk (:718-726): the feature (mx, my) against the point's projection under that frame's pose, Mahalanobis
// distance^2 > 1 under J cov J^T + sigma^2 I
__device__ __forceinline__ bool mg_term_fails(const double* __restrict__ K, const double* R, const double* t, const double* M, const double* cov,
                                              double sigma, double mx, double my) {
    const PuProj q = pu_project(K, R, t, M);
    const double rm0 = q.u / q.w, rm1 = q.v / q.w;
    double JC[6], var[4], ivar[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) JC[3 * i + k] = (q.J[3 * i] * cov[k] + q.J[3 * i + 1] * cov[3 + k]) + q.J[3 * i + 2] * cov[6 + k];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double sv = (JC[3 * i] * q.J[3 * k] + JC[3 * i + 1] * q.J[3 * k + 1]) + JC[3 * i + 2] * q.J[3 * k + 2];
            var[2 * i + k] = (i == k) ? sv + sigma * sigma : sv;
        }
    pu_mat22_inv(var, ivar);
    const double dx = rm0 - mx, dy = rm1 - my;
    return dx * (ivar[0] * dx + ivar[1] * dy) + dy * (ivar[2] * dx + ivar[3] * dy) > 1.0;  // :723
}
__global__ __launch_bounds__(256) void k_register_mergability(MgArgs A) {
    extern __shared__ double mg_pose[];  // [nHist][12]
    CS_POSE_STREAM_PRIO();
    const int c = A.cam0 + blockIdx.y, tid = threadIdx.x, N = A.N, H = A.H;
    const cs_poseupdate_cam& C = A.cam[c];
    const double* hR = A.histR + (size_t)c * H * 9;
    const double* hT = A.histT + (size_t)c * H * 3;
    const double* hXY = A.histXY + (size_t)c * H * 2 * N;
    for (int q = tid; q < A.nHist * 12; q += 256) {
        const int j = q / 12, e = q - 12 * j, rs = (A.head - j + H) % H;
        mg_pose[q] = e < 9 ? hR[(size_t)rs * 9 + e] : hT[(size_t)rs * 3 + (e - 9)];
    }
    __syncthreads();
    const int lane = tid & 63, r = lane % MG_LPC, g = lane / MG_LPC;
    const int p = (blockIdx.x * 256 + tid) / MG_LPC;
    const bool live = p < A.P;
    const int s = live ? A.slot[(size_t)p * A.nCams + c] : -1;
    bool fail = false, cut = false;
    if (s >= 0) {
        double M[3], cov[9];
#pragma unroll
        for (int q = 0; q < 3; ++q) M[q] = A.M[3 * (size_t)p + q];
#pragma unroll
        for (int q = 0; q < 9; ++q) cov[q] = A.cov[9 * (size_t)p + q];
        const int f1 = C.trackSpan[s], f2 = C.trackSpan[N + s];
        const int len = f1 >= 0 ? f2 - f1 + 1 : 0;
        const int depth = len < A.nHist ? len : A.nHist;
        // the reference walks the whole preFrame chain: a track longer than the ring cannot be judged (verdict 2, never attached) -- and
        // is not walked at all: whatever the frames the ring holds say, nothing reads it (in a long run that is most candidates: old,
        // unmapped tracks; the walk over them was most of this kernel's time)
        cut = len > A.nHist;
        for (int j = r; j < depth && !fail && !cut; j += MG_LPC) {
            const double* R = mg_pose + 12 * j;
            const double* t = R + 9;
            const int rs = (A.head - j + H) % H;
            const double mx = hXY[(size_t)rs * 2 * N + s], my = hXY[(size_t)rs * 2 * N + N + s];
            fail = mg_term_fails(C.K, R, t, M, cov, A.sigma, mx, my);
        }
    }
    const unsigned long long b = __builtin_amdgcn_ballot_w64(fail);
    if (live && r == 0) {
        const bool anyFail = ((b >> (MG_LPC * g)) & ((1ull << MG_LPC) - 1ull)) != 0ull;
        A.out[(size_t)p * A.nCams + c] = s < 0 ? 255 : (cut ? 2 : (anyFail ? 0 : 1));
    }
}


// ---- staticCheckMergability over WHOLE tracks as a running verdict --------------------------------------------------------------
// The reference walks a candidate's track to its first frame (:715-729); tracks live for hundreds of frames, the candidates of a
// frame are ~10 k, and re-walking all of it every frame would cost more than the rest of the frame.  Here the walk is split at
// W = the history's walk depth (64 frames):
//   the WINDOW -- the newest W frames of the track -- is walked every frame with the point, covariance and poses as they stand
//                 (these are the frames a bundle adjustment still moves: a window BA's first key frame is < W frames old);
//   the TAIL   -- every older frame, back to the track's first -- has its verdict CACHED per (map point, camera): which slot, which
//                 track (its first frame), up to which frame the tail has been judged, the AND of those terms, and the point's
//                 position they were judged with.  A frame that crosses from the window into the tail adds its one term (judged
//                 with the point as it stands then; its pose is final by then).  The cache is dropped -- and the whole tail walked
//                 again from the kept frames (the ring stores storeLen >> W frames: cs_track_history_create_ex) -- when the
//                 candidate is another slot, the slot's track restarted, or the point has moved by more than tolPix pixels in
//                 this camera's image since the tail was judged (|M - Mref| fx / z: every cached term is then at most that far
//                 from what a fresh walk would test, against a gate of sigma = 10 px).
// With the point and the poses held still (the golden tracks of the reference's own function, 200-420 frames) the verdict is the
// reference's, term for term, whichever way the frames arrived.  A track whose first frame has left even the store: verdict 2.
struct MgCache {   // 48 bytes, zero = empty
    int slot1;     // candidate slot + 1
    int f1;        // the track's first frame
    int upto;      // the tail covers frames f1 .. upto
    int ok;        // AND of the tail's terms
    double M[3];   // the point the terms were judged with
    int epoch;     // the history's tail-rewrite epoch the terms were judged in (see MgRunArgs::epoch)
    int pad;
};
static_assert(sizeof(MgCache) == 48, "cs_register_mergability_cache_bytes");
struct MgRunArgs {
    MgArgs a;
    int W, count, curFrame;  // window depth, frames the ring holds, frame number of the ring's head
    double tolPix;
    MgCache* cache;          // [P][nCams]
    // poses of TAIL frames rewritten behind a cached verdict (an apply whose window reaches further back than the walk depth: the key frames the
    // decision places can lie hundreds of frames apart): the history counts such rewrites (epoch) and keeps the oldest frame any of them touched
    // (fromMin).  A cached tail stands if it was judged in this epoch or ends before fromMin; else it is walked again.  (The fixed cadence never
    // rewrites a tail frame: epoch stays 0.  ADVICE r05.)
    int epoch, fromMin;
    int* counts;             // [4] or null: cache hits, full tail walks, verdicts 2, tail terms evaluated
    const int* list;         // null, or the rows to judge: list[0 .. nList), entries < 0 skipped
    int nList;
    const int* flags;        // null, or the search's flags [P][nCams]: a candidate that already CARRIES a map point (bit 0 clear) is not
                             // judged (verdict 0): the registration walks end at it (:789-790) or ask checkUnify, never this test
};
constexpr int MG_GAP_MAX = 2 * MG_LPC;   // a cached tail this many frames behind is caught up by the candidate's own 8 lanes; further: a wave
struct MgMiss {   // a tail that a whole WAVE walks (phase B)
    int p, s, start, end;
};
// Phase A, 8 lanes per candidate: the WINDOW (the newest W frames, exact) first -- a candidate that fails there is done: no tail is
// looked at, none is built; then the tail: a cache hit is extended by the frames that crossed over since (usually one); anything
// longer -- no entry, another slot, another track, a point that moved, a cache that fell behind -- goes on the block's list.
// Phase B, a WAVE per listed tail: 64 lanes stride over its frames (a 900-frame tail is 15 steps deep instead of 113), any failure
// ends it.  The verdict and the cache entry are written by whoever finished the candidate.
__global__ __launch_bounds__(256) void k_register_mergability_running(MgRunArgs B) {
    extern __shared__ double mg_pose[];  // [W][12]
    __shared__ MgMiss missList[256 / MG_LPC];
    __shared__ int nMiss;
    CS_POSE_STREAM_PRIO();
    const MgArgs& A = B.a;
    const int c = A.cam0 + blockIdx.y, tid = threadIdx.x, N = A.N, H = A.H, W = B.W;
    // (a compact list is padded with -1: a workgroup whose first row lies behind the list's end has nothing to do -- before it stages a pose)
    if (B.list && (int)(blockIdx.x * (256 / MG_LPC)) < B.nList && B.list[blockIdx.x * (256 / MG_LPC)] < 0) return;
    const cs_poseupdate_cam& C = A.cam[c];
    const double* hR = A.histR + (size_t)c * H * 9;
    const double* hT = A.histT + (size_t)c * H * 3;
    const double* hXY = A.histXY + (size_t)c * H * 2 * N;
    for (int q = tid; q < W * 12; q += 256) {
        const int j = q / 12, e = q - 12 * j, rs = (A.head - j + H) % H;
        mg_pose[q] = e < 9 ? hR[(size_t)rs * 9 + e] : hT[(size_t)rs * 3 + (e - 9)];
    }
    if (tid == 0) nMiss = 0;
    __syncthreads();
    const int lane = tid & 63, r = lane % MG_LPC, g = lane / MG_LPC;
    const unsigned long long gmask = ((1ull << MG_LPC) - 1ull) << (MG_LPC * g);
    const int jj = (blockIdx.x * 256 + tid) / MG_LPC;
    const int p = B.list ? (jj < B.nList ? B.list[jj] : -1) : (jj < A.P ? jj : -1);
    const bool live = p >= 0 && p < A.P;
    int s = live ? A.slot[(size_t)p * A.nCams + c] : -1;
    bool skipped = false;
    if (s >= 0 && B.flags && !(B.flags[(size_t)p * A.nCams + c] & 1)) skipped = true, s = -2;   // the candidate carries a map point
    bool failW = false, cut = false, tailOK = true, hit = false, listed = false;
    int nTerms = 0;
    if (s >= 0) {
        double M[3], cov[9];
#pragma unroll
        for (int q = 0; q < 3; ++q) M[q] = A.M[3 * (size_t)p + q];
#pragma unroll
        for (int q = 0; q < 9; ++q) cov[q] = A.cov[9 * (size_t)p + q];
        const int f1 = C.trackSpan[s], f2 = C.trackSpan[N + s];
        const int len = f1 >= 0 ? f2 - f1 + 1 : 0;
        const int depth = len < W ? len : W;
        // ---- the window
        for (int j = r; j < depth && !failW; j += MG_LPC) {
            const double* R = mg_pose + 12 * j;
            const double* t = R + 9;
            const int rs = (A.head - j + H) % H;
            const double mx = hXY[(size_t)rs * 2 * N + s], my = hXY[(size_t)rs * 2 * N + N + s];
            failW = mg_term_fails(C.K, R, t, M, cov, A.sigma, mx, my);
        }
        failW = (__builtin_amdgcn_ballot_w64(failW) & gmask) != 0ull;
        // ---- the tail: frames f1 .. end (walk depths W .. len - 1)
        const int end = B.curFrame - W;
        if (len > W) {
            MgCache* E = B.cache + (size_t)p * A.nCams + c;
            const MgCache e = *E;   // (the group's lanes read the same 48 bytes)
            // how far has the point's projection in THIS camera moved since the tail was judged?
            const PuProj q0 = pu_project(C.K, mg_pose, mg_pose + 9, M), q1 = pu_project(C.K, mg_pose, mg_pose + 9, e.M);
            const double du = q0.u / q0.w - q1.u / q1.w, dv = q0.v / q0.w - q1.v / q1.w;
            hit = e.slot1 == s + 1 && e.f1 == f1 && e.upto >= f1 - 1 && e.upto <= end && q0.w > 0 && q1.w > 0 &&
                  du * du + dv * dv <= B.tolPix * B.tolPix && (e.epoch == B.epoch || e.upto < B.fromMin);
            const int start = hit ? e.upto + 1 : f1;
            tailOK = hit ? e.ok != 0 : true;
            const bool shortGap = end - start + 1 <= MG_GAP_MAX;
            // a cached tail is kept up to date by its one new term per frame whatever the window says (a stable candidate then never needs
            // the store); anything longer is only worth walking for a candidate whose window passed
            bool wrote = false;
            if (start > end) {
                wrote = hit;
            } else if (!tailOK) {
                wrote = true;   // a failed tail stays failed: only its reach moves on
            } else if (B.curFrame - start >= B.count) {
                cut = !failW;   // the frames to judge have left even the store
            } else if (shortGap && (hit || !failW)) {
                for (int f = start; f <= end && tailOK; f += MG_LPC) {
                    bool fail = false;
                    const int ff = f + r;
                    if (ff <= end) {
                        const int j = B.curFrame - ff, rs = ((A.head - j) % H + H) % H;
                        double Rt[12];
#pragma unroll
                        for (int q = 0; q < 9; ++q) Rt[q] = hR[(size_t)rs * 9 + q];
#pragma unroll
                        for (int q = 0; q < 3; ++q) Rt[9 + q] = hT[(size_t)rs * 3 + q];
                        const double mx = hXY[(size_t)rs * 2 * N + s], my = hXY[(size_t)rs * 2 * N + N + s];
                        fail = mg_term_fails(C.K, Rt, Rt + 9, M, cov, A.sigma, mx, my);
                        ++nTerms;
                    }
                    if (__builtin_amdgcn_ballot_w64(fail) & gmask) tailOK = false;
                }
                wrote = true;
            } else if (!failW) {
                listed = true;
                if (r == 0) {
                    const int k = atomicAdd(&nMiss, 1);
                    missList[k] = MgMiss{p, s, start, end};
                }
            }
            if (wrote && r == 0) {
                MgCache w;
                w.slot1 = s + 1, w.f1 = f1, w.upto = end, w.ok = tailOK ? 1 : 0;
                w.M[0] = hit ? e.M[0] : M[0], w.M[1] = hit ? e.M[1] : M[1], w.M[2] = hit ? e.M[2] : M[2], w.epoch = B.epoch, w.pad = 0;
                *E = w;
            }
        } else if (len > 0 && r == 0) {
            // the track still fits the window: an empty tail, judged with the point as it stands
            MgCache w;
            w.slot1 = s + 1, w.f1 = f1, w.upto = f1 - 1, w.ok = 1;
            w.M[0] = M[0], w.M[1] = M[1], w.M[2] = M[2], w.epoch = B.epoch, w.pad = 0;
            B.cache[(size_t)p * A.nCams + c] = w;
        }
    }
    if (live && r == 0 && !listed) A.out[(size_t)p * A.nCams + c] = s == -1 ? 255 : (skipped ? 0 : (cut ? 2 : (failW || !tailOK ? 0 : 1)));
    __syncthreads();
    // ---- phase B: a wave per listed tail
    const int wv = tid >> 6, nM = nMiss;
    for (int k = wv; k < nM; k += 4) {
        const MgMiss m = missList[k];
        double M[3], cov[9];
#pragma unroll
        for (int q = 0; q < 3; ++q) M[q] = A.M[3 * (size_t)m.p + q];
#pragma unroll
        for (int q = 0; q < 9; ++q) cov[q] = A.cov[9 * (size_t)m.p + q];
        bool ok = true;
        for (int f = m.start; f <= m.end && ok; f += 64) {
            bool fail = false;
            const int ff = f + lane;
            if (ff <= m.end) {
                const int j = B.curFrame - ff, rs = ((A.head - j) % H + H) % H;
                double Rt[12];
#pragma unroll
                for (int q = 0; q < 9; ++q) Rt[q] = hR[(size_t)rs * 9 + q];
#pragma unroll
                for (int q = 0; q < 3; ++q) Rt[9 + q] = hT[(size_t)rs * 3 + q];
                const double mx = hXY[(size_t)rs * 2 * N + m.s], my = hXY[(size_t)rs * 2 * N + N + m.s];
                fail = mg_term_fails(C.K, Rt, Rt + 9, M, cov, A.sigma, mx, my);
                ++nTerms;
            }
            if (__builtin_amdgcn_ballot_w64(fail)) ok = false;
        }
        if (lane == 0) {
            MgCache* E = B.cache + (size_t)m.p * A.nCams + c;
            // (a tail caught up from a cached prefix keeps the point the prefix was judged with; a fresh walk: the point as it stands)
            const bool fresh = m.start == C.trackSpan[m.s];
            MgCache w;
            w.slot1 = m.s + 1, w.f1 = C.trackSpan[m.s], w.upto = m.end, w.ok = ok ? 1 : 0;
            w.M[0] = fresh ? M[0] : E->M[0], w.M[1] = fresh ? M[1] : E->M[1], w.M[2] = fresh ? M[2] : E->M[2], w.epoch = B.epoch, w.pad = 0;
            *E = w;
            A.out[(size_t)m.p * A.nCams + c] = ok ? 1 : 0;   // (its window passed: that is why it was listed)
        }
    }
    if (B.counts) {
        const unsigned long long bh = __builtin_amdgcn_ballot_w64(hit && r == 0), bw = __builtin_amdgcn_ballot_w64(listed && !hit && r == 0),
                                 bc = __builtin_amdgcn_ballot_w64(cut && r == 0);
        int terms = nTerms;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) terms += __shfl_xor(terms, o, 64);
        if (lane == 0) {
            if (bh) atomicAdd(B.counts, __popcll(bh));
            if (bw) atomicAdd(B.counts + 1, __popcll(bw));
            if (bc) atomicAdd(B.counts + 2, __popcll(bc));
            if (terms) atomicAdd(B.counts + 3, terms);
        }
    }
}


// ---- RobustBundleRTS::updateNewPosesPoints (src/app/SL_CoSLAMRobustBA.cpp:248-271) -------------------------------------------------
// Behind a bundle adjustment + relaxation every map point seen after the window's first key frame is triangulated again from the
// moved poses: updateStaticPointPosition (src/slam/SL_CoSLAMHelper.cpp:338-394) takes, per camera that holds a feature of the
// point, that feature and the feature of the same track whose camera centre subtends the largest angle with the current one at
// the point (first of equal angles in the backward walk; an angle of 0 never), updateDynamicPointPosition (:455-484) this frame's
// features only (at least one of them TYPE_FEATPOINT_DYNAMIC); then triangulateMultiView over the views' normalised points and
// getTriangulateCovMat at the new point.  The reference walks FeaturePoint::preFrame lists per point on the host; here a WAVE
// = one map point: the lanes take the frames of the backward walk of a camera's track (poses from the history ring -- which therefore has to
// hold the poses AS ADJUSTED: cs_track_history_set_poses_dev), compare COSINES (acos is monotone; no libm on the device or in the
// oracle's matching mode) and fold to the walk's first minimum; the 3x3 normal equations are summed view by view in the
// reference's order and solved by cofactors.  getCameraCenter, getAbsRadiansBetween, normPoint, triangulateMultiView,
// getTriangulateCovMat are un-vendored LibVisualSLAM: definitions in DESIGN.md 3.9 (the oracle's, operation for operation).
struct UpArgs {
    int nCams, N, nMap, H, head, nHist, firstKeyFrame;
    const int* pointFeat;            // [nMap][nCams]
    const int* lastFrame;            // [nMap] or null
    const unsigned char* isCurrent;  // [nMap] or null
    const double* histXY;
    const double* histR;
    const double* histT;
    double* mapPts;
    double* mapCov;
    const unsigned char* mapFlags;
    double sigma;
    int* counts;  // [2] static / dynamic points re-triangulated, or null
    const double* cen;  // [nCams][nHist][3] camera centres by walk depth (0 = this frame): k_ring_centres, once per launch
    int refine;                   // CoSLAM::refineMapPoint: the points `select` names, whatever their type, no frame test
    const unsigned char* select;  // [nMap] or null (= all)
    // the features as REFERENCES (cs_feat_ref: MapPoint::pFeatures[c] of whatever age, with its preFrame chain) instead of pointFeat
    const int4* featRef;              // [nMap][nCams] {slot, frame, first, seg} or null
    const unsigned char* refStatic;   // [nMap][nCams] the features' types in their own frames, or null (= isStatic of the slot)
    const int4* segPool;              // [nCams][segCap] {slot, last, first, next}
    int segCap, curFrame, stored;     // stored: frames the ring holds (a node further back ends a walk)
    cs_poseupdate_cam cam[PU_MAX_CAMS];
};
constexpr int UP_LPP = 64;  // a WAVE per map point

static_assert(UP_LPP == 64 && PU_MAX_CAMS <= UP_LPP, "the covariance tail broadcasts from lane = camera of the point's own wave");

struct UpNormalEq {
    double N[6], g[3];
};
__device__ __forceinline__ void up_add_view(UpNormalEq& E, const double* __restrict__ iK, const double* __restrict__ R,
                                            const double* __restrict__ t, double mx, double my) {
    const double w = (iK[6] * mx + iK[7] * my) + iK[8];
    const double x = ((iK[0] * mx + iK[1] * my) + iK[2]) / w, y = ((iK[3] * mx + iK[4] * my) + iK[5]) / w;  // normPoint
    const double a0[3] = {R[0] - x * R[6], R[1] - x * R[7], R[2] - x * R[8]}, a1[3] = {R[3] - y * R[6], R[4] - y * R[7], R[5] - y * R[8]};
    const double b0 = x * t[2] - t[0], b1 = y * t[2] - t[1];
    E.N[0] = E.N[0] + (a0[0] * a0[0] + a1[0] * a1[0]);
    E.N[1] = E.N[1] + (a0[0] * a0[1] + a1[0] * a1[1]);
    E.N[2] = E.N[2] + (a0[0] * a0[2] + a1[0] * a1[2]);
    E.N[3] = E.N[3] + (a0[1] * a0[1] + a1[1] * a1[1]);
    E.N[4] = E.N[4] + (a0[1] * a0[2] + a1[1] * a1[2]);
    E.N[5] = E.N[5] + (a0[2] * a0[2] + a1[2] * a1[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) E.g[q] = E.g[q] + (a0[q] * b0 + a1[q] * b1);
}
__device__ __forceinline__ void up_add_jtj(double* S, const double* J) {
    S[0] = S[0] + (J[0] * J[0] + J[3] * J[3]);
    S[1] = S[1] + (J[0] * J[1] + J[3] * J[4]);
    S[2] = S[2] + (J[0] * J[2] + J[3] * J[5]);
    S[3] = S[3] + (J[1] * J[1] + J[4] * J[4]);
    S[4] = S[4] + (J[1] * J[2] + J[4] * J[5]);
    S[5] = S[5] + (J[2] * J[2] + J[5] * J[5]);
}
// symmetric 3x3 {n00, n01, n02, n11, n12, n22}: cofactors (same order) and the determinant
__device__ __forceinline__ double up_sym33_cof(const double* N, double* c) {
    c[0] = N[3] * N[5] - N[4] * N[4];
    c[1] = N[2] * N[4] - N[1] * N[5];
    c[2] = N[1] * N[4] - N[2] * N[3];
    c[3] = N[0] * N[5] - N[2] * N[2];
    c[4] = N[1] * N[2] - N[0] * N[4];
    c[5] = N[0] * N[3] - N[1] * N[1];
    return (N[0] * c[0] + N[1] * c[1]) + N[2] * c[2];
}
__device__ __forceinline__ void up_cam_center(const double* __restrict__ R, const double* __restrict__ t, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i) C[i] = -((R[i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2]);
}

// ---- walks along FeaturePoint::preFrame ------------------------------------------------------------------------------------------------------
// A point's feature in a camera as the reference holds it: MapPoint::pFeatures[c] -- of this frame, or an older one when the camera lost
// the point (nothing clears the pointer) -- and the chain behind it: consecutive frames on the feature's own slot, frame - 1 .. first,
// then the segments the registration loops linked behind it (`pFeat->preFrame = p->pFeatures[iCam]`, src/app/SL_CoSLAM.cpp:775-779,
// :997-1000: the point's OLD chain hangs behind the feature of the new track, whose own earlier frames drop out).  Pixels and poses of a
// node are the ring's entry of its frame.  The walks are bounded by `cap` NODES (the reference's are not) and end at a node older than the
// ring.  The plain tables (pointFeat + trackSpan) are the special case {slot, this frame, the track's first frame, no segment}.
struct ChainCtx {
    int N, H, head, cap, curFrame, stored, segCap, nCen;
    int minFrame;          // the walk ends at the first node before this frame (isStaticPoint's window; INT_MIN: no window)
    const double* cen;     // [nCams][nCen][3] camera centres of ring depths < nCen
    const int4* segPool;   // or null
};
// the widest-parallax node behind `ref` around point M seen from centre C0 (a = C0 - M, na = |a|^2): the smallest cosine, the first of
// equal ones in walk order, never an angle of 0 (cosine 1).  Wave-cooperative (lane r takes every 64th node of a segment); every lane
// returns the node's ring depth (-1: none) and its slot.
// W lanes work together (a group of W consecutive lanes, W a power of two; r = the lane's index in its group): W = 64 is the wave.
__device__ __forceinline__ int chain_widest_w(const ChainCtx& X, int c, int4 ref, const double* __restrict__ hR, const double* __restrict__ hT,
                                              const double* a, double na, const double* M, int r, int& bestSlot, int W) {
    int best = -1, bestK = 0x7fffffff, bSlot = -1;
    double bestCos = 1.0;
    int slot = ref.x, hi = ref.y - 1, lo = ref.z, seg = ref.w, k0 = 1;   // (node 0 is the feature itself)
    for (;;) {
        const int ringOldest = X.curFrame - X.stored + 1;   // the oldest frame the ring holds
        const int oldest = ringOldest > X.minFrame ? ringOldest : X.minFrame;
        int cnt = hi - (lo > oldest ? lo : oldest) + 1;
        const bool cut = lo < oldest;
        if (cnt > X.cap - k0) cnt = X.cap - k0;
        for (int i = r; i < cnt; i += W) {
            const int j = X.curFrame - (hi - i);
            double Cj[3];
            if (j < X.nCen) {
                const double* q = X.cen + 3 * ((size_t)c * X.nCen + j);
                Cj[0] = q[0], Cj[1] = q[1], Cj[2] = q[2];
            } else {
                const int rs = ((X.head - j) % X.H + X.H) % X.H;
                up_cam_center(hR + (size_t)rs * 9, hT + (size_t)rs * 3, Cj);
            }
            const double b0 = Cj[0] - M[0], b1 = Cj[1] - M[1], b2 = Cj[2] - M[2];
            const double d = (a[0] * b0 + a[1] * b1) + a[2] * b2;
            const double nb = (b0 * b0 + b1 * b1) + b2 * b2;
            const double cv = d / sqrt(na * nb);
            if (cv < bestCos) bestCos = cv, best = j, bestK = k0 + i, bSlot = slot;   // (i ascending: the first of equal cosines stays)
        }
        if (cnt > 0) k0 += cnt;
        if (cut || k0 >= X.cap || seg < 0 || !X.segPool || seg >= X.segCap) break;
        const int4 g = X.segPool[(size_t)c * X.segCap + seg];
        slot = g.x, hi = g.y, lo = g.z, seg = g.w;
    }
    for (int off = 1; off < W; off <<= 1) {
        const double oc = __shfl_xor(bestCos, off, 64);
        const int oj = __shfl_xor(best, off, 64), ok = __shfl_xor(bestK, off, 64), os = __shfl_xor(bSlot, off, 64);
        if (oj >= 0 && (oc < bestCos || (oc == bestCos && ok < bestK))) bestCos = oc, best = oj, bestK = ok, bSlot = os;
    }
    bestSlot = bSlot;
    return best;
}
__device__ __forceinline__ int chain_widest(const ChainCtx& X, int c, int4 ref, const double* __restrict__ hR, const double* __restrict__ hT,
                                            const double* a, double na, const double* M, int r, int& bestSlot) {
    return chain_widest_w(X, c, ref, hR, hT, a, na, M, r, bestSlot, 64);
}
// the reference of map point m in camera c: the table's, or the feature of this frame pointFeat names with its slot's track behind it
// (frames then count from curFrame = 0: only differences are used)
__device__ __forceinline__ int4 chain_ref(const int4* featRef, const int* pointFeat, const cs_poseupdate_cam& C, int N, int nCams, int m, int c) {
    if (featRef) return featRef[(size_t)m * nCams + c];
    const int s = pointFeat[(size_t)m * nCams + c];
    int4 ref = make_int4(s, 0, 0, -1);
    if (s >= 0) {
        const int f1 = C.trackSpan[s], f2 = C.trackSpan[N + s];
        if (f1 >= 0) ref.z = -(f2 - f1);
    }
    return ref;
}

// the camera centres of all (camera, ring entry) pairs by walk depth: every point's walk reads the same nCams x nHist of them
__global__ __launch_bounds__(256) void k_ring_centres(int nCams, int H, int head, int nHist, const double* __restrict__ hR,
                                                      const double* __restrict__ hT, double* __restrict__ cen) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nCams * nHist) return;
    const int c = q / nHist, j = q - c * nHist, rs = (head - j + H) % H;
    up_cam_center(hR + ((size_t)c * H + rs) * 9, hT + ((size_t)c * H + rs) * 3, cen + 3 * (size_t)q);
}

// one map point, one wave (lane r): every test below is uniform over the point's lanes, the shuffles stay inside the wave
__device__ __forceinline__ void up_point(const UpArgs& A, int m, int r) {
    if (m >= A.nMap) return;
    if (A.refine) {
        if (A.select && !A.select[m]) return;
    } else if (A.lastFrame) {
        if (A.lastFrame[m] <= A.firstKeyFrame) return;  // :250, :261
    } else if (A.featRef) {  // MapPoint::lastFrame = the newest frame a camera saw the point in (propagateFeatureStates, SL_SingleSLAM.cpp:51)
        int lastF = -0x7fffffff;
        for (int c = 0; c < A.nCams; ++c) {
            const int4 q = A.featRef[(size_t)m * A.nCams + c];
            if (q.x >= 0 && q.y > lastF) lastF = q.y;
        }
        if (lastF <= A.firstKeyFrame) return;
    }
    const unsigned char fl = A.refine ? 0 : A.mapFlags[m];  // (refineMapPoint asks nothing about the point's type)
    const bool locStatic = (fl & (CS_MAP_DYNAMIC | CS_MAP_FALSE)) == 0;               // isLocalStatic()
    const bool locDynamic = (fl & (CS_MAP_DYNAMIC | CS_MAP_FALSE)) == CS_MAP_DYNAMIC;  // isLocalDynamic()
    const bool cur = A.isCurrent ? A.isCurrent[m] != 0 : true;
    if (!(locStatic || (locDynamic && cur))) return;  // (:266-269: the active list's dynamic points are never reached)
    const int N = A.N, H = A.H;
    double M[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) M[q] = A.mapPts[3 * (size_t)m + q];
    UpNormalEq E;
#pragma unroll
    for (int q = 0; q < 6; ++q) E.N[q] = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) E.g[q] = 0;
    int numView = 0, nDynamic = 0;
    int mySecond = -2, myFirst = 0;  // lane r keeps camera r's views: ring depths; second -1 none, -2 the camera holds no feature of the point
    ChainCtx X;
    X.N = N, X.H = H, X.head = A.head, X.cap = A.nHist, X.curFrame = A.featRef ? A.curFrame : 0, X.stored = A.featRef ? A.stored : A.nHist;
    X.segCap = A.segCap, X.nCen = A.nHist, X.cen = A.cen, X.segPool = A.segPool;
    X.minFrame = -2147483647 - 1;
    for (int c = 0; c < A.nCams; ++c) {
        const cs_poseupdate_cam& C = A.cam[c];
        const int4 ref = chain_ref(A.featRef, A.pointFeat, C, N, A.nCams, m, c);
        const int s = ref.x, j0 = X.curFrame - ref.y;
        if (s < 0 || j0 >= X.stored) continue;   // (a feature older than the ring is no view)
        const double* hR = A.histR + (size_t)c * H * 9;
        const double* hT = A.histT + (size_t)c * H * 3;
        const double* hXY = A.histXY + (size_t)c * H * 2 * N;
        const int rs0 = (A.head - j0 + H) % H;
        const double* R0 = hR + (size_t)rs0 * 9;
        const double* t0 = hT + (size_t)rs0 * 3;
        up_add_view(E, C.iK, R0, t0, hXY[(size_t)rs0 * 2 * N + s], hXY[(size_t)rs0 * 2 * N + N + s]);  // :347-356 / :463-470
        ++numView;
        int best = -1;
        if (locStatic) {
            double C0[3];
            up_cam_center(R0, t0, C0);
            const double a[3] = {C0[0] - M[0], C0[1] - M[1], C0[2] - M[2]};
            const double na = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
            int bs = s;
            best = chain_widest(X, c, ref, hR, hT, a, na, M, r, bs);  // :362-373 fp = fp->preFrame
            if (best >= 0) {  // :374-383
                const int rs = (A.head - best + H) % H;
                up_add_view(E, C.iK, hR + (size_t)rs * 9, hT + (size_t)rs * 3, hXY[(size_t)rs * 2 * N + bs], hXY[(size_t)rs * 2 * N + N + bs]);
                ++numView;
            }
        } else if (A.refStatic ? !A.refStatic[(size_t)m * A.nCams + c] : !C.isStatic[s])
            ++nDynamic;  // :471-472
        if (r == c) mySecond = best, myFirst = j0;
    }
    if (numView < 2 || (!locStatic && nDynamic < 1)) return;  // :388, :475
    double cf[6];
    const double det = up_sym33_cof(E.N, cf);
    M[0] = ((cf[0] * E.g[0] + cf[1] * E.g[1]) + cf[2] * E.g[2]) / det;  // triangulateMultiView
    M[1] = ((cf[1] * E.g[0] + cf[3] * E.g[1]) + cf[4] * E.g[2]) / det;
    M[2] = ((cf[2] * E.g[0] + cf[4] * E.g[1]) + cf[5] * E.g[2]) / det;
    // getTriangulateCovMat over the same views: lane c computes camera c's (up to) two Jacobians at the new point -- the divisions
    // are the expensive part --, then every lane sums the J^T J blocks in the reference's view order from lane c's registers
    double J1[6] = {0, 0, 0, 0, 0, 0}, J2[6] = {0, 0, 0, 0, 0, 0};
    if (r < A.nCams && mySecond != -2) {
        const double* hR = A.histR + (size_t)r * H * 9;
        const double* hT = A.histT + (size_t)r * H * 3;
        const int rs1 = (A.head - myFirst + H) % H;
        const PuProj q1 = pu_project(A.cam[r].K, hR + (size_t)rs1 * 9, hT + (size_t)rs1 * 3, M);
#pragma unroll
        for (int k = 0; k < 6; ++k) J1[k] = q1.J[k];
        if (mySecond >= 0) {
            const int rs = (A.head - mySecond + H) % H;
            const PuProj q2 = pu_project(A.cam[r].K, hR + (size_t)rs * 9, hT + (size_t)rs * 3, M);
#pragma unroll
            for (int k = 0; k < 6; ++k) J2[k] = q2.J[k];
        }
    }
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < A.nCams; ++c) {
        const int sv = __shfl(mySecond, c, 64);
        if (sv == -2) continue;
        double Jc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) Jc[k] = __shfl(J1[k], c, 64);
        up_add_jtj(S, Jc);
        if (sv >= 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) Jc[k] = __shfl(J2[k], c, 64);
            up_add_jtj(S, Jc);
        }
    }
    if (r != 0) return;
    const double dS = up_sym33_cof(S, cf), s2 = A.sigma * A.sigma;
    double* cov = A.mapCov + 9 * (size_t)m;
    const double c01 = (cf[1] / dS) * s2, c02 = (cf[2] / dS) * s2, c12 = (cf[4] / dS) * s2;
    cov[0] = (cf[0] / dS) * s2, cov[1] = c01, cov[2] = c02;
    cov[3] = c01, cov[4] = (cf[3] / dS) * s2, cov[5] = c12;
    cov[6] = c02, cov[7] = c12, cov[8] = (cf[5] / dS) * s2;
#pragma unroll
    for (int q = 0; q < 3; ++q) A.mapPts[3 * (size_t)m + q] = M[q];
    if (A.counts) atomicAdd(A.counts + (locStatic ? 0 : 1), 1);
}
__global__ __launch_bounds__(256) void k_update_points(UpArgs A) {
    const int tid = threadIdx.x, g = tid / UP_LPP, r = tid % UP_LPP;
    up_point(A, blockIdx.x * (256 / UP_LPP) + g, r);
}

// ---- CoSLAM::checkUnify (src/app/SL_CoSLAM.cpp:561-665) for a batch of pairs ----------------------------------------------------------
// What the registration loops ask when a point's nearest feature already carries ANOTHER static point (bMerge, every 50th frame,
// :791-796): can the two be one?  The views of both points -- per camera, point 1 then point 2, each its feature of this frame and the
// widest-parallax one further back on that track (around the point's OWN position) -- triangulated together, the covariance, and
// every view within Mahalanobis distance 1 of the re-projection.  A wave per pair, as k_update_points: the walk for the second view
// split over the lanes, the normal equations summed in the reference's view order by every lane, then lane v owns view v for the
// Jacobian and the gate.  The gate's covariance term takes its rotation from `Rs + 3 * i` (:657: three doubles per view into an
// array that holds nine) -- restated as written: the views' rotations go to LDS back to back and view i reads nine from offset 3 i.
struct CuArgs {
    int nCams, N, H, head, nHist, nPairs;
    const int *pf1, *pf2;      // [nPairs][nCams] slot of the point's feature of this frame, < 0 none
    const double *M1, *M2;     // [nPairs][3]
    const double *histXY, *histR, *histT, *cen;
    double sigma;
    unsigned char* ok;         // [nPairs]
    double *M, *cov;           // [nPairs][3], [nPairs][9]
    // the two points' features as references (rows of cs_feat_ref, see ChainCtx) instead of pf1 / pf2
    const int4 *ref1, *ref2;   // [nPairs][nCams] or null
    const int4* segPool;
    int segCap, curFrame, stored;
    cs_poseupdate_cam cam[PU_MAX_CAMS];
};
// one pair, one wave: pf1 / pf2 the two points' rows of pointFeat, M1 / M2 their positions, sRw the wave's 64 * 9 + 16 doubles of LDS;
// returns the verdict (uniform over the wave), M / cov the unified point (every lane)
// The (camera, point) pairs of a call -- 2 nCams of them, in the order the reference walks them: camera by camera, point 1 then point 2 --
// are worked on SIDE BY SIDE: a group of LPP = 64 / (2 nCams) lanes (a power of two, at most 8) per pair reads the pair's feature and
// walks its chain for the widest-parallax second view (chain_widest_w over the group), the group's first two lanes then hold the pair's
// (up to) two views: each works out its view's terms of the normal equations, and the sums are taken in the reference's view order from
// those lanes' registers -- the same additions in the same order as one pair after the other (which is what a single wave did until
// round 6: sixteen dependent rounds of loads, ~27 us a call, most of a bMerge walk's 5 ms).  View v then belongs to
// the lane that holds it (myV) for the Jacobian, the flat rotation array and the gate.
__device__ __forceinline__ void up_view_terms(const double* __restrict__ iK, const double* __restrict__ R, const double* __restrict__ t,
                                              double mx, double my, double (&q)[9]) {
    const double w = (iK[6] * mx + iK[7] * my) + iK[8];
    const double x = ((iK[0] * mx + iK[1] * my) + iK[2]) / w, y = ((iK[3] * mx + iK[4] * my) + iK[5]) / w;  // normPoint
    const double a0[3] = {R[0] - x * R[6], R[1] - x * R[7], R[2] - x * R[8]}, a1[3] = {R[3] - y * R[6], R[4] - y * R[7], R[5] - y * R[8]};
    const double b0 = x * t[2] - t[0], b1 = y * t[2] - t[1];
    q[0] = (a0[0] * a0[0] + a1[0] * a1[0]);
    q[1] = (a0[0] * a0[1] + a1[0] * a1[1]);
    q[2] = (a0[0] * a0[2] + a1[0] * a1[2]);
    q[3] = (a0[1] * a0[1] + a1[1] * a1[1]);
    q[4] = (a0[1] * a0[2] + a1[1] * a1[2]);
    q[5] = (a0[2] * a0[2] + a1[2] * a1[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) q[6 + k] = (a0[k] * b0 + a1[k] * b1);
}
__device__ __forceinline__ bool check_unify_wave(const CuArgs& A, const int* pf1, const int* pf2, const int4* rf1, const int4* rf2,
                                                 const double* M1, const double* M2, double* sRw, double (&M)[3], double (&cov)[9]) {
    const int r = threadIdx.x % 64;
    const int N = A.N, H = A.H, nPairs = 2 * A.nCams;
    int LPP = 8;
    while (LPP * nPairs > 64) LPP >>= 1;   // (nCams <= 16: LPP >= 2)
    const int pair = r / LPP, sub = r - pair * LPP;
    const bool mine = pair < nPairs;
    const int c = mine ? pair >> 1 : 0, which = pair & 1;
    ChainCtx X;
    X.N = N, X.H = H, X.head = A.head, X.cap = A.nHist, X.curFrame = rf1 ? A.curFrame : 0, X.stored = rf1 ? A.stored : A.nHist;
    X.segCap = A.segCap, X.nCen = A.nHist, X.cen = A.cen, X.segPool = A.segPool;
    X.minFrame = -2147483647 - 1;
    const cs_poseupdate_cam& C = A.cam[c];
    const double* hR = A.histR + (size_t)c * H * 9;
    const double* hT = A.histT + (size_t)c * H * 3;
    const double* hXY = A.histXY + (size_t)c * H * 2 * N;
    int4 ref = make_int4(-1, 0, 0, -1);
    if (mine) ref = rf1 ? (which ? rf2 : rf1)[c] : chain_ref(nullptr, which ? pf2 : pf1, C, N, 0, 0, c);
    const int s = ref.x, j0 = X.curFrame - ref.y;
    const bool valid = mine && s >= 0 && j0 < X.stored;
    const double* Mold = which ? M2 : M1;
    const int rs0 = valid ? (A.head - j0 + H) % H : 0;
    const double* R0 = hR + (size_t)rs0 * 9;
    const double* t0 = hT + (size_t)rs0 * 3;
    int best = -1, bs = s;
    {
        double C0[3] = {0, 0, 0};
        if (valid) up_cam_center(R0, t0, C0);
        const double a[3] = {C0[0] - Mold[0], C0[1] - Mold[1], C0[2] - Mold[2]};
        const double na = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
        // (an invalid pair walks an empty chain: the group's shuffles are executed by every lane)
        best = chain_widest_w(X, c, valid ? ref : make_int4(-1, 0, 1, -1), hR, hT, a, na, Mold, sub, bs, LPP);
        if (!valid) best = -1;
    }
    // the group's lane 0 holds the pair's first view, lane 1 the second (if there is one)
    const bool has = valid && (sub == 0 || (sub == 1 && best >= 0));
    const int myC = c, myJ = sub == 0 ? j0 : best, myS = sub == 0 ? s : bs;
    double q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double *Rv = nullptr, *tv = nullptr;
    if (has) {
        const int rs = (A.head - myJ + H) % H;
        Rv = hR + (size_t)rs * 9, tv = hT + (size_t)rs * 3;
        up_view_terms(C.iK, Rv, tv, hXY[(size_t)rs * 2 * N + myS], hXY[(size_t)rs * 2 * N + N + myS], q);
    }
    UpNormalEq E;
#pragma unroll
    for (int k = 0; k < 6; ++k) E.N[k] = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) E.g[k] = 0;
    const unsigned long long hasMask = __builtin_amdgcn_ballot_w64(has);
    int nv = 0, myV = -1;
    for (int pr = 0; pr < nPairs; ++pr) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int src = pr * LPP + w;
            if (!((hasMask >> src) & 1ull)) continue;   // (uniform)
#pragma unroll
            for (int k = 0; k < 6; ++k) E.N[k] = E.N[k] + __shfl(q[k], src, 64);
#pragma unroll
            for (int k = 0; k < 3; ++k) E.g[k] = E.g[k] + __shfl(q[6 + k], src, 64);
            if (r == src) myV = nv;
            ++nv;
        }
    }
    double cf[6];
    const double det = up_sym33_cof(E.N, cf);
    M[0] = ((cf[0] * E.g[0] + cf[1] * E.g[1]) + cf[2] * E.g[2]) / det;  // triangulateMultiView
    M[1] = ((cf[1] * E.g[0] + cf[3] * E.g[1]) + cf[4] * E.g[2]) / det;
    M[2] = ((cf[2] * E.g[0] + cf[4] * E.g[1]) + cf[5] * E.g[2]) / det;
    // the lane of view v: the view's pose, its Jacobian at M (getTriangulateCovMat), its rotation into the flat array
    double J[6] = {0, 0, 0, 0, 0, 0};
    PuProj pv;
    if (has) {
        pv = pu_project(A.cam[myC].K, Rv, tv, M);
#pragma unroll
        for (int k = 0; k < 6; ++k) J[k] = pv.J[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) sRw[9 * myV + k] = Rv[k];
    }
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int pr = 0; pr < nPairs; ++pr) {   // the J^T J blocks in view order
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int src = pr * LPP + w;
            if (!((hasMask >> src) & 1ull)) continue;
            double Jv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) Jv[k] = __shfl(J[k], src, 64);
            up_add_jtj(S, Jv);
        }
    }
    const double dS = up_sym33_cof(S, cf), s2 = A.sigma * A.sigma;
    cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
    cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
    cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
    __builtin_amdgcn_wave_barrier();   // (a wave's own LDS stores are visible to its own loads in program order)
    bool fail = false;
    if (has) {   // :652-661
        const double rm0 = pv.u / pv.w, rm1 = pv.v / pv.w;
        double Rq[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rq[k] = sRw[3 * myV + k];   // `Rs + 3 * i`
        const PuProj pq = pu_project(A.cam[myC].K, Rq, tv, M);
        double JC[6], var[4], ivar[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) JC[3 * i + k] = (pq.J[3 * i] * cov[k] + pq.J[3 * i + 1] * cov[3 + k]) + pq.J[3 * i + 2] * cov[6 + k];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double sv = (JC[3 * i] * pq.J[3 * k] + JC[3 * i + 1] * pq.J[3 * k + 1]) + JC[3 * i + 2] * pq.J[3 * k + 2];
                var[2 * i + k] = (i == k) ? sv + s2 : sv;
            }
        pu_mat22_inv(var, ivar);
        const double* px = hXY + (size_t)((A.head - myJ + H) % H) * 2 * N;
        const double dx = rm0 - px[myS], dy = rm1 - px[N + myS];
        fail = dx * (ivar[0] * dx + ivar[1] * dy) + dy * (ivar[2] * dx + ivar[3] * dy) > 1.0;
    }
    const bool anyFail = __builtin_amdgcn_ballot_w64(fail) != 0;
    __builtin_amdgcn_wave_barrier();   // (the next call of this wave overwrites sRw)
    return !anyFail;
}
__global__ __launch_bounds__(256) void k_check_unify(CuArgs A) {
    __shared__ double sR[4][64 * 9 + 16];
    const int tid = threadIdx.x, g = tid / 64, r = tid % 64;
    const int q = blockIdx.x * 4 + g;
    if (q >= A.nPairs) return;
    double M[3], cov[9];
    const bool ok = check_unify_wave(A, A.ref1 ? nullptr : A.pf1 + (size_t)q * A.nCams, A.ref1 ? nullptr : A.pf2 + (size_t)q * A.nCams,
                                     A.ref1 ? A.ref1 + (size_t)q * A.nCams : nullptr, A.ref1 ? A.ref2 + (size_t)q * A.nCams : nullptr, A.M1 + 3 * (size_t)q,
                                     A.M2 + 3 * (size_t)q, sR[g], M, cov);
    if (r == 0) {
        A.ok[q] = ok ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) A.M[3 * (size_t)q + k] = M[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) A.cov[9 * (size_t)q + k] = cov[k];
    }
}

// ---- CoSLAM::curStaticPointsRegInGroup with bMerge == true (src/app/SL_CoSLAM.cpp:854-898, 731-830; every 50th frame) ---------------------
// The walks in the reference's order, ONE wave, one point after the other: a walk that meets a feature of ANOTHER static point asks
// checkUnify -- with both points' features and positions as they stand at that moment -- and on a yes takes that point's place
// (:797-826: the position, the other point false, its features in the cameras up to the one of the conflict -- the loop reads
// `pFeat->mpt->pFeatures[v]`, and pFeat->mpt is the walking point once pFeat itself has moved: the features behind it stay), which ends
// the walk; on a no it goes on (with bMerge a mapped feature does not end a walk).  Sequential by nature (every verdict depends on what
// the walks before it attached); lane c prefetches camera c's entries of the point, the wave evaluates checkUnify together.  The parity
// mode's kernel: ~5 us per conflict, a frame of 8 cameras x 1500 points ~0.1 s -- for the frames that carry bMerge when the reference's
// run is wanted step for step; the frame loop's single pass does not unify points.
constexpr int DM_MAX_CAMS = 16;   // (cs_register_decide_merge_dev refuses more: lane c = camera c, four lanes' worth of columns)
__device__ __forceinline__ void wave_fence_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
struct DmArgs {
    CuArgs cu;                       // the history ring, cameras (slot2map is written), sigma
    int P, mapBase, onlyCam;
    const int* slot;                 // [P][nCams] the search's candidates
    const int* flags;                // [P][nCams] bit 1: the candidate is dynamic
    const unsigned char* mergeable;  // [P][nCams]
    unsigned char* mapFlags;         // [P] in / out (a point unified away becomes false)
    int* pointFeat;                  // [P][nCams] in / out
    double* mapPts;                  // [P][3] in / out (the survivor takes the unified position)
    double* mapCov;                  // [P][9]
    unsigned char* attached;         // [P][nCams] out
    unsigned char* regged;           // [P] out
    unsigned char* inVec;            // scratch [P]: the camera loop's visiting list, fixed when the loop starts (:864-869)
    int* counts;                     // [4] out: features attached, points registered, points unified away, checkUnify calls
    const int* list;                 // null, or the points to walk as a compact list in map order (entries < 0 behind it): the frame's current
    int nList;                       // points -- a point with a feature in a camera is one of them.  attached / regged are then cleared by the caller
    // by list only: checkUnify of every (listed point, camera) whose candidate carries ANOTHER static point, evaluated side by side BEFORE the
    // walk (k_merge_precheck) on the state the walk starts from; the walk takes a verdict from here as long as neither point has been
    // touched by an earlier step (inVec[point] != 0: it gained a feature, moved, or was unified away) and evaluates it itself otherwise
    int debug;                       // COSLAM_MERGE_DEBUG=1: the walk prints where its time went (diagnostic)
    unsigned char* preOk;            // [nList][nCams]: 0 not evaluated, 1 checkUnify said no, 2 yes
    double* preM;                    // [nList][nCams][12]: the unified position and covariance of a yes
    // MapPoint::pFeatures as references (cs_track_history_set_merge_refs; null: this frame's features alone): checkUnify reads both points'
    // rows composed from pointFeat + the table (dm_compose), an attach brings its entry up at once (a stale feature there becomes a linked
    // segment, :775-779), the hand-over of a unification follows the reference's loop over pFeatures (:806-816)
    int4* featRef;                   // [P][nCams]
    unsigned char* refStatic;        // [P][nCams] or null
    int4* segPoolW;                  // [nCams][segCap]
    int* segCount;                   // [nCams]
};
// the feature of map point x in camera c as the reference holds it NOW: this frame's (pointFeat) with what the table knows of its chain -- the
// table's entry moved on by a frame, or already brought up -- or, without a feature of this frame, the table's stale one
__device__ __forceinline__ int4 dm_compose(const DmArgs& A, int x, int c, int pf) {
    const int4 ref = A.featRef[(size_t)x * A.cu.nCams + c];
    const int cur = A.cu.curFrame;
    if (pf >= 0) {
        if (ref.x == pf && (ref.y == cur || ref.y == cur - 1)) return make_int4(pf, cur, ref.z, ref.w);
        const int f1 = A.cu.cam[c].trackSpan[pf];
        return make_int4(pf, cur, f1 >= 0 ? f1 : cur, -1);
    }
    if (ref.x >= 0 && ref.y < cur) return ref;
    return make_int4(-1, 0, 0, -1);
}
// what cs_feat_ref_advance_dev does to the entry of (point x, camera c) when the walk gives it feature s of this frame
__device__ __forceinline__ void dm_attach_ref(const DmArgs& A, int x, int c, int s) {
    const size_t e = (size_t)x * A.cu.nCams + c;
    int4 ref = A.featRef[e];
    const int cur = A.cu.curFrame;
    if (ref.x >= 0 && ref.y < cur) {   // a stale feature held there: it hangs behind the new one (pFeat->preFrame = p->pFeatures[iCam])
        const int idx = atomicAdd(A.segCount + c, 1);
        if (idx < A.cu.segCap) A.segPoolW[(size_t)c * A.cu.segCap + idx] = ref;
        ref = make_int4(s, cur, cur, idx < A.cu.segCap ? idx : -1);
    } else {
        const int f1 = A.cu.cam[c].trackSpan[s];
        ref = make_int4(s, cur, f1 >= 0 ? f1 : cur, -1);
    }
    A.featRef[e] = ref;
    if (A.refStatic) A.refStatic[e] = A.cu.cam[c].isStatic ? A.cu.cam[c].isStatic[s] : 1;
}
// the conflicts a bMerge walk can meet, judged side by side on the state the walk starts from: a wave per (listed point, camera)
__global__ __launch_bounds__(256) void k_merge_precheck(DmArgs A) {
    __shared__ double sR[4][64 * 9 + 16];
    const int tid = threadIdx.x, g = tid / 64, r = tid % 64, C = A.cu.nCams, N = A.cu.N;
    const int e = blockIdx.x * 4 + g;   // entry = list position x camera
    if (e >= A.nList * C) return;
    const int j = e / C, i = e - j * C;
    if (r == 0) A.preOk[e] = 0;
    const int p = A.list[j];
    if (p < 0 || p >= A.P) return;
    if ((A.mapFlags[p] & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN)) != 0) return;
    if (A.pointFeat[(size_t)p * C + i] >= 0) return;
    const int s = A.slot[(size_t)p * C + i];
    if (s < 0 || s >= N || (A.flags[(size_t)p * C + i] & 2)) return;
    const int q = A.cu.cam[i].slot2map[s] - A.mapBase;
    if (q < 0 || q >= A.P || q == p) return;
    if (A.mapFlags[q] & (CS_MAP_DYNAMIC | CS_MAP_FALSE)) return;
    double M[3], cov[9];
    __shared__ int4 sRef[4][2][DM_MAX_CAMS];
    if (A.featRef) {
        if (r < C) sRef[g][0][r] = dm_compose(A, p, r, A.pointFeat[(size_t)p * C + r]);
        else if (r >= 16 && r < 16 + C) sRef[g][1][r - 16] = dm_compose(A, q, r - 16, A.pointFeat[(size_t)q * C + r - 16]);
        wave_fence_lds();
    }
    const bool ok = check_unify_wave(A.cu, A.pointFeat + (size_t)p * C, A.pointFeat + (size_t)q * C, A.featRef ? sRef[g][0] : nullptr,
                                     A.featRef ? sRef[g][1] : nullptr, A.mapPts + 3 * (size_t)p, A.mapPts + 3 * (size_t)q, sR[g], M, cov);
    if (r == 0) {
        double* o = A.preM + 12 * (size_t)e;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = M[k];
#pragma unroll
        for (int k = 0; k < 9; ++k) o[3 + k] = cov[k];
        A.preOk[e] = ok ? 2 : 1;
    }
}
__device__ __forceinline__ int mg_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned char mg_ldb(const unsigned char* p) { return *(volatile const unsigned char*)p; }
constexpr int DM_DIRTY_WORDS = 2048;   // by list: the "touched" marks of up to 65536 points as a bitmap in LDS
__global__ __launch_bounds__(64) void k_decide_merge(DmArgs A) {
    __shared__ double sR[64 * 9 + 16];
    __shared__ int4 sRefW[2][DM_MAX_CAMS];
    __shared__ unsigned dirtyBits[DM_DIRTY_WORDS];
    for (int w = threadIdx.x; w < DM_DIRTY_WORDS; w += 64) dirtyBits[w] = 0u;
    __syncthreads();
    auto is_dirty = [&](int x) { return (dirtyBits[(x >> 5) & (DM_DIRTY_WORDS - 1)] >> (x & 31)) & 1u; };
    auto set_dirty = [&](int x) { dirtyBits[(x >> 5) & (DM_DIRTY_WORDS - 1)] |= 1u << (x & 31); };   // (lane 0, between two barriers)
    const int lane = threadIdx.x, C = A.cu.nCams, N = A.cu.N, P = A.P;
    int nAtt = 0, nReg = 0, nMerged = 0, nAsked = 0;
    long long tPre = 0, tInline = 0, tUnify = 0, tHead = 0, tWalk = 0, tAll = wall_clock64();
    int nBatch = 0, nVisit = 0, nAct = 0, nInline = 0, nPreUsed = 0;
    const bool byList = A.list != nullptr;
    if (!byList) {
        for (int k = lane; k < P * C; k += 64) A.attached[k] = 0;
        for (int p = lane; p < P; p += 64) A.regged[p] = 0;
    }
    const int o0 = A.onlyCam >= 0 ? A.onlyCam : 0, o1 = A.onlyCam >= 0 ? A.onlyCam + 1 : C;
    const int nRows = byList ? A.nList : P;
    for (int o = o0; o < o1; ++o) {
        if (!byList) {
            for (int p = lane; p < P; p += 64)
                A.inVec[p] = ((A.mapFlags[p] & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN)) == 0 && A.pointFeat[(size_t)p * C + o] >= 0) ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        for (int p0 = 0; p0 < nRows; p0 += 64) {
          // the next 64 points' places on the visiting list as a mask: the wave steps through the set bits only.  By list: the visiting
          // list of camera o (:864-869) is read off as the walk reaches a point -- the same set: only a point's OWN walk gives it features,
          // and a point that lost them to a unification is false by then (tested below)
          int myP = -1;
          bool in = false;
          if (byList) {
              myP = p0 + lane < nRows ? A.list[p0 + lane] : -1;
              in = myP >= 0 && myP < P && (mg_ldb(A.mapFlags + myP) & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN)) == 0 &&
                   mg_ld(A.pointFeat + (size_t)myP * C + o) >= 0;
              if (__builtin_amdgcn_ballot_w64(myP >= 0) == 0ull) break;   // behind the list's end
          } else {
              myP = p0 + lane;
              in = myP < P && A.inVec[myP] != 0;
          }
          // by list: the batch's 64 points read their rows side by side (lane = point: every camera's candidate, its flags, whether the point
          // has a feature there, who owns the candidate) -- the walk below then takes a point's row out of its lane's registers instead of
          // waiting for five dependent loads per visit (3900 visits of ~1.2 us were the kernel's 4.7 ms).  The rows stand until the wave
          // itself changes something (an attach, a unification): from then on the rest of the batch reads memory again.
          int bSlot[DM_MAX_CAMS], bOwner[DM_MAX_CAMS];
          unsigned bHas = 0, bDyn = 0, bMerge = 0, bPre = 0;   // bPre: two bits per camera, the pre-check's verdict (0 none, 1 no, 2 yes)
          bool batchClean = byList;
          const long long tb0 = wall_clock64();
          ++nBatch;
          if (byList) {
              // ONE acquire at agent scope (the wave's own earlier stores went out behind a release fence: the L1 is dropped, what follows
              // comes from L2), then plain loads -- 5 x nCams of them in flight per lane.  (Read one by one as agent-scope atomic loads, each
              // waited for, the batch's rows cost ~40 us: 112 batches of them were most of the kernel's 8 ms.)
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (one wave: workgroup scope orders its own accesses; agent scope wrote the XCD's L2 back every time)
              int hasV[DM_MAX_CAMS];
#pragma unroll
              for (int i = 0; i < DM_MAX_CAMS; ++i) {
                  bSlot[i] = -1, bOwner[i] = -1, hasV[i] = -1;
                  if (in && i < C) {
                      int sl = A.slot[(size_t)myP * C + i];
                      const int fl = A.flags[(size_t)myP * C + i];
                      if (sl >= N) sl = -1;
                      bSlot[i] = sl;
                      if (fl & 2) bDyn |= 1u << i;
                      if (A.mergeable[(size_t)myP * C + i] == 1) bMerge |= 1u << i;
                      hasV[i] = A.pointFeat[(size_t)myP * C + i];
                      if (A.preOk) bPre |= (unsigned)(A.preOk[(size_t)(p0 + lane) * C + i] & 3) << (2 * i);
                  }
              }
#pragma unroll
              for (int i = 0; i < DM_MAX_CAMS; ++i) {
                  if (hasV[i] >= 0) bHas |= 1u << i;
                  if (bSlot[i] >= 0) bOwner[i] = A.cu.cam[i].slot2map[bSlot[i]];
              }
          }
          // ... and every lane works out, for ITS point on the rows as read, whether its walk could change anything: an unmapped candidate it
          // may attach, a conflict the pre-check answered with yes, or a pair with a touched point (whose verdict has to be formed now).
          // A walk that only meets conflicts answered with no changes nothing -- it is counted, not walked -- for as long as the batch
          // stands as read (the wave's first attach / unification sends the rest of the batch through the walk proper).  Of ~4500 visits
          // a pass ~100 remain: each was ~1.5 us of dependent latencies on the one wave, 7 of the kernel's 8 ms.
          int lnAsk = 0;
          bool lnVisit = false;
          if (byList && in && A.preOk) {
              const bool pDirty = is_dirty(myP);
#pragma unroll
              for (int i = 0; i < DM_MAX_CAMS; ++i) {
                  if (i < C && !((bHas >> i) & 1u) && bSlot[i] >= 0 && !((bDyn >> i) & 1u)) {
                      const int own = bOwner[i];
                      if (own < 0) {
                          lnVisit |= ((bMerge >> i) & 1u) != 0;
                      } else {
                          const int q = own - A.mapBase;
                          if (q >= 0 && q < P && q != myP) {
                              const unsigned pre = (bPre >> (2 * i)) & 3u;
                              if (pDirty || is_dirty(q)) lnVisit = true;
                              else if (pre == 2u) lnVisit = true;
                              else if (pre == 1u) ++lnAsk;
                          }
                      }
                  }
              }
          } else {
              lnVisit = true;
          }
          const unsigned long long needMask = __builtin_amdgcn_ballot_w64(lnVisit);
          tPre += wall_clock64() - tb0;
          unsigned long long todo = __builtin_amdgcn_ballot_w64(in);
          while (todo) {
            if (batchClean) {
                // the walks in front of the next one that may change something: counted in one go
                const unsigned long long need = todo & needMask;
                const unsigned long long skip = need ? (todo & ~needMask & ((need & (~need + 1ull)) - 1ull)) : (todo & ~needMask);
                if (skip) {
                    int a = ((skip >> lane) & 1ull) ? lnAsk : 0;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                    nAsked += a, nPreUsed += a, nVisit += __popcll(skip);
                    todo &= ~skip;
                    if (!todo) break;
                }
            }
            ++nVisit;
            const long long tv0 = wall_clock64();
            const int src = __builtin_ctzll(todo);
            const int jP = p0 + src;   // (by list: the point's place on the list)
            todo &= todo - 1;
            if (batchClean && !((needMask >> src) & 1ull)) {   // nothing this walk meets can change anything: its conflicts, all answered no, counted
                const int a = __shfl(lnAsk, src, 64);
                nAsked += a, nPreUsed += a;
                continue;
            }
            const int p = __shfl(myP, src, 64);
            // :734 isLocalStatic(): unified away meanwhile?  (only the wave's own steps unify: an untouched batch stands as it was read)
            if (!batchClean && (*(volatile unsigned char*)(A.mapFlags + p) & (CS_MAP_DYNAMIC | CS_MAP_FALSE))) continue;
            // lane c: camera c's entry of the point -- the candidate, and what it would meet there as things stand (the state changes only
            // through this wave's own steps: an attach leaves the point's later cameras as they were, a unify ends the walk)
            int mySlot = -1, myFlags = 0, myMerge = 0, myHas = 0, myOwner = -1;
            const unsigned hPre = byList ? (unsigned)__shfl((int)bPre, src, 64) : 0u;   // (the pre-check's verdicts do not go stale: only unused)
            if (batchClean) {
                const unsigned hHas = (unsigned)__shfl((int)bHas, src, 64), hDyn = (unsigned)__shfl((int)bDyn, src, 64),
                               hMerge = (unsigned)__shfl((int)bMerge, src, 64);
#pragma unroll
                for (int i = 0; i < DM_MAX_CAMS; ++i) {
                    const int sl = __shfl(bSlot[i], src, 64), ow = __shfl(bOwner[i], src, 64);
                    if (lane == i) mySlot = sl, myOwner = ow;
                }
                if (lane < C) myFlags = ((hDyn >> lane) & 1u) ? 2 : 0, myMerge = (hMerge >> lane) & 1u, myHas = (hHas >> lane) & 1u;
            } else if (lane < C) {
                mySlot = A.slot[(size_t)p * C + lane], myFlags = A.flags[(size_t)p * C + lane], myMerge = A.mergeable[(size_t)p * C + lane];
                myHas = mg_ld(A.pointFeat + (size_t)p * C + lane) >= 0;
                if (mySlot >= N) mySlot = -1;
                if (mySlot >= 0) myOwner = mg_ld(A.cu.cam[lane].slot2map + mySlot);
            }
            // cameras with something to do: no feature of the point, a non-dynamic candidate that is unmapped-and-mergeable or carries a point
            const unsigned long long act = __builtin_amdgcn_ballot_w64(lane < C && !myHas && mySlot >= 0 && !(myFlags & 2) && (myOwner >= 0 || myMerge == 1));
            tHead += wall_clock64() - tv0;
            if (!act) continue;
            ++nAct;
            const long long tw0 = wall_clock64();
            bool reg = false;
            for (int i = 0; i < C; ++i) {
                if (!((act >> i) & 1)) continue;                                            // :736-737, :757: has a feature / nothing found / DYNAMIC
                const int s = __shfl(mySlot, i, 64), mg = __shfl(myMerge, i, 64);
                int* s2m = const_cast<int*>(A.cu.cam[i].slot2map);
                const int m = __shfl(myOwner, i, 64);
                if (m < 0) {
                    if (mg == 1) {                                                          // :760-787
                        if (lane == 0) {
                            s2m[s] = A.mapBase + p;
                            A.pointFeat[(size_t)p * C + i] = s;
                            A.attached[(size_t)p * C + i] = 1;
                            if (A.featRef) dm_attach_ref(A, p, i, s);
                            if (byList) set_dirty(p);   // the point has changed: a pre-checked verdict about it no longer stands
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __syncthreads();
                        batchClean = false;
                        reg = true, ++nAtt;
                    }
                    continue;
                }
                const int q = m - A.mapBase;                                               // :791-796
                if (q < 0 || q >= P || q == p) continue;
                double M[3], cov[9];
                bool ok;
                // By list, while NEITHER point has been touched by this pass the pair stands as the pre-check saw it: a verdict from there (no
                // memory is read: the marks live in LDS, the verdict came with the batch's rows), and "none" means the pre-check's own
                // conditions ruled the pair out -- the other point is not a static one.  (Each conflict used to wait for five dependent loads:
                // 2500 of them were 7.6 of the kernel's 9 ms.)
                const bool clean = byList && A.preOk && !is_dirty(p) && !is_dirty(q);
                const int pre = clean ? (int)((hPre >> (2 * i)) & 3u) : 0;
                if (clean && pre == 0) continue;
                if (!clean && (*(volatile unsigned char*)(A.mapFlags + q) & (CS_MAP_DYNAMIC | CS_MAP_FALSE))) continue;   // !isLocalStatic()
                ++nAsked;
                if (clean) {
                    ++nPreUsed;
                    ok = pre == 2;   // judged before the walk on the very state it is asked about now
                    if (ok) {
                        const double* o = A.preM + 12 * ((size_t)jP * C + i);
#pragma unroll
                        for (int k = 0; k < 3; ++k) M[k] = o[k];
#pragma unroll
                        for (int k = 0; k < 9; ++k) cov[k] = o[3 + k];
                    }
                } else {
                    const long long ti = wall_clock64();
                    if (A.featRef) {
                        if (lane < C) sRefW[0][lane] = dm_compose(A, p, lane, mg_ld(A.pointFeat + (size_t)p * C + lane));
                        else if (lane >= 16 && lane < 16 + C) sRefW[1][lane - 16] = dm_compose(A, q, lane - 16, mg_ld(A.pointFeat + (size_t)q * C + lane - 16));
                        wave_fence_lds();
                    }
                    ok = check_unify_wave(A.cu, A.pointFeat + (size_t)p * C, A.pointFeat + (size_t)q * C, A.featRef ? sRefW[0] : nullptr,
                                          A.featRef ? sRefW[1] : nullptr, A.mapPts + 3 * (size_t)p, A.mapPts + 3 * (size_t)q, sR, M, cov);
                    tInline += wall_clock64() - ti, ++nInline;
                }
                if (!ok) continue;
                const long long tu = wall_clock64();
                if (lane == 0) {                                                            // :797-826
                    for (int k = 0; k < 3; ++k) A.mapPts[3 * (size_t)p + k] = M[k];
                    for (int k = 0; k < 9; ++k) A.mapCov[9 * (size_t)p + k] = cov[k];
                    A.mapFlags[q] = (unsigned char)((A.mapFlags[q] & CS_MAP_UNCERTAIN) | CS_MAP_FALSE);
                    if (byList) set_dirty(p), set_dirty(q);
                    for (int v = 0; v < C; ++v) {
                        const int sq = A.pointFeat[(size_t)q * C + v];
                        if (A.featRef) {
                            // :806-816 over pFeatures as they are: `pFt && !p->pFeatures[v]` -- a stale feature of p blocks the hand-over in its
                            // camera, a stale feature of q moves like a live one (its chain with it)
                            const size_t ep = (size_t)p * C + v, eq = (size_t)q * C + v;
                            const int4 rq = dm_compose(A, q, v, sq), rp = dm_compose(A, p, v, A.pointFeat[ep]);
                            if (rq.x < 0 || rp.x >= 0) continue;
                            A.featRef[ep] = rq, A.featRef[eq] = make_int4(-1, 0, 0, -1);
                            if (A.refStatic) A.refStatic[ep] = A.refStatic[eq];
                            if (sq >= 0) {
                                A.pointFeat[eq] = -1;
                                const_cast<int*>(A.cu.cam[v].slot2map)[sq] = A.mapBase + p;
                                A.pointFeat[ep] = sq;
                                if (v == i && sq == s) break;   // pFeat->mpt is p from here on (:808)
                            }
                            continue;
                        }
                        if (sq >= 0 && A.pointFeat[(size_t)p * C + v] < 0) {
                            A.pointFeat[(size_t)q * C + v] = -1;
                            const_cast<int*>(A.cu.cam[v].slot2map)[sq] = A.mapBase + p;
                            A.pointFeat[(size_t)p * C + v] = sq;
                            if (v == i && sq == s) break;   // pFeat->mpt is p from here on (:808)
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                tUnify += wall_clock64() - tu;
                batchClean = false;
                reg = true, ++nMerged;
                break;                                                                      // :825 return
            }
            if (reg) {
                if (lane == 0) A.regged[p] = 1;
                ++nReg;
            }
            tWalk += wall_clock64() - tw0;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
    }
    if (lane == 0 && A.counts) A.counts[0] = nAtt, A.counts[1] = nReg, A.counts[2] = nMerged, A.counts[3] = nAsked;
    if (lane == 0 && A.debug)
        printf("k_decide_merge: %lld us; %d batches read in %lld us; %d visits, %d with work; %d conflicts asked: %d from the pre-check, %d evaluated here in "
               "%lld us; %d unified in %lld us; %d attached; visit heads %lld us, walks %lld us\n",
               (wall_clock64() - tAll) / 100, nBatch, tPre / 100, nVisit, nAct, nAsked, nPreUsed, nInline, tInline / 100, nMerged, tUnify / 100, nAtt,
               tHead / 100, tWalk / 100);
}

// ---- CoSLAM::mapPointsClassify (src/app/SL_CoSLAM.cpp:418-520) ------------------------------------------------------------------------
// Every frame behind the pose update (CoSLAM::poseUpdate, :381-385) the reference re-examines every map point of the current list that
// is uncertain (what the gate above made of it, and every new point until it is 30 frames old) or locally dynamic: static again
// (isStaticPoint over the last 60 frames: the views of updateStaticPointPosition inside that window, triangulation, covariance, every
// view within Mahalanobis distance 1), dynamic (isDynamicPoint: this frame's features, in front of the first camera, a covariance
// small against the distance, every view within the gate), static without its worst view (isStaticRemovable -> that feature is
// detached), or false; dynamic points that stand still (isLittleMove) for more than 50 frames may return to static
// (src/slam/SL_CoSLAMHelper.cpp:67-330).
//
// Two launches.  k_classify_select: a lane per map point puts the points to examine on a worklist (tens to a few hundred of the
// map's thousands; new points sit next to each other at the map's end, so a wave that worked through its own 64 points would
// serialise exactly the busy stretch).  k_map_points_classify: A WAVE PER EXAMINED POINT.  Lane c keeps camera c's feature of
// the point; the widest-parallax walks (up to 60 frames back per camera) are split over the lanes and reduced (smallest cosine, the
// nearest frame among equals -- what the reference's backwards walk with `<` keeps); then lane v owns view v: its row pair of the
// triangulation's normal equations, its Jacobian for the covariance, its gate -- the f64 divisions, which are most of the
// arithmetic, run once per view side by side, and the sums over views are taken in the reference's view order from the lanes'
// registers (bit-identical to the sequential sum: -ffp-contract=off).  Every branch of the decision is uniform over the wave.
// Nothing is shared between points (a feature belongs to one point), so the order of the worklist does not matter.
// Helper definitions as above, plus isAtCameraBack(R, t, M) = (R M + t).z < 0 and dist3 = Euclidean distance.
struct ClsArgs {
    int nCams, N, nMap, H, head, nHist, curFrame;
    int* pointFeat;        // [nMap][nCams] in / out (a detached feature becomes -1)
    const int* featFrame;  // [nMap][nCams] or null: the frame of MapPoint::pFeatures[iCam] (null: all of this frame)
    const int* featFirst;  // [nMap][nCams] or null: the first frame of that feature's track (null: the slot's trackSpan)
    int4* featRef;         // [nMap][nCams] or null: the features as references (cs_feat_ref as the END of the previous frame left the table,
                           // or already at this frame); then featFrame / featFirst are not read.  In / out: a detached view's is cleared
    unsigned char* refStatic;  // [nMap][nCams] or null: the stale features' types (a point that returns to static sets them)
    const int4* segPool;   // [nCams][segCap]
    int segCap, stored;    // stored: frames the ring holds
    const double* histXY;
    const double* histR;
    const double* histT;
    const double* cen;
    double* mapPts;
    double* mapCov;
    unsigned char* mapFlags;
    unsigned char* newPt;
    int* staticFrameNum;
    const int* firstFrame;
    double sigma;
    int* counts;  // [2] points examined / points that became false, or null
    int* list;    // [2 + nMap] worklist: two counters (this call's, the next call's: zeroed here for it), then the points
    int par;      // which counter is this call's
    cs_poseupdate_cam cam[PU_MAX_CAMS];
};
// camera c's feature of point m as the lane c of the point's wave keeps it: slot (< 0: none, or older than the history), walk
// depth of its frame, that frame, the first frame of its track
struct ClsFeat {
    int s, j0, f, ff, seg;   // seg: the first linked segment behind the feature's own run (-1 none)
};
// With references (A.featRef): pointFeat names this frame's features -- the hand-back has moved a live pointer along its track
// (SL_SingleSLAM.cpp:34-60) -- and the table is MapPoint::pFeatures as the END of the previous frame left it: a feature of this frame is
// the reference moved on by one frame (a table already at this frame is taken as it is); no feature of this frame and an older reference:
// the camera lost the point, the stale feature stands (a view of isStaticPoint inside its window, of isLittleMove and isStaticRemovable).
__device__ __forceinline__ ClsFeat cls_feature(const ClsArgs& A, int m, int c) {
    ClsFeat F;
    F.s = -1, F.j0 = 0, F.f = 0, F.ff = 0, F.seg = -1;
    if (c >= A.nCams) return F;
    const int s = A.pointFeat[(size_t)m * A.nCams + c];
    if (A.featRef) {
        const int4 ref = A.featRef[(size_t)m * A.nCams + c];
        int sl = s;
        if (s >= 0) {
            F.f = A.curFrame;
            if (ref.x == s && (ref.y == A.curFrame || ref.y == A.curFrame - 1)) F.ff = ref.z, F.seg = ref.w;
            else F.ff = A.cam[c].trackSpan[s];
        } else if (ref.x >= 0 && ref.y < A.curFrame)
            sl = ref.x, F.f = ref.y, F.ff = ref.z, F.seg = ref.w;
        else
            return F;
        F.j0 = A.curFrame - F.f;
        if (F.j0 < 0 || F.j0 >= A.stored) {   // older than the ring: treated as absent
            F.seg = -1;
            return F;
        }
        F.s = sl;
        return F;
    }
    if (s < 0) return F;
    F.f = A.featFrame ? A.featFrame[(size_t)m * A.nCams + c] : A.curFrame;
    F.j0 = A.curFrame - F.f;
    if (F.j0 < 0 || F.j0 >= A.nHist) return F;  // older than the history: treated as absent
    F.ff = A.featFirst ? A.featFirst[(size_t)m * A.nCams + c] : A.cam[c].trackSpan[s];
    F.s = s;
    return F;
}
__device__ __forceinline__ const double* cls_R(const ClsArgs& A, int c, int j) {
    return A.histR + ((size_t)c * A.H + (A.head - j + A.H) % A.H) * 9;
}
__device__ __forceinline__ const double* cls_t(const ClsArgs& A, int c, int j) {
    return A.histT + ((size_t)c * A.H + (A.head - j + A.H) % A.H) * 3;
}
__device__ __forceinline__ void cls_pixel(const ClsArgs& A, int c, int j, int s, double& mx, double& my) {
    const double* h = A.histXY + ((size_t)c * A.H + (A.head - j + A.H) % A.H) * 2 * A.N;
    mx = h[s], my = h[A.N + s];
}
// mahaDist2 of a pixel from a projection of (M, cov): getProjectionCovMat, mat22Inv
__device__ __forceinline__ double cls_maha(const PuProj& q, const double* cov, double s2, double mx, double my) {
    const double rm0 = q.u / q.w, rm1 = q.v / q.w;
    double JC[6], var[4], ivar[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) JC[3 * i + k] = (q.J[3 * i] * cov[k] + q.J[3 * i + 1] * cov[3 + k]) + q.J[3 * i + 2] * cov[6 + k];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double sv = (JC[3 * i] * q.J[3 * k] + JC[3 * i + 1] * q.J[3 * k + 1]) + JC[3 * i + 2] * q.J[3 * k + 2];
            var[2 * i + k] = (i == k) ? sv + s2 : sv;
        }
    pu_mat22_inv(var, ivar);
    const double dx = rm0 - mx, dy = rm1 - my;
    return dx * (ivar[0] * dx + ivar[1] * dy) + dy * (ivar[2] * dx + ivar[3] * dy);
}
// the views of a wave's point: lane v < nv holds view v (camera, walk depth, slot)
struct ClsViews {
    int nv, c, j, s;
};
// triangulateMultiView + getTriangulateCovMat over the views (+ the gate of every view when asked: any view > 1 fails); M and cov
// come back in every lane
__device__ __forceinline__ bool cls_solve(const ClsArgs& A, const ClsViews& V, int r, double* M, double* cov, bool gate,
                                          bool* atBackOfFirst = nullptr) {
    UpNormalEq T;
#pragma unroll
    for (int q = 0; q < 6; ++q) T.N[q] = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) T.g[q] = 0;
    double mx = 0, my = 0;
    const double *Rv = A.histR, *tv = A.histT;
    if (r < V.nv) {
        cls_pixel(A, V.c, V.j, V.s, mx, my);
        Rv = cls_R(A, V.c, V.j), tv = cls_t(A, V.c, V.j);
        up_add_view(T, A.cam[V.c].iK, Rv, tv, mx, my);
    }
    UpNormalEq E;
#pragma unroll
    for (int q = 0; q < 6; ++q) E.N[q] = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) E.g[q] = 0;
    for (int v = 0; v < V.nv; ++v) {
#pragma unroll
        for (int q = 0; q < 6; ++q) E.N[q] = E.N[q] + __shfl(T.N[q], v, 64);
#pragma unroll
        for (int q = 0; q < 3; ++q) E.g[q] = E.g[q] + __shfl(T.g[q], v, 64);
    }
    double cf[6];
    const double det = up_sym33_cof(E.N, cf);
    M[0] = ((cf[0] * E.g[0] + cf[1] * E.g[1]) + cf[2] * E.g[2]) / det;
    M[1] = ((cf[1] * E.g[0] + cf[3] * E.g[1]) + cf[4] * E.g[2]) / det;
    M[2] = ((cf[2] * E.g[0] + cf[4] * E.g[1]) + cf[5] * E.g[2]) / det;
    if (atBackOfFirst) {  // isAtCameraBack under the first view's pose
        const bool back = r == 0 && ((Rv[6] * M[0] + Rv[7] * M[1]) + Rv[8] * M[2]) + tv[2] < 0;
        *atBackOfFirst = (__builtin_amdgcn_ballot_w64(back) & 1ull) != 0;
    }
    PuProj q;
    q.u = q.v = 0, q.w = 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) q.J[k] = 0;
    if (r < V.nv) q = pu_project(A.cam[V.c].K, Rv, tv, M);
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int v = 0; v < V.nv; ++v) {
        double Jv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) Jv[k] = __shfl(q.J[k], v, 64);
        up_add_jtj(S, Jv);
    }
    const double dS = up_sym33_cof(S, cf), s2 = A.sigma * A.sigma;
    cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
    cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
    cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
    if (!gate) return true;
    const bool fail = r < V.nv && cls_maha(q, cov, s2, mx, my) > 1.0;
    return __builtin_amdgcn_ballot_w64(fail) == 0;
}
// isStaticPoint (exclude < 0) / isStaticPointExclude (src/slam/SL_CoSLAMHelper.cpp:117-250)
__device__ __noinline__ bool cls_is_static(const ClsArgs& A, const ClsFeat& F, int r, const double* Mold, double* M, double* cov, int exclude,
                                           int numFrame) {
    ClsViews V;
    V.nv = 0, V.c = 0, V.j = 0, V.s = 0;
    const int firstFrame = A.curFrame - numFrame;
    ChainCtx X;
    X.N = A.N, X.H = A.H, X.head = A.head, X.cap = A.nHist, X.curFrame = A.curFrame, X.segCap = A.segCap, X.nCen = A.nHist, X.cen = A.cen;
    X.stored = A.stored < A.nHist ? A.stored : A.nHist;   // (the walks stay inside the centre table: nHist frames)
    X.segPool = A.featRef ? A.segPool : nullptr, X.minFrame = firstFrame;
    // the cameras' walks SIDE BY SIDE: a group of LPC = 64 / nCams lanes (a power of two, at most 8) per camera walks that camera's chain
    // (one camera after the other on the whole wave was eight dependent rounds in front of the solve); the views are then handed to the
    // lanes in the reference's order -- camera by camera, the feature and then its widest-parallax predecessor
    int LPC = 8;
    while (LPC * A.nCams > 64) LPC >>= 1;
    const int grp = r / LPC, sub = r - grp * LPC;
    const int c = grp < A.nCams ? grp : 0;
    const int s = __shfl(F.s, c, 64), f = __shfl(F.f, c, 64), j0 = __shfl(F.j0, c, 64), ff = __shfl(F.ff, c, 64), seg = __shfl(F.seg, c, 64);
    const bool valid = grp < A.nCams && !(c == exclude || s < 0 || f < firstFrame || j0 >= A.nHist);
    int best = -1, bestSlot = -1;
    {
        const double* C0 = A.cen + 3 * ((size_t)c * A.nHist + (valid ? j0 : 0));
        const double a[3] = {C0[0] - Mold[0], C0[1] - Mold[1], C0[2] - Mold[2]};
        const double na = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
        // fp = fp->preFrame while fp && fp->f >= firstFrame (:141-152): the feature's own run of frames f-1 .. ff, then the linked segments,
        // ending at the first node before the window (or the ring); the smallest cosine, the nearest node among equals
        best = chain_widest_w(X, c, valid ? make_int4(s, f, ff, seg) : make_int4(-1, 0, 1, -1), A.histR + (size_t)c * A.H * 9,
                              A.histT + (size_t)c * A.H * 3, a, na, Mold, sub, bestSlot, LPC);
        if (!valid) best = -1;
    }
    for (int cc = 0; cc < A.nCams; ++cc) {
        const int lead = cc * LPC;
        if (!__shfl((int)valid, lead, 64)) continue;   // (uniform)
        const int j0c = __shfl(j0, lead, 64), sc = __shfl(s, lead, 64), b = __shfl(best, lead, 64), bsl = __shfl(bestSlot, lead, 64);
        if (r == V.nv) V.c = cc, V.j = j0c, V.s = sc;
        ++V.nv;
        if (b >= 0) {
            if (r == V.nv) V.c = cc, V.j = b, V.s = bsl;
            ++V.nv;
        }
    }
    return cls_solve(A, V, r, M, cov, true);
}
// isDynamicPoint (:251-312)
__device__ __noinline__ bool cls_is_dynamic(const ClsArgs& A, const ClsFeat& F, int r, double* M, double* cov) {
    ClsViews V;
    V.nv = 0, V.c = 0, V.j = 0, V.s = 0;
    for (int c = 0; c < A.nCams; ++c) {
        const int s = __shfl(F.s, c, 64), f = __shfl(F.f, c, 64);
        if (s < 0 || f != A.curFrame) continue;
        if (r == V.nv) V.c = c, V.s = s;
        ++V.nv;
    }
    if (V.nv < 2) return false;
    const double* org = A.cen + 3 * ((size_t)__shfl(V.c, 0, 64) * A.nHist);
    bool atBack;
    cls_solve(A, V, r, M, cov, false, &atBack);
    if (atBack) return false;
    const double sc = (fabs(cov[0]) + fabs(cov[4])) + fabs(cov[8]);
    const double dx = M[0] - org[0], dy = M[1] - org[1], dz = M[2] - org[2];
    if (sqrt((dx * dx + dy * dy) + dz * dz) * 0.2 < sqrt(sc)) return false;  // :290-293
    bool fail = false;
    if (r < V.nv) {
        double mx, my;
        cls_pixel(A, V.c, 0, V.s, mx, my);
        fail = cls_maha(pu_project(A.cam[V.c].K, cls_R(A, V.c, 0), cls_t(A, V.c, 0), M), cov, A.sigma * A.sigma, mx, my) > 1.0;
    }
    return __builtin_amdgcn_ballot_w64(fail) == 0;
}
// lane c: the Mahalanobis distance of camera c's feature from the projection of (M, cov) under its own frame's pose (-1: no feature)
__device__ __forceinline__ double cls_err_of_lane(const ClsArgs& A, const ClsFeat& F, int r, const double* M, const double* cov) {
    if (F.s < 0) return -1.0;
    double mx, my;
    cls_pixel(A, r, F.j0, F.s, mx, my);
    return cls_maha(pu_project(A.cam[r].K, cls_R(A, r, F.j0), cls_t(A, r, F.j0), M), cov, A.sigma * A.sigma, mx, my);
}

__global__ __launch_bounds__(256) void k_classify_select(ClsArgs A) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m == 0 && A.counts) A.counts[1] = 0;   // (the worker kernel adds to it)
    if (m >= A.nMap) return;
    const unsigned char fl = A.mapFlags[m];
    if (!((fl & CS_MAP_UNCERTAIN) || (fl & (CS_MAP_DYNAMIC | CS_MAP_FALSE)) == CS_MAP_DYNAMIC)) return;  // :431
    // the current list after mapStateUpdate (:1183-1197): points with a feature in this frame
    bool vis = false;
    for (int c = 0; c < A.nCams && !vis; ++c) {
        const int s = A.pointFeat[(size_t)m * A.nCams + c];
        vis = s >= 0 && (A.featFrame ? A.featFrame[(size_t)m * A.nCams + c] : A.curFrame) == A.curFrame;
    }
    if (!vis) return;
    A.list[2 + atomicAdd(A.list + A.par, 1)] = m;
}

constexpr int CLS_WAVES = 1024;   // the worker grid: 256 workgroups of 4 waves, a wave per listed point (and round again past that)
__global__ __launch_bounds__(256) void k_map_points_classify(ClsArgs A) {
    constexpr int FRAME_NUM_FOR_NEWPOINT = 30, FRAME_NUM_FOR_DONTMOVE = 50, NUM_FRAME_CHECK_STATIC = 60;
    CS_POSE_STREAM_PRIO();
    const int r = threadIdx.x % 64;
    const int n = A.list[A.par];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (A.counts) A.counts[0] = n;
        A.list[A.par ^ 1] = 0;   // the next call's counter (nobody reads it before that call's select kernel)
    }
    for (int e = blockIdx.x * 4 + threadIdx.x / 64; e < n; e += CLS_WAVES) {
        const int m = A.list[2 + e];
        const ClsFeat F = cls_feature(A, m, r);
        const int numVisCam = __popcll(__builtin_amdgcn_ballot_w64(F.s >= 0 && F.f == A.curFrame));
        const unsigned char fl0 = A.mapFlags[m];
        unsigned char fl = fl0;
        const bool uncertain = (fl & CS_MAP_UNCERTAIN) != 0;
        double* pM = A.mapPts + 3 * (size_t)m;
        double* pCov = A.mapCov + 9 * (size_t)m;
        double Mold[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) Mold[q] = pM[q];
        double M[3], cov[9];
        bool write = false;   // updatePosition(M, cov)
        bool clearNew = false;
        int sfn = A.staticFrameNum[m];
        if (numVisCam == 1) {  // :433-437
            fl = (unsigned char)((fl & ~CS_MAP_DYNAMIC) | CS_MAP_FALSE);
        } else if (uncertain) {
            if (A.newPt[m]) {
                if (cls_is_static(A, F, r, Mold, M, cov, -1, NUM_FRAME_CHECK_STATIC)) {
                    if (A.curFrame - A.firstFrame[m] > FRAME_NUM_FOR_NEWPOINT) {
                        fl = 0, sfn = 0;
                        clearNew = true;
                        write = true;
                    }
                } else if (cls_is_dynamic(A, F, r, M, cov)) {
                    fl = CS_MAP_DYNAMIC, sfn = 0;
                    write = true;
                    clearNew = true;
                } else
                    fl = (unsigned char)((fl & ~CS_MAP_DYNAMIC) | CS_MAP_FALSE);
            } else {
                if (cls_is_dynamic(A, F, r, M, cov)) {
                    fl = CS_MAP_DYNAMIC, sfn = 0;
                    write = true;
                } else {
                    // isStaticRemovable (:67-115): the view with the largest error (> 1) under the point as it stands; static without it?
                    double covOld[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) covOld[q] = pCov[q];
                    double err = cls_err_of_lane(A, F, r, Mold, covOld);
                    if (!(err == err)) err = -1.0;   // (`err > maxErr` is false for a NaN: never the worst view)
                    const int nVis = __popcll(__builtin_amdgcn_ballot_w64(F.s >= 0));
                    int maxI = r;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {   // the first camera with the largest error
                        const double oe = __shfl_xor(err, off, 64);
                        const int oi = __shfl_xor(maxI, off, 64);
                        if (oe > err || (oe == err && oi < maxI)) err = oe, maxI = oi;
                    }
                    if (!(err > 1.0)) maxI = -1;
                    if (maxI >= 0 && nVis > 2 && cls_is_static(A, F, r, Mold, M, cov, maxI, NUM_FRAME_CHECK_STATIC)) {  // :476-482
                        if (r == maxI) {   // (with references the view that goes may be a stale one: then only the reference is cleared)
                            if (F.f == A.curFrame) {
                                if (A.cam[maxI].slot2map) const_cast<int*>(A.cam[maxI].slot2map)[F.s] = -1;
                                A.pointFeat[(size_t)m * A.nCams + maxI] = -1;
                            }
                            if (A.featRef) A.featRef[(size_t)m * A.nCams + maxI] = make_int4(-1, 0, 0, -1);   // p->pFeatures[outlierViewId] = 0
                        }
                        fl = 0, sfn = 0;
                        write = true;
                    } else
                        fl = (unsigned char)((fl & ~CS_MAP_DYNAMIC) | CS_MAP_FALSE);
                }
            }
        } else {  // locally dynamic (:489-516)
            if (cls_is_dynamic(A, F, r, M, cov)) {
                const double err = cls_err_of_lane(A, F, r, M, cov);
                const bool little = __builtin_amdgcn_ballot_w64(err >= 1) == 0;  // isLittleMove: >= 1 fails
                if (little) {
                    ++sfn;
                    if (sfn > FRAME_NUM_FOR_DONTMOVE) {
                        double M0[3], cov0[9];
                        if (cls_is_static(A, F, r, Mold, M0, cov0, -1, NUM_FRAME_CHECK_STATIC)) {
                            fl = 0, sfn = 0;
                            if (F.s >= 0 && F.f == A.curFrame) A.cam[r].isStatic[F.s] = 1;
                            else if (F.s >= 0 && A.refStatic) A.refStatic[(size_t)m * A.nCams + r] = 1;   // (:494-498: every feature held)
                        } else
                            sfn = 0;
                    }
                } else
                    sfn = 0;
                write = true;  // :512: the dynamic triangulation is what stays, also for a point that went back to static
            } else
                fl = (unsigned char)((fl & ~CS_MAP_DYNAMIC) | CS_MAP_FALSE);
        }
        if (r == 0) {
            if (write) {
#pragma unroll
                for (int q = 0; q < 3; ++q) pM[q] = M[q];
#pragma unroll
                for (int q = 0; q < 9; ++q) pCov[q] = cov[q];
            }
            if (clearNew) A.newPt[m] = 0;
            A.staticFrameNum[m] = sfn;
            A.mapFlags[m] = fl;
            if (A.counts && (fl & CS_MAP_FALSE) && !(fl0 & CS_MAP_FALSE)) atomicAdd(A.counts + 1, 1);
        }
    }
}

// poses of (camera, frame) pairs into the ring: what RobustBundleRTS::output() writes through the CamPoseItem pointers the features
// share (src/app/SL_CoSLAMRobustBA.cpp:283-285 key poses, :239-244 relaxed non-key poses)
__global__ __launch_bounds__(256) void k_history_set_poses(int n, const int* __restrict__ cam, const int* __restrict__ frame,
                                                           const double* __restrict__ R, const double* __restrict__ t, double* hR,
                                                           double* hT, int nCams, int H, int head, int count, int lastFrame) {
    const int q = blockIdx.x * 256 + threadIdx.x, i = q / 12, e = q - 12 * i;
    if (i >= n) return;
    const int c = cam[i], back = lastFrame - frame[i];
    if (c < 0 || c >= nCams || back < 0 || back >= count) return;  // a frame the ring does not hold
    const int rs = (head - back + H) % H;
    if (e < 9)
        hR[((size_t)c * H + rs) * 9 + e] = R[9 * (size_t)i + e];
    else
        hT[((size_t)c * H + rs) * 3 + (e - 9)] = t[3 * (size_t)i + (e - 9)];
}

// a run of consecutive frames [firstFrame, firstFrame + nFrames) of every camera out of / into the ring, camera-major
// ([nCams][nFrames][9] / [3]): the nodes of the camera graphs RobustBundleRTS::constructCameraGraphs walks
// (src/app/SL_CoSLAMRobustBA.cpp:182-227: every CamPoseItem from the window's first key frame to the newest frame) and where
// updateNonKeyCameraPoses puts the relaxed poses back (:230-247).  set != 0: the array into the ring.
__global__ __launch_bounds__(256) void k_history_span(int set, int nCams, int firstFrame, int nFrames, double* R, double* t, double* hR,
                                                      double* hT, int H, int head, int lastFrame) {
    const int q = blockIdx.x * 256 + threadIdx.x, i = q / 12, e = q - 12 * i;
    if (i >= nCams * nFrames) return;
    const int c = i / nFrames, f = firstFrame + (i - c * nFrames), rs = (head - (lastFrame - f) + 2 * H) % H;
    double* a = e < 9 ? R + 9 * (size_t)i + e : t + 3 * (size_t)i + (e - 9);
    double* b = e < 9 ? hR + ((size_t)c * H + rs) * 9 + e : hT + ((size_t)c * H + rs) * 3 + (e - 9);
    if (set)
        *b = *a;
    else
        *a = *b;
}

}  // namespace

struct cs_track_history {
    int device, nCams, N, H;  // H: ring capacity in frames (cs_track_history_create_ex: storeLen >= histLen)
    int walkLen;              // histLen: how deep the bounded walks go (dynamic test, classification, re-triangulation, checkUnify, the
                              // mergability walk's exact window); the frames beyond it are only read by cs_register_mergability_running_dev
    int head, count, lastFrame;
    double *xy, *R, *t;
    // scratch of the map-point kernels (one stream at a time may run them on a handle): the camera centres by walk depth
    // [nCams][H][3], and the worklist of cs_map_points_classify_dev [1 + clsCap] (grown when a larger map is passed)
    double* cen;
    mutable int* clsList;
    mutable int clsCap, clsPar;
    // the centres are computed once per state of the ring's poses: every call that (re)writes poses moves ringVersion on, the
    // kernels that walk the centres launch k_ring_centres only when cenVersion is behind (calls on a handle are enqueued in order)
    mutable long long ringVersion, cenVersion;
    // the segments the registration loops linked behind features (ChainCtx): a pool per camera, filled by cs_feat_ref_advance_dev
    int4* segPool;   // [nCams][segCap]
    int* segCount;   // [nCams]
    int segCap;
    unsigned char* alive;   // cs_feat_ref_advance_dev's scratch [aliveCap]
    int aliveCap;
    // cs_track_history_set_classify_refs: the classification reads (and clears) the points' features as references
    int4* clsFeatRef;
    unsigned char* clsRefStatic;
    // cs_track_history_set_merge_refs: the bMerge walks read and write them
    int4* mergeFeatRef;
    unsigned char* mergeRefStatic;
    // poses rewritten in frames older than the walk depth (the running mergability verdict's cached tails: MgRunArgs::epoch)
    mutable int tailEpoch = 0, tailFromMin = 0x7fffffff;
};

// the camera centres by walk depth, if the ring's poses changed since they were last computed
static void hist_centres(const cs_track_history* h, hipStream_t s);

constexpr int PU_SEG_CAP = 1 << 15;  // linked segments a camera's pool holds (16 bytes each; never recycled: a full pool drops further links)
static inline int hist_walk(const cs_track_history* h) { return h->count < h->walkLen ? h->count : h->walkLen; }

extern "C" cs_track_history* cs_track_history_create(int device, int nCams, int N, int histLen) {
    return cs_track_history_create_ex(device, nCams, N, histLen, histLen);
}

// storeLen >= histLen frames are KEPT (pixels of every slot + poses: 16 N + 96 bytes per camera and frame -- 4096 frames of 8 cameras x
// 2000 slots are 1 GB of the 288); the walks of every kernel but the running mergability verdict stay histLen deep
extern "C" cs_track_history* cs_track_history_create_ex(int device, int nCams, int N, int histLen, int storeLen) {
    if (nCams < 1 || nCams > PU_MAX_CAMS || N < 1 || histLen < 1 || histLen > PU_MAX_HIST || storeLen < histLen || storeLen > PU_MAX_STORE) {
        cs_set_error("cs_track_history_create: nCams in 1..%d, N >= 1, histLen in 1..%d, histLen <= storeLen <= %d", PU_MAX_CAMS, PU_MAX_HIST,
                     PU_MAX_STORE);
        return nullptr;
    }
    const int walkLen_ = histLen;
    histLen = storeLen;
    if (hipSetDevice(device) != hipSuccess) {
        cs_set_error("cs_track_history_create: no usable HIP device %d (there is no CPU fallback)", device);
        return nullptr;
    }
    cs_track_history* h = new cs_track_history();
    h->device = device, h->nCams = nCams, h->N = N, h->H = histLen;
    h->walkLen = walkLen_;
    h->head = -1, h->count = 0, h->lastFrame = -0x7fffffff;
    h->clsList = nullptr, h->clsCap = 0, h->clsPar = 0;
    h->ringVersion = 1, h->cenVersion = 0;
    const size_t nXY = (size_t)nCams * histLen * 2 * N, nR = (size_t)nCams * histLen * 9, nT = (size_t)nCams * histLen * 3;
    if (hipMalloc((void**)&h->xy, sizeof(double) * nXY) != hipSuccess || hipMalloc((void**)&h->R, sizeof(double) * nR) != hipSuccess ||
        hipMalloc((void**)&h->t, sizeof(double) * nT) != hipSuccess || hipMalloc((void**)&h->cen, sizeof(double) * nT) != hipSuccess ||
        hipMalloc((void**)&h->segPool, sizeof(int4) * (size_t)nCams * PU_SEG_CAP) != hipSuccess ||
        hipMalloc((void**)&h->segCount, sizeof(int) * nCams) != hipSuccess) {
        cs_set_error("cs_track_history_create: hipMalloc failed");
        (void)hipFree(h->xy), (void)hipFree(h->R), (void)hipFree(h->t), (void)hipFree(h->cen), (void)hipFree(h->segPool), (void)hipFree(h->segCount);
        delete h;
        return nullptr;
    }
    h->segCap = PU_SEG_CAP;
    (void)hipMemset(h->segCount, 0, sizeof(int) * nCams);
    (void)hipMemset(h->xy, 0, sizeof(double) * nXY);
    (void)hipMemset(h->R, 0, sizeof(double) * nR);
    (void)hipMemset(h->t, 0, sizeof(double) * nT);
    return h;
}

extern "C" void cs_track_history_destroy(cs_track_history* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->xy), (void)hipFree(h->R), (void)hipFree(h->t), (void)hipFree(h->cen), (void)hipFree(h->segPool), (void)hipFree(h->segCount);
    if (h->clsList) (void)hipFree(h->clsList);
    if (h->alive) (void)hipFree(h->alive);
    delete h;
}

extern "C" int cs_track_history_frames(const cs_track_history* h) { return h ? h->count : 0; }

static void hist_centres(const cs_track_history* h, hipStream_t s) {
    if (h->cenVersion == h->ringVersion) return;
    hipLaunchKernelGGL(k_ring_centres, dim3((h->nCams * hist_walk(h) + 255) / 256), dim3(256), 0, s, h->nCams, h->H, h->head, hist_walk(h), h->R,
                       h->t, h->cen);
    h->cenVersion = h->ringVersion;
}

namespace {

int pu_launch(const char* who, int device, void* hip_stream, PuArgs& A, const cs_poseupdate_cam* cams, bool gate, bool dyn) {
    for (int c = 0; c < A.nCams; ++c) {
        const cs_poseupdate_cam& q = cams[c];
        const bool run = c >= A.cam0 && c < A.cam0 + A.nCamsRun;
        if (run && (!q.K || !q.xy || !q.state || !q.slot2map || (gate && !q.reprojErr) || (dyn && (!q.iK || !q.trackSpan || !q.isStatic)))) {
            cs_set_error("%s: null pointer in camera %d", who, c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = q;
    }
    CS_HIP(hipSetDevice(device));
    A.gateBlocks = gate ? (A.nMap + 255) / 256 : 0;
    A.dynBlocksPerCam = (A.N * PU_LPS + 255) / 256;
    const int dynBlocks = dyn ? A.dynBlocksPerCam * A.nCamsRun : 0;
    if (A.gateBlocks + dynBlocks == 0) return CS_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    if (gate && A.numNodes) CS_HIP(hipMemsetAsync(A.numNodes + A.cam0, 0, sizeof(int) * A.nCamsRun, s));
    if (gate && A.numOut) CS_HIP(hipMemsetAsync(A.numOut + A.cam0, 0, sizeof(int) * A.nCamsRun, s));
    if (dyn && A.numDyn) CS_HIP(hipMemsetAsync(A.numDyn + A.cam0, 0, sizeof(int) * A.nCamsRun, s));
    const size_t lds = dyn ? sizeof(double) * 9 * (size_t)(A.nHist > 0 ? A.nHist : 1) : 0;
    A.cenBlocks = A.cenOut ? (A.nCams * A.nHist + 255) / 256 : 0;
    hipLaunchKernelGGL(k_pose_update, dim3(A.gateBlocks + dynBlocks + A.cenBlocks), dim3(256), lds, s, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int pu_check_common(const char* who, int nCams, int cam0, int nCamsRun, const cs_poseupdate_cam* cams, int N, const double* d_R,
                    const double* d_t) {
    if (nCams < 1 || nCams > PU_MAX_CAMS || cam0 < 0 || nCamsRun < 0 || cam0 + nCamsRun > nCams || !cams || N < 1 || !d_R || !d_t) {
        cs_set_error("%s: nCams in 1..%d, 0 <= cam0, cam0 + nCamsRun <= nCams, N >= 1, non-null cams / d_R / d_t", who, PU_MAX_CAMS);
        return CS_ERR_INVALID;
    }
    return CS_OK;
}

void pu_fill_gate(PuArgs& A, const int* d_pointFeat, int nMap, double* d_mapPts, double* d_mapCov, unsigned char* d_mapFlags,
                  int largeErr, double pixelErrVar, int* d_numNodes, int* d_numOut) {
    A.pointFeat = d_pointFeat;
    A.nMap = nMap;
    A.mapPts = d_mapPts, A.mapCov = d_mapCov, A.mapFlags = d_mapFlags;
    A.errThres = largeErr ? 6.0 : 2.0;  // :673
    A.sigma = pixelErrVar;
    A.numNodes = d_numNodes, A.numOut = d_numOut;
}

// advance the ring to `frame` (host bookkeeping only; the launch writes the entry)
void pu_advance(cs_track_history* h, int frame) {
    h->ringVersion += 1;
    if (frame == h->lastFrame) return;  // the same frame again (camera-by-camera calls): the entry is rewritten
    if (frame != h->lastFrame + 1) h->count = 0;  // Track2D's length() counts frames: a gap in the numbering loses the history
    h->head = (h->head + 1) % h->H;
    h->count = h->count < h->H ? h->count + 1 : h->H;
    h->lastFrame = frame;
}

void pu_fill_dyn(PuArgs& A, cs_track_history* h, int minLen, int minOutNum, double maxEpiErr, int* d_numDyn) {
    A.H = h->H, A.head = h->head, A.nHist = hist_walk(h);
    A.histXY = h->xy, A.histR = h->R, A.histT = h->t;
    A.minLen = minLen, A.minOutNum = minOutNum, A.maxEpiErr = maxEpiErr;
    A.numDyn = d_numDyn;
}

}  // namespace

extern "C" int cs_pose_update3d_dev(int device, void* hip_stream, int nCams, int cam0, int nCamsRun, const cs_poseupdate_cam* cams,
                                    int N, const int* d_pointFeat, int nMap, const double* d_R, const double* d_t, double* d_mapPts,
                                    double* d_mapCov, unsigned char* d_mapFlags, int largeErr, double pixelErrVar, int* d_numNodes,
                                    int* d_numOut) {
    int rc = pu_check_common("cs_pose_update3d_dev", nCams, cam0, nCamsRun, cams, N, d_R, d_t);
    if (rc != CS_OK) return rc;
    if (nMap < 0 || (nMap > 0 && (!d_pointFeat || !d_mapPts || !d_mapCov || !d_mapFlags))) {
        cs_set_error("cs_pose_update3d_dev: null map pointer");
        return CS_ERR_INVALID;
    }
    PuArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = nCams, A.N = N, A.cam0 = cam0, A.nCamsRun = nCamsRun, A.R = d_R, A.t = d_t;
    pu_fill_gate(A, d_pointFeat, nMap, d_mapPts, d_mapCov, d_mapFlags, largeErr, pixelErrVar, d_numNodes, d_numOut);
    return pu_launch("cs_pose_update3d_dev", device, hip_stream, A, cams, true, false);
}

extern "C" int cs_detect_dynamic_dev(cs_track_history* h, void* hip_stream, int cam0, int nCamsRun, const cs_poseupdate_cam* cams,
                                     const double* d_R, const double* d_t, int nMap, const unsigned char* d_mapFlags, int frame,
                                     int maxLen, int minLen, int minOutNum, double maxEpiErr, int* d_numDyn) {
    if (!h) {
        cs_set_error("cs_detect_dynamic_dev: null history");
        return CS_ERR_INVALID;
    }
    (void)maxLen;  // SL_SingleSLAM.cpp:799: the reference never advances the counter it compares with maxLen
    int rc = pu_check_common("cs_detect_dynamic_dev", h->nCams, cam0, nCamsRun, cams, h->N, d_R, d_t);
    if (rc != CS_OK) return rc;
    if (nMap > 0 && !d_mapFlags) {
        cs_set_error("cs_detect_dynamic_dev: null map flags");
        return CS_ERR_INVALID;
    }
    PuArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = h->nCams, A.N = h->N, A.cam0 = cam0, A.nCamsRun = nCamsRun, A.R = d_R, A.t = d_t;
    A.nMap = nMap, A.mapFlags = (unsigned char*)d_mapFlags;
    pu_advance(h, frame);
    pu_fill_dyn(A, h, minLen, minOutNum, maxEpiErr, d_numDyn);
    return pu_launch("cs_detect_dynamic_dev", h->device, hip_stream, A, cams, false, true);
}

extern "C" int cs_pose_update_frame_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const int* d_pointFeat,
                                        int nMap, const double* d_R, const double* d_t, double* d_mapPts, double* d_mapCov,
                                        unsigned char* d_mapFlags, int largeErr, double pixelErrVar, int frame, int maxLen, int minLen,
                                        int minOutNum, double maxEpiErr, int* d_numNodes, int* d_numOut, int* d_numDyn) {
    if (!h) {
        cs_set_error("cs_pose_update_frame_dev: null history");
        return CS_ERR_INVALID;
    }
    (void)maxLen;
    int rc = pu_check_common("cs_pose_update_frame_dev", h->nCams, 0, h->nCams, cams, h->N, d_R, d_t);
    if (rc != CS_OK) return rc;
    if (nMap < 0 || (nMap > 0 && (!d_pointFeat || !d_mapPts || !d_mapCov || !d_mapFlags))) {
        cs_set_error("cs_pose_update_frame_dev: null map pointer");
        return CS_ERR_INVALID;
    }
    PuArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = h->nCams, A.N = h->N, A.cam0 = 0, A.nCamsRun = h->nCams, A.R = d_R, A.t = d_t;
    pu_fill_gate(A, d_pointFeat, nMap, d_mapPts, d_mapCov, d_mapFlags, largeErr, pixelErrVar, d_numNodes, d_numOut);
    pu_advance(h, frame);
    pu_fill_dyn(A, h, minLen, minOutNum, maxEpiErr, d_numDyn);
    return pu_launch("cs_pose_update_frame_dev", h->device, hip_stream, A, cams, true, true);
}

extern "C" int cs_register_mergability_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int P,
                                           const double* d_M, const double* d_cov, const int* d_slot, double pixelErrVar,
                                           unsigned char* d_mergeable) {
    return cs_register_mergability_range_dev(h, hip_stream, 0, h ? h->nCams : 0, cams, P, d_M, d_cov, d_slot, pixelErrVar, d_mergeable);
}

extern "C" int cs_register_mergability_range_dev(const cs_track_history* h, void* hip_stream, int cam0, int nCamsRun,
                                                 const cs_poseupdate_cam* cams, int P, const double* d_M, const double* d_cov,
                                                 const int* d_slot, double pixelErrVar, unsigned char* d_mergeable) {
    if (h && (cam0 < 0 || nCamsRun < 0 || cam0 + nCamsRun > h->nCams)) {
        cs_set_error("cs_register_mergability_range_dev: camera range %d + %d of %d", cam0, nCamsRun, h->nCams);
        return CS_ERR_INVALID;
    }
    if (!h || !cams || P < 0 || (P > 0 && (!d_M || !d_cov || !d_slot || !d_mergeable))) {
        cs_set_error("cs_register_mergability_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (h->count < 1) {
        cs_set_error("cs_register_mergability_dev: the history holds no frame (cs_pose_update_frame_dev / cs_detect_dynamic_dev push one per frame)");
        return CS_ERR_INVALID;
    }
    if (P == 0 || nCamsRun == 0) return CS_OK;
    MgArgs A;
    memset(&A, 0, sizeof(A));
    A.cam0 = cam0;
    A.nCams = h->nCams, A.N = h->N, A.P = P, A.H = h->H, A.head = h->head, A.nHist = hist_walk(h);
    A.sigma = pixelErrVar;
    A.M = d_M, A.cov = d_cov, A.slot = d_slot, A.out = d_mergeable;
    A.histXY = h->xy, A.histR = h->R, A.histT = h->t;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].K || !cams[c].trackSpan) {
            cs_set_error("cs_register_mergability_dev: null pointer in camera %d", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    CS_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_register_mergability, dim3((P * MG_LPC + 255) / 256, nCamsRun), dim3(256), sizeof(double) * 12 * (size_t)hist_walk(h),
                       (hipStream_t)hip_stream, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

extern "C" size_t cs_register_mergability_cache_bytes(int P, int nCams) { return P > 0 && nCams > 0 ? sizeof(MgCache) * (size_t)P * nCams : 0; }

// staticCheckMergability over whole tracks with the tail's verdict cached per (map point, camera) -- see k_register_mergability_running.
// d_cache: cs_register_mergability_cache_bytes(P, nCams) bytes, ZERO-filled before the first call, kept between the frames (point p of
// one call must be point p of the next); tolPix: how far a point may move in the camera's image before its cached tail is judged
// again (0: any motion); d_counts [4] or NULL (added to): cache hits, full tail walks, verdicts 2, tail terms evaluated.
extern "C" int cs_register_mergability_running_dev(const cs_track_history* h, void* hip_stream, int cam0, int nCamsRun,
                                                   const cs_poseupdate_cam* cams, int P, const double* d_M, const double* d_cov,
                                                   const int* d_slot, double pixelErrVar, double tolPix, void* d_cache,
                                                   unsigned char* d_mergeable, int* d_counts) {
    return cs_register_mergability_running_list_dev(h, hip_stream, cam0, nCamsRun, cams, P, nullptr, 0, d_M, d_cov, d_slot, nullptr, pixelErrVar, tolPix,
                                                    d_cache, d_mergeable, d_counts);
}
extern "C" int cs_register_mergability_running_list_dev(const cs_track_history* h, void* hip_stream, int cam0, int nCamsRun,
                                                        const cs_poseupdate_cam* cams, int P, const int* d_list, int nList, const double* d_M,
                                                        const double* d_cov, const int* d_slot, const int* d_flags, double pixelErrVar,
                                                        double tolPix, void* d_cache, unsigned char* d_mergeable, int* d_counts) {
    if (h && (cam0 < 0 || nCamsRun < 0 || cam0 + nCamsRun > h->nCams)) {
        cs_set_error("cs_register_mergability_running_dev: camera range %d + %d of %d", cam0, nCamsRun, h->nCams);
        return CS_ERR_INVALID;
    }
    if (!h || !cams || P < 0 || tolPix < 0 || (P > 0 && (!d_M || !d_cov || !d_slot || !d_mergeable || !d_cache))) {
        cs_set_error("cs_register_mergability_running_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (h->count < 1) {
        cs_set_error("cs_register_mergability_running_dev: the history holds no frame");
        return CS_ERR_INVALID;
    }
    if (d_list && nList < 0) {
        cs_set_error("cs_register_mergability_running_list_dev: nList < 0");
        return CS_ERR_INVALID;
    }
    const int rows = d_list ? nList : P;
    if (P == 0 || nCamsRun == 0 || rows == 0) return CS_OK;
    MgRunArgs B;
    memset(&B, 0, sizeof(B));
    B.list = d_list, B.nList = nList, B.flags = d_flags;
    MgArgs& A = B.a;
    A.cam0 = cam0;
    A.nCams = h->nCams, A.N = h->N, A.P = P, A.H = h->H, A.head = h->head, A.nHist = hist_walk(h);
    A.sigma = pixelErrVar;
    A.M = d_M, A.cov = d_cov, A.slot = d_slot, A.out = d_mergeable;
    A.histXY = h->xy, A.histR = h->R, A.histT = h->t;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].K || !cams[c].trackSpan) {
            cs_set_error("cs_register_mergability_running_dev: null pointer in camera %d", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    B.W = hist_walk(h), B.count = h->count, B.curFrame = h->lastFrame;
    B.epoch = h->tailEpoch, B.fromMin = h->tailFromMin;
    B.tolPix = tolPix;
    B.cache = (MgCache*)d_cache;
    B.counts = d_counts;
    CS_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_register_mergability_running, dim3((rows * MG_LPC + 255) / 256, nCamsRun), dim3(256), sizeof(double) * 12 * (size_t)B.W,
                       (hipStream_t)hip_stream, B);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

extern "C" int cs_track_history_set_poses_dev(cs_track_history* h, void* hip_stream, int n, const int* d_cam, const int* d_frame,
                                              const double* d_R, const double* d_t) {
    if (!h || n < 0 || (n > 0 && (!d_cam || !d_frame || !d_R || !d_t))) {
        cs_set_error("cs_track_history_set_poses_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (n == 0 || h->count < 1) return CS_OK;
    h->ringVersion += 1;
    h->tailEpoch += 1, h->tailFromMin = -0x7fffffff;   // (which frames: only the device knows -- every cached tail is walked again)
    CS_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_history_set_poses, dim3((n * 12 + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, n, d_cam, d_frame, d_R, d_t,
                       h->R, h->t, h->nCams, h->H, h->head, h->count, h->lastFrame);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

namespace {
int hist_span(const char* who, cs_track_history* h, void* hip_stream, int set, int firstFrame, int nFrames, double* d_R, double* d_t) {
    if (!h || nFrames < 0 || (nFrames > 0 && (!d_R || !d_t))) {
        cs_set_error("%s: bad arguments", who);
        return CS_ERR_INVALID;
    }
    if (nFrames == 0) return CS_OK;
    if (firstFrame + nFrames - 1 > h->lastFrame || h->lastFrame - firstFrame >= h->count) {
        cs_set_error("%s: frames %d..%d are not all in the ring (it holds %d frame(s), the newest %d)", who, firstFrame, firstFrame + nFrames - 1,
                     h->count, h->lastFrame);
        return CS_ERR_INVALID;
    }
    if (set) {
        h->ringVersion += 1;
        if (firstFrame <= h->lastFrame - h->walkLen) {   // tail frames rewritten: the cached verdicts that cover them are stale
            h->tailEpoch += 1;
            if (firstFrame < h->tailFromMin) h->tailFromMin = firstFrame;
        }
    }
    CS_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_history_span, dim3((h->nCams * nFrames * 12 + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, set, h->nCams,
                       firstFrame, nFrames, d_R, d_t, h->R, h->t, h->H, h->head, h->lastFrame);
    CS_HIP(hipGetLastError());
    return CS_OK;
}
}  // namespace

extern "C" int cs_track_history_get_span_dev(const cs_track_history* h, void* hip_stream, int firstFrame, int nFrames, double* d_R, double* d_t) {
    return hist_span("cs_track_history_get_span_dev", (cs_track_history*)h, hip_stream, 0, firstFrame, nFrames, d_R, d_t);
}
extern "C" int cs_track_history_set_span_dev(cs_track_history* h, void* hip_stream, int firstFrame, int nFrames, const double* d_R,
                                             const double* d_t) {
    return hist_span("cs_track_history_set_span_dev", h, hip_stream, 1, firstFrame, nFrames, (double*)d_R, (double*)d_t);
}
extern "C" int cs_track_history_newest_frame(const cs_track_history* h) { return h ? h->lastFrame : -0x7fffffff; }
extern "C" int cs_track_history_cams(const cs_track_history* h) { return h ? h->nCams : 0; }

namespace {
int up_launch(const char* who, const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, UpArgs& A, int* d_counts, int nCounts) {
    if (h->count < 1) {
        cs_set_error("%s: the history holds no frame (cs_pose_update_frame_dev / cs_detect_dynamic_dev push one per frame)", who);
        return CS_ERR_INVALID;
    }
    A.nCams = h->nCams, A.N = h->N, A.H = h->H, A.head = h->head, A.nHist = hist_walk(h);
    A.histXY = h->xy, A.histR = h->R, A.histT = h->t;
    A.counts = d_counts;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].K || !cams[c].iK || (!A.featRef && !cams[c].trackSpan) || (!A.refine && !A.refStatic && !cams[c].isStatic)) {
            cs_set_error("%s: null pointer in camera %d (K, iK, trackSpan%s are read)", who, c, A.refine ? "" : ", isStatic");
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    CS_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (d_counts) {
        cs_small::List ops;
        ops.fill(d_counts, 0, nCounts * sizeof(int));
        CS_HIP(ops.run(s));
    }
    if (A.nMap == 0) return CS_OK;
    A.cen = h->cen;
    hist_centres(h, s);
    hipLaunchKernelGGL(k_update_points, dim3((A.nMap * UP_LPP + 255) / 256), dim3(256), 0, s, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}
}  // namespace

extern "C" int cs_update_new_poses_points_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams,
                                              const int* d_pointFeat, int nMap, const int* d_lastFrame,
                                              const unsigned char* d_isCurrent, int firstKeyFrame, double* d_mapPts, double* d_mapCov,
                                              const unsigned char* d_mapFlags, double pixelErrVar, int* d_counts) {
    if (!h || !cams || nMap < 0 || (nMap > 0 && (!d_pointFeat || !d_mapPts || !d_mapCov || !d_mapFlags))) {
        cs_set_error("cs_update_new_poses_points_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    UpArgs A;
    memset(&A, 0, sizeof(A));
    A.nMap = nMap, A.firstKeyFrame = firstKeyFrame;
    A.pointFeat = d_pointFeat, A.lastFrame = d_lastFrame, A.isCurrent = d_isCurrent;
    A.mapPts = d_mapPts, A.mapCov = d_mapCov, A.mapFlags = d_mapFlags;
    A.sigma = pixelErrVar;
    return up_launch("cs_update_new_poses_points_dev", h, hip_stream, cams, A, d_counts, 2);
}

extern "C" int cs_refine_map_points_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const int* d_pointFeat,
                                        int nMap, const unsigned char* d_select, double* d_mapPts, double* d_mapCov, double pixelErrVar,
                                        int* d_count) {
    if (!h || !cams || nMap < 0 || (nMap > 0 && (!d_pointFeat || !d_mapPts || !d_mapCov))) {
        cs_set_error("cs_refine_map_points_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    UpArgs A;
    memset(&A, 0, sizeof(A));
    A.nMap = nMap;
    A.pointFeat = d_pointFeat;
    A.mapPts = d_mapPts, A.mapCov = d_mapCov;
    A.sigma = pixelErrVar;
    A.refine = 1, A.select = d_select;
    return up_launch("cs_refine_map_points_dev", h, hip_stream, cams, A, d_count, 1);
}

// ---- SingleSLAM::newMapPoints (src/app/SL_SingleSLAM.cpp:922-1004): the intra-camera source of new map points -----------------------------
// What CoSLAM::genNewMapPoints calls for a camera that IsReadyForKeyFrame (SL_CoSLAM.cpp:1310-1330; cs_keyframe_ready_dev's codes): every
// unmapped feature on a track of at least minTrackLen frames (getUnMappedAndTrackedFeatPts, :152-172) is triangulated from its OWN track --
// the track's oldest feature the history still holds against the current one --, thrown out when the point lies behind the camera, nearer
// than the square root of its covariance's trace (:960-962) or re-projects further off than maxEpiErr in either view; refineTriangulation
// (:1005-1049) then pairs the current view with the widest-parallax one behind it (chain_widest: at most maxWalk nodes back) and the tests
// run once more.  The reference walks every candidate on the host; here a wave per (camera, slot) tries its slot and leaves the result in a
// scratch record, and ONE workgroup then appends the successes to the map in (camera, slot) order -- the order the reference's loop over
// the cameras and its loop over the tracks would create them in -- so that map indices do not depend on scheduling.
struct InArgs {
    int nCams, N, H, head, nHist, stored, curFrame, minTrackLen, maxWalk, readyMin, mapCap;
    const double *histXY, *histR, *histT, *cen;
    const int* ready;          // [nCams] or null (= every camera)
    double maxEpiErr, sigma;
    double* rec;               // scratch [nCams * N][12]: M, cov (symmetric: 9 stored)
    int* first;                // scratch [nCams * N]
    unsigned char* ok;         // scratch [nCams * N]
    double *mapPts, *mapCov;
    unsigned char *mapFlags, *newPt;
    int *firstFrame, *pointFeat, *mapCount, *counts;
    cs_poseupdate_cam cam[PU_MAX_CAMS];
};
__device__ __forceinline__ void in_tri2(const double* iK, const double* R1, const double* t1, double x1, double y1, const double* R2, const double* t2,
                                        double x2, double y2, double* M) {
    UpNormalEq E;
#pragma unroll
    for (int q = 0; q < 6; ++q) E.N[q] = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) E.g[q] = 0;
    up_add_view(E, iK, R1, t1, x1, y1);
    up_add_view(E, iK, R2, t2, x2, y2);
    double cf[6];
    const double det = up_sym33_cof(E.N, cf);
    M[0] = ((cf[0] * E.g[0] + cf[1] * E.g[1]) + cf[2] * E.g[2]) / det;
    M[1] = ((cf[1] * E.g[0] + cf[3] * E.g[1]) + cf[4] * E.g[2]) / det;
    M[2] = ((cf[2] * E.g[0] + cf[4] * E.g[1]) + cf[5] * E.g[2]) / det;
}
__device__ __forceinline__ void in_cov2(const double* K, const double* R1, const double* t1, const double* R2, const double* t2, const double* M,
                                        double sigma, double* cov) {
    double S[6] = {0, 0, 0, 0, 0, 0}, cf[6];
    const PuProj q1 = pu_project(K, R1, t1, M), q2 = pu_project(K, R2, t2, M);
    up_add_jtj(S, q1.J);
    up_add_jtj(S, q2.J);
    const double dS = up_sym33_cof(S, cf), s2 = sigma * sigma;
    cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
    cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
    cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
}
__device__ __forceinline__ double in_reproj_err(const double* K, const double* R, const double* t, const double* M, double mx, double my) {
    const PuProj q = pu_project(K, R, t, M);
    const double dx = mx - q.u / q.w, dy = my - q.v / q.w;
    return sqrt(dx * dx + dy * dy);
}
__device__ __forceinline__ bool in_behind(const double* R, const double* t, const double* M) {
    return ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2] < 0;
}
__global__ __launch_bounds__(256) void k_intracam_newpts_try(InArgs A) {
    const int tid = threadIdx.x, g = tid / 64, r = tid % 64;
    const int e = blockIdx.x * 4 + g;   // (camera, slot)
    if (e >= A.nCams * A.N) return;
    const int c = e / A.N, k = e - c * A.N, N = A.N, H = A.H;
    if (r == 0) A.ok[e] = 0;
    if (A.ready && A.ready[c] < A.readyMin) return;
    const cs_poseupdate_cam& C = A.cam[c];
    const int st = C.state[k];
    if (st != 0 && st != 1) return;
    const int f1 = C.trackSpan[k], f2 = C.trackSpan[N + k];
    if (f1 < 0 || f2 - f1 < A.minTrackLen || C.slot2map[k] >= 0 || !C.isStatic[k]) return;   // :159, :938-944
    int jp = f2 - f1;
    if (jp > A.stored - 1) jp = A.stored - 1;   // the oldest feature that still has a pose (:937)
    if (jp < 1) return;
    if (r == 0 && A.counts) atomicAdd(A.counts, 1);
    const double* hR = A.histR + (size_t)c * H * 9;
    const double* hT = A.histT + (size_t)c * H * 3;
    const double* hXY = A.histXY + (size_t)c * H * 2 * N;
    const int rsP = ((A.head - jp) % H + H) % H;
    const double *R0 = hR + (size_t)A.head * 9, *t0 = hT + (size_t)A.head * 3, *Rp = hR + (size_t)rsP * 9, *tp = hT + (size_t)rsP * 3;
    const double cx = hXY[(size_t)A.head * 2 * N + k], cy = hXY[(size_t)A.head * 2 * N + N + k];
    const double px = hXY[(size_t)rsP * 2 * N + k], py = hXY[(size_t)rsP * 2 * N + N + k];
    double M[3], cov[9];
    in_tri2(C.iK, Rp, tp, px, py, R0, t0, cx, cy, M);                                   // :950
    if (in_behind(R0, t0, M)) return;                                                    // :953
    in_cov2(C.K, Rp, tp, R0, t0, M, A.sigma, cov);                                       // :957
    double org[3];
    up_cam_center(R0, t0, org);
    {
        const double sTr = fabs((cov[0] + cov[4]) + cov[8]);
        const double dx = org[0] - M[0], dy = org[1] - M[1], dz = org[2] - M[2];
        if (sqrt((dx * dx + dy * dy) + dz * dz) < sqrt(sTr)) return;                     // :960-962
    }
    if (!(in_reproj_err(C.K, Rp, tp, M, px, py) < A.maxEpiErr && in_reproj_err(C.K, R0, t0, M, cx, cy) < A.maxEpiErr)) return;   // :965-970
    // refineTriangulation(cur_fp, M, cov): the current view and the widest-parallax one of the track behind it
    ChainCtx X;
    X.N = N, X.H = H, X.head = A.head, X.cap = (jp + 1 < A.maxWalk ? jp + 1 : A.maxWalk), X.curFrame = A.curFrame, X.stored = A.stored;
    X.segCap = 0, X.nCen = A.nHist, X.cen = A.cen, X.segPool = nullptr;
    X.minFrame = -2147483647 - 1;
    const double a[3] = {org[0] - M[0], org[1] - M[1], org[2] - M[2]};
    const double na = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
    int bs = k;
    const int best = chain_widest(X, c, make_int4(k, A.curFrame, A.curFrame - jp, -1), hR, hT, a, na, M, r, bs);
    if (best >= 0) {
        const int rs = ((A.head - best) % H + H) % H;
        const double *Rb = hR + (size_t)rs * 9, *tb = hT + (size_t)rs * 3;
        in_tri2(C.iK, R0, t0, cx, cy, Rb, tb, hXY[(size_t)rs * 2 * N + k], hXY[(size_t)rs * 2 * N + N + k], M);
        in_cov2(C.K, R0, t0, Rb, tb, M, A.sigma, cov);
    }
    const double e1 = in_reproj_err(C.K, Rp, tp, M, px, py), e2 = in_reproj_err(C.K, R0, t0, M, cx, cy);
    if (in_behind(R0, t0, M) || in_behind(Rp, tp, M)) return;                            // :977-979
    if (!(e1 < A.maxEpiErr && e2 < A.maxEpiErr)) return;                                 // :980
    if (r != 0) return;
    double* o = A.rec + 12 * (size_t)e;
#pragma unroll
    for (int q = 0; q < 3; ++q) o[q] = M[q];
#pragma unroll
    for (int q = 0; q < 9; ++q) o[3 + q] = cov[q];
    A.first[e] = f2 - jp;
    A.ok[e] = 1;
}
// the successes into the map in (camera, slot) order: MapPoint(M, pre_fp->f), cov, addFeature(camId, cur_fp), setLocalStatic() (:981-990)
__global__ __launch_bounds__(1024) void k_intracam_newpts_commit(InArgs A) {
    __shared__ int sWave[16], sBase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, total = A.nCams * A.N;
    if (tid == 0) sBase = *A.mapCount;
    __syncthreads();
    int added = 0;
    for (int e0 = 0; e0 < total; e0 += 1024) {
        const int e = e0 + tid;
        const bool ok = e < total && A.ok[e] != 0;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(ok);
        if (lane == 0) sWave[wv] = __popcll(b);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) before += w < wv ? sWave[w] : 0, all += sWave[w];
        const int idx = sBase + added + before + __popcll(b & ((1ull << lane) - 1ull));
        if (ok && idx < A.mapCap) {
            const int c = e / A.N, k = e - c * A.N;
            const double* o = A.rec + 12 * (size_t)e;
#pragma unroll
            for (int q = 0; q < 3; ++q) A.mapPts[3 * (size_t)idx + q] = o[q];
#pragma unroll
            for (int q = 0; q < 9; ++q) A.mapCov[9 * (size_t)idx + q] = o[3 + q];
            A.mapFlags[idx] = 0;   // setLocalStatic(), bUncertain = false
            A.newPt[idx] = 1;      // bNewPt = true (the constructor's)
            A.firstFrame[idx] = A.first[e];
            for (int v = 0; v < A.nCams; ++v) A.pointFeat[(size_t)idx * A.nCams + v] = v == c ? k : -1;
            const_cast<int*>(A.cam[c].slot2map)[k] = idx;
        }
        added += all;
        __syncthreads();
    }
    if (tid == 0) {
        int kept = added;
        if (sBase + kept > A.mapCap) kept = A.mapCap - sBase > 0 ? A.mapCap - sBase : 0;
        *A.mapCount = sBase + kept;
        if (A.counts) A.counts[1] = kept, A.counts[2] = added - kept;
    }
}

extern "C" size_t cs_newpts_intracam_scratch_bytes(int nCams, int N) {
    if (nCams < 1 || N < 1) return 0;
    const size_t n = (size_t)nCams * N;
    return n * 12 * sizeof(double) + n * sizeof(int) + ((n + 15) & ~(size_t)15);
}
extern "C" int cs_newpts_intracam_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const int* d_ready, int readyMin,
                                      int minTrackLen, int maxWalk, double maxEpiErr, double pixelErrVar, double* d_mapPts, double* d_mapCov,
                                      unsigned char* d_mapFlags, unsigned char* d_newPt, int* d_firstFrame, int* d_pointFeat, int mapCap,
                                      int* d_mapCount, void* d_scratch, int* d_counts) {
    if (!h || !cams || minTrackLen < 1 || maxWalk < 2 || !d_mapPts || !d_mapCov || !d_mapFlags || !d_newPt || !d_firstFrame || !d_pointFeat ||
        mapCap < 1 || !d_mapCount || !d_scratch) {
        cs_set_error("cs_newpts_intracam_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (h->count < 1) {
        cs_set_error("cs_newpts_intracam_dev: the history holds no frame");
        return CS_ERR_INVALID;
    }
    InArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = h->nCams, A.N = h->N, A.H = h->H, A.head = h->head, A.nHist = hist_walk(h), A.stored = h->count < h->H ? h->count : h->H;
    A.curFrame = h->lastFrame, A.minTrackLen = minTrackLen, A.maxWalk = maxWalk, A.readyMin = readyMin, A.mapCap = mapCap;
    A.histXY = h->xy, A.histR = h->R, A.histT = h->t, A.cen = h->cen;
    A.ready = d_ready, A.maxEpiErr = maxEpiErr, A.sigma = pixelErrVar;
    const size_t n = (size_t)h->nCams * h->N;
    A.rec = (double*)d_scratch, A.first = (int*)((char*)d_scratch + n * 12 * sizeof(double)), A.ok = (unsigned char*)(A.first + n);
    A.mapPts = d_mapPts, A.mapCov = d_mapCov, A.mapFlags = d_mapFlags, A.newPt = d_newPt, A.firstFrame = d_firstFrame, A.pointFeat = d_pointFeat;
    A.mapCount = d_mapCount, A.counts = d_counts;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].K || !cams[c].iK || !cams[c].state || !cams[c].slot2map || !cams[c].trackSpan || !cams[c].isStatic) {
            cs_set_error("cs_newpts_intracam_dev: null pointer in camera %d (K, iK, state, slot2map, trackSpan, isStatic)", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    CS_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (d_counts) CS_HIP(hipMemsetAsync(d_counts, 0, 3 * sizeof(int), s));
    hist_centres(h, s);
    hipLaunchKernelGGL(k_intracam_newpts_try, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_intracam_newpts_commit, dim3(1), dim3(1024), 0, s, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ---- MapPoint::pFeatures kept as references (cs_feat_ref) ---------------------------------------------------------------------------------
// What the reference does to p->pFeatures[c] and the chain behind it, once per frame behind the registration's decisions:
//   the camera tracks the point on (SingleSLAM::propagateFeatureStates, src/app/SL_SingleSLAM.cpp:34-60): the reference moves to this frame;
//   the point gained a feature on a NEW track while it still held an older one there (curStaticPointRegInGroup, src/app/SL_CoSLAM.cpp:775-779;
//     the dynamic loop :997-1000): the old reference becomes a segment of the camera's pool, the new one {slot, this frame, first = this
//     frame, that segment} -- the new track's own earlier frames are cut off, exactly as `pFeat->preFrame = p->pFeatures[iCam]` cuts them;
//   the point holds no feature there yet (MapPoint::addFeature on a null entry: new map points, a unification's hand-over): {slot, this
//     frame, the track's first frame, no segment};
//   the camera does not see the point: the reference stays as it is -- stale, and still a view of every walk.
//   ... except when the reference was alive in the frame before and its slot's track lives ON without the point: the feature was detached
//     (`p->pFeatures[outlierViewId] = 0` of mapPointsClassify, :470-472; `pFeat->mpt->pFeatures[v] = 0` of a unification, :810): cleared.
// The call has to be made EVERY frame (a detachment is recognised against the frame before).
struct FrCore {
    int nCams, N, nMap, curFrame, segCap;
    const int* pointFeat;
    int4* featRef;
    unsigned char* refStatic;
    int4* segPool;
    int* segCount;
    int* counts;   // [5] or null: tracked on, fresh, re-linked, links dropped (pool full), detached
    unsigned char* alive;   // [nMap]: one of the point's references was of the frame of the last call (the history's scratch)
    const int* list;        // null, or the rows to look at: list[0 .. nList), entries < 0 skipped (a SECOND call within a frame, behind a
    int nList;              // registration round that changed just these points' features: every other row stands as the first call left it)
};
struct FrArgs : FrCore {
    cs_poseupdate_cam cam[PU_MAX_CAMS];
};
// one row (a map point's nCams references) of cs_feat_ref_advance_dev; cnt: tracked on, first, re-linked, links dropped, detached
__device__ __forceinline__ void fr_advance_row(const FrCore& A, const cs_poseupdate_cam* cam, int m, int (&cnt)[5]) {
    if (m >= 0 && m < A.nMap) {
        const int* pf = A.pointFeat + (size_t)m * A.nCams;
        int top = -1;   // (no short circuit: the row's loads go out together, not one after the other)
#pragma unroll 8
        for (int c = 0; c < A.nCams; ++c) top = max(top, pf[c]);
        const bool any = top >= 0;
        const bool was = A.alive[m] != 0;
        bool aliveNow = false;
        if (any || was) {
            // four cameras at a time: their references, then the track spans those name, are requested together (a point's eight cameras one
            // after the other were sixteen dependent memory round trips under the tracker's load: most of this kernel's 15 us)
            for (int c0 = 0; c0 < A.nCams; c0 += 4) {
                int4 refs[4];
                int sl[4], g1[4], g2[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = c0 + q < A.nCams ? c0 + q : A.nCams - 1;
                    refs[q] = A.featRef[(size_t)m * A.nCams + c];
                    sl[q] = pf[c];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = c0 + q < A.nCams ? c0 + q : A.nCams - 1;
                    const int* span = cam[c].trackSpan;
                    const bool on = sl[q] >= 0 && sl[q] < A.N;
                    const int look = on ? sl[q] : ((refs[q].x >= 0 && refs[q].x < A.N) ? refs[q].x : 0);
                    g1[q] = span[look], g2[q] = span[A.N + look];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = c0 + q;
                    if (c >= A.nCams) break;
                    const size_t e = (size_t)m * A.nCams + c;
                    const int s = sl[q];
                    int4 ref = refs[q];
                    const cs_poseupdate_cam& C = cam[c];
                    if (s < 0 || s >= A.N) {
                        if (ref.x >= 0 && ref.x < A.N && ref.y == A.curFrame - 1 && g1[q] >= 0 && g1[q] <= ref.y && g2[q] == A.curFrame) {
                            ref.x = -1, ++cnt[4];   // the same track, alive in this frame, no longer the point's: detached
                            A.featRef[e] = ref;
                        }
                        aliveNow = aliveNow || (ref.x >= 0 && ref.y == A.curFrame);
                        continue;
                    }
                    const int f1 = g1[q];
                    if (ref.x == s && f1 >= 0 && ref.y >= f1 && ref.y <= A.curFrame) {
                        if (ref.y != A.curFrame) ++cnt[0];   // (a second call within the frame changes and counts nothing)
                        ref.y = A.curFrame;
                    } else if (ref.x >= 0 && ref.y < A.curFrame) {
                        const int idx = atomicAdd(A.segCount + c, 1);
                        ++cnt[2];
                        if (idx < A.segCap)
                            A.segPool[(size_t)c * A.segCap + idx] = ref;   // {slot, last = its frame, first, next = its segment}
                        else
                            ++cnt[3];
                        ref = make_int4(s, A.curFrame, A.curFrame, idx < A.segCap ? idx : -1);
                    } else {
                        ref = make_int4(s, A.curFrame, f1 >= 0 ? f1 : A.curFrame, -1), ++cnt[1];
                    }
                    A.featRef[e] = ref;
                    if (A.refStatic) A.refStatic[e] = C.isStatic ? C.isStatic[s] : 1;
                    aliveNow = true;
                }
            }
            A.alive[m] = aliveNow ? 1 : 0;
        }
    }
}
// a thread per MAP POINT (its nCams entries): most of a map's points are seen by no camera in a frame and were not the frame before --
// their row of pointFeat (and one byte saying whether any of their references was alive last frame) is all that is read
__global__ __launch_bounds__(256) void k_feat_ref_advance(FrArgs A) {
    const int idx0 = blockIdx.x * 256 + threadIdx.x;
    const int m = A.list ? (idx0 < A.nList ? A.list[idx0] : -1) : idx0;
    int cnt[5] = {0, 0, 0, 0, 0};   // tracked on, first, re-linked, links dropped, detached
    fr_advance_row(A, A.cam, m, cnt);
    if (A.counts) {   // one atomic per WORKGROUP and counter: atomics on one address serialise (360 waves' worth were most of this kernel's time)
        __shared__ int sCnt[4][5];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int v = cnt[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) sCnt[wv][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            const int v = (sCnt[0][threadIdx.x] + sCnt[1][threadIdx.x]) + (sCnt[2][threadIdx.x] + sCnt[3][threadIdx.x]);
            if (v) atomicAdd(A.counts + threadIdx.x, v);
        }
    }
}

extern "C" int cs_feat_ref_advance_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nMap, const int* d_pointFeat,
                                       int curFrame, cs_feat_ref* d_featRef, unsigned char* d_refStatic, int* d_counts) {
    return cs_feat_ref_advance_list_dev(h, hip_stream, cams, nMap, d_pointFeat, curFrame, d_featRef, d_refStatic, d_counts, nullptr, 0);
}

extern "C" int cs_feat_ref_advance_list_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nMap, const int* d_pointFeat,
                                            int curFrame, cs_feat_ref* d_featRef, unsigned char* d_refStatic, int* d_counts, const int* d_list,
                                            int nList) {
    if (!h || !cams || nMap < 0 || nList < 0 || (nMap > 0 && (!d_pointFeat || !d_featRef))) {
        cs_set_error("cs_feat_ref_advance_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    FrArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = h->nCams, A.N = h->N, A.nMap = nMap, A.curFrame = curFrame, A.segCap = h->segCap;
    A.pointFeat = d_pointFeat, A.featRef = (int4*)d_featRef, A.refStatic = d_refStatic, A.segPool = h->segPool, A.segCount = h->segCount;
    A.counts = d_counts;
    CS_HIP(hipSetDevice(h->device));   // (before the allocation below: the calling thread's current device may be another)
    if (nMap > h->aliveCap) {   // (grown on first use / for a larger map: a point nobody has seen yet has no live reference)
        if (h->alive) CS_HIP(hipFree(h->alive));
        h->alive = nullptr, h->aliveCap = 0;
        CS_HIP(hipMalloc((void**)&h->alive, (size_t)nMap));
        CS_HIP(hipMemsetAsync(h->alive, 1, (size_t)nMap, (hipStream_t)hip_stream));   // (1: look at every row once)
        h->aliveCap = nMap;
    }
    A.alive = h->alive;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].trackSpan) {
            cs_set_error("cs_feat_ref_advance_dev: null trackSpan in camera %d", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    if (nMap == 0 || (d_list && nList == 0)) return CS_OK;
    A.list = d_list, A.nList = nList;
    hipLaunchKernelGGL(k_feat_ref_advance, dim3(((d_list ? nList : nMap) + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// one ENTRY (map point m, camera c) of the same: the row's cameras side by side on the lanes of the point's own wave (k_advance_refine's
// refined rows: a lane walking the row's cameras one after the other was 10 us of dependent loads in front of every refine).  Returns
// whether the reference is alive in this frame; the row's `alive` byte is the caller's (any lane's yes).
__device__ __forceinline__ bool fr_advance_entry(const FrCore& A, const cs_poseupdate_cam* cam, int m, int c, int (&cnt)[5]) {
    const size_t e = (size_t)m * A.nCams + c;
    const int s = A.pointFeat[e];
    int4 ref = A.featRef[e];
    const int* span = cam[c].trackSpan;
    const bool on = s >= 0 && s < A.N;
    const int look = on ? s : ((ref.x >= 0 && ref.x < A.N) ? ref.x : 0);
    const int g1 = span[look], g2 = span[A.N + look];
    if (!on) {
        if (ref.x >= 0 && ref.x < A.N && ref.y == A.curFrame - 1 && g1 >= 0 && g1 <= ref.y && g2 == A.curFrame) {
            ref.x = -1, ++cnt[4];   // the same track, alive in this frame, no longer the point's: detached
            A.featRef[e] = ref;
        }
        return ref.x >= 0 && ref.y == A.curFrame;
    }
    const int f1 = g1;
    if (ref.x == s && f1 >= 0 && ref.y >= f1 && ref.y <= A.curFrame) {
        if (ref.y != A.curFrame) ++cnt[0];
        ref.y = A.curFrame;
    } else if (ref.x >= 0 && ref.y < A.curFrame) {
        const int idx = atomicAdd(A.segCount + c, 1);
        ++cnt[2];
        if (idx < A.segCap)
            A.segPool[(size_t)c * A.segCap + idx] = ref;
        else
            ++cnt[3];
        ref = make_int4(s, A.curFrame, A.curFrame, idx < A.segCap ? idx : -1);
    } else {
        ref = make_int4(s, A.curFrame, f1 >= 0 ? f1 : A.curFrame, -1), ++cnt[1];
    }
    A.featRef[e] = ref;
    if (A.refStatic) A.refStatic[e] = cam[c].isStatic ? cam[c].isStatic[s] : 1;
    return true;
}
// cs_feat_ref_advance_(list_)dev + cs_refine_map_points_ref_dev as ONE launch: the rows that are not refined are advanced by the first
// blocks (a thread per row, as k_feat_ref_advance); a row that IS refined (select[m] != 0; it has to be on `list`) is advanced by lane 0 of
// its own wave, which then re-triangulates the point from the references it has just written (k_update_points' refine mode).  Rows are
// independent (a pool segment is taken with an atomic), so the table and the map come out as the two calls leave them.
struct ArArgs : UpArgs {
    FrCore F;
    const int* list;     // the rows refined are among list[0 .. nList) (entries < 0 skipped)
    int nList, advAll;   // advAll: the first blocks advance every row of the map (the frame's first call), else the listed rows
    int blocksA, clearSelect;
    unsigned char* selectW;   // = select (written when clearSelect: the mark is consumed)
};
__global__ __launch_bounds__(256) void k_advance_refine(ArArgs A) {
    int cnt[5] = {0, 0, 0, 0, 0};
    if ((int)blockIdx.x < A.blocksA) {
        const int idx0 = blockIdx.x * 256 + threadIdx.x;
        const int m = A.advAll ? idx0 : (idx0 < A.nList ? A.list[idx0] : -1);
        if (m >= 0 && m < A.nMap && !A.select[m]) fr_advance_row(A.F, A.cam, m, cnt);
    } else {
        const int j = ((int)blockIdx.x - A.blocksA) * 4 + (int)threadIdx.x / 64, r = threadIdx.x % 64;
        const int m = j < A.nList ? A.list[j] : -1;
        if (m >= 0 && m < A.nMap && A.select[m]) {   // (uniform over the wave)
            const bool al = r < A.nCams && fr_advance_entry(A.F, A.cam, m, r, cnt);
            const bool anyAlive = __builtin_amdgcn_ballot_w64(al) != 0ull;
            if (r == 0) A.F.alive[m] = anyAlive ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            up_point(A, m, r);
            if (A.clearSelect && r == 0) A.selectW[m] = 0;
        }
    }
    if (A.F.counts) {
        __shared__ int sCnt[4][5];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int v = cnt[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) sCnt[wv][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            const int v = (sCnt[0][threadIdx.x] + sCnt[1][threadIdx.x]) + (sCnt[2][threadIdx.x] + sCnt[3][threadIdx.x]);
            if (v) atomicAdd(A.F.counts + threadIdx.x, v);
        }
    }
}

extern "C" int cs_feat_ref_advance_refine_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nMap, const int* d_pointFeat,
                                              int curFrame, cs_feat_ref* d_featRef, unsigned char* d_refStatic, int* d_counts, const int* d_list,
                                              int nList, int advanceAll, unsigned char* d_select, int clearSelect, double* d_mapPts,
                                              double* d_mapCov, double pixelErrVar) {
    if (!h || !cams || nMap < 1 || nList < 1 || !d_pointFeat || !d_featRef || !d_list || !d_select || !d_mapPts || !d_mapCov) {
        cs_set_error("cs_feat_ref_advance_refine_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (h->count < 1 || curFrame != h->lastFrame) {
        cs_set_error("cs_feat_ref_advance_refine_dev: the history's newest entry must be frame %d (it holds %d frame(s), the newest %d)", curFrame,
                     h->count, h->lastFrame);
        return CS_ERR_INVALID;
    }
    ArArgs A;
    memset((void*)&A, 0, sizeof(A));
    A.nMap = nMap, A.featRef = (const int4*)d_featRef, A.mapPts = d_mapPts, A.mapCov = d_mapCov, A.sigma = pixelErrVar;
    A.refine = 1, A.select = d_select, A.selectW = d_select, A.clearSelect = clearSelect ? 1 : 0;
    A.segPool = h->segPool, A.segCap = h->segCap, A.curFrame = h->lastFrame, A.stored = h->count < h->H ? h->count : h->H;
    A.nCams = h->nCams, A.N = h->N, A.H = h->H, A.head = h->head, A.nHist = hist_walk(h);
    A.histXY = h->xy, A.histR = h->R, A.histT = h->t, A.cen = h->cen;
    A.F.nCams = h->nCams, A.F.N = h->N, A.F.nMap = nMap, A.F.curFrame = curFrame, A.F.segCap = h->segCap;
    A.F.pointFeat = d_pointFeat, A.F.featRef = (int4*)d_featRef, A.F.refStatic = d_refStatic, A.F.segPool = h->segPool, A.F.segCount = h->segCount;
    A.F.counts = d_counts;
    A.list = d_list, A.nList = nList, A.advAll = advanceAll ? 1 : 0;
    CS_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (nMap > h->aliveCap) {
        if (h->alive) CS_HIP(hipFree(h->alive));
        h->alive = nullptr, h->aliveCap = 0;
        CS_HIP(hipMalloc((void**)&h->alive, (size_t)nMap));
        CS_HIP(hipMemsetAsync(h->alive, 1, (size_t)nMap, s));
        h->aliveCap = nMap;
    }
    A.F.alive = h->alive;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].K || !cams[c].iK || !cams[c].trackSpan) {
            cs_set_error("cs_feat_ref_advance_refine_dev: null pointer in camera %d (K, iK, trackSpan are read)", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    hist_centres(h, s);
    A.blocksA = ((advanceAll ? nMap : nList) + 255) / 256;
    hipLaunchKernelGGL(k_advance_refine, dim3(A.blocksA + (nList + 3) / 4), dim3(256), 0, s, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

extern "C" int cs_track_history_segments(const cs_track_history* h, cs_feat_seg** d_pool, int* cap, int** d_count) {
    if (!h) {
        cs_set_error("cs_track_history_segments: null handle");
        return CS_ERR_INVALID;
    }
    if (d_pool) *d_pool = (cs_feat_seg*)h->segPool;
    if (cap) *cap = h->segCap;
    if (d_count) *d_count = h->segCount;
    return CS_OK;
}

extern "C" int cs_track_history_segment_counts(const cs_track_history* h, int* counts) {
    if (!h || !counts) {
        cs_set_error("cs_track_history_segment_counts: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(h->device));
    CS_HIP(hipMemcpy(counts, h->segCount, sizeof(int) * h->nCams, hipMemcpyDeviceToHost));
    return CS_OK;
}

// host copy of `n` segments per camera ([nCams][n] cs_feat_seg) into the pools (tests, restoring a checkpoint); synchronous
extern "C" int cs_track_history_load_segments(cs_track_history* h, const cs_feat_seg* segs, int n) {
    if (!h || n < 0 || n > h->segCap || (n > 0 && !segs)) {
        cs_set_error("cs_track_history_load_segments: bad arguments (at most %d segments per camera)", h ? h->segCap : 0);
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(h->device));
    std::vector<int> cnt(h->nCams, n);
    for (int c = 0; c < h->nCams && n > 0; ++c)
        CS_HIP(hipMemcpy(h->segPool + (size_t)c * h->segCap, segs + (size_t)c * n, sizeof(int4) * n, hipMemcpyHostToDevice));
    CS_HIP(hipMemcpy(h->segCount, cnt.data(), sizeof(int) * h->nCams, hipMemcpyHostToDevice));
    return CS_OK;
}

// the first n segments of every camera's pool to the host ([nCams][n] cs_feat_seg; tests, saving a state); synchronous
extern "C" int cs_track_history_download_segments(const cs_track_history* h, cs_feat_seg* segs, int n) {
    if (!h || !segs || n < 0 || n > h->segCap) {
        cs_set_error("cs_track_history_download_segments: bad arguments (at most %d segments per camera)", h ? h->segCap : 0);
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(h->device));
    for (int c = 0; c < h->nCams && n > 0; ++c)
        CS_HIP(hipMemcpy(segs + (size_t)c * n, h->segPool + (size_t)c * h->segCap, sizeof(int4) * n, hipMemcpyDeviceToHost));
    return CS_OK;
}

namespace {
// the reference tables of a *_ref_dev call into the launch arguments
template <class Args>
void up_set_refs(Args& A, const cs_track_history* h) {
    A.segPool = h->segPool, A.segCap = h->segCap, A.curFrame = h->lastFrame, A.stored = h->count < h->H ? h->count : h->H;
}
}  // namespace

extern "C" int cs_update_new_poses_points_ref_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams,
                                                  const cs_feat_ref* d_featRef, const unsigned char* d_refStatic, int nMap, const int* d_lastFrame,
                                                  const unsigned char* d_isCurrent, int firstKeyFrame, double* d_mapPts, double* d_mapCov,
                                                  const unsigned char* d_mapFlags, double pixelErrVar, int* d_counts) {
    if (!h || !cams || nMap < 0 || (nMap > 0 && (!d_featRef || !d_mapPts || !d_mapCov || !d_mapFlags))) {
        cs_set_error("cs_update_new_poses_points_ref_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    UpArgs A;
    memset(&A, 0, sizeof(A));
    A.nMap = nMap, A.firstKeyFrame = firstKeyFrame;
    A.featRef = (const int4*)d_featRef, A.refStatic = d_refStatic, A.lastFrame = d_lastFrame, A.isCurrent = d_isCurrent;
    A.mapPts = d_mapPts, A.mapCov = d_mapCov, A.mapFlags = d_mapFlags;
    A.sigma = pixelErrVar;
    up_set_refs(A, h);
    return up_launch("cs_update_new_poses_points_ref_dev", h, hip_stream, cams, A, d_counts, 2);
}

extern "C" int cs_refine_map_points_ref_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, const cs_feat_ref* d_featRef,
                                            int nMap, const unsigned char* d_select, double* d_mapPts, double* d_mapCov, double pixelErrVar,
                                            int* d_count) {
    if (!h || !cams || nMap < 0 || (nMap > 0 && (!d_featRef || !d_mapPts || !d_mapCov))) {
        cs_set_error("cs_refine_map_points_ref_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    UpArgs A;
    memset(&A, 0, sizeof(A));
    A.nMap = nMap;
    A.featRef = (const int4*)d_featRef;
    A.mapPts = d_mapPts, A.mapCov = d_mapCov;
    A.sigma = pixelErrVar;
    A.refine = 1, A.select = d_select;
    up_set_refs(A, h);
    return up_launch("cs_refine_map_points_ref_dev", h, hip_stream, cams, A, d_count, 1);
}

namespace {
int cu_launch(const char* who, const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nPairs, const int* d_pf1,
              const int* d_pf2, const cs_feat_ref* d_ref1, const cs_feat_ref* d_ref2, const double* d_M1, const double* d_M2, double pixelErrVar,
              unsigned char* d_ok, double* d_M, double* d_cov) {
    const bool byRef = d_ref1 || d_ref2;
    if (!h || !cams || nPairs < 0 || h->nCams * 4 > 64 ||
        (nPairs > 0 && ((byRef ? (!d_ref1 || !d_ref2) : (!d_pf1 || !d_pf2)) || !d_M1 || !d_M2 || !d_ok || !d_M || !d_cov))) {
        cs_set_error("%s: bad arguments (at most 16 cameras)", who);
        return CS_ERR_INVALID;
    }
    if (h->count < 1) {
        cs_set_error("%s: the history holds no frame", who);
        return CS_ERR_INVALID;
    }
    if (nPairs == 0) return CS_OK;
    CuArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = h->nCams, A.N = h->N, A.H = h->H, A.head = h->head, A.nHist = hist_walk(h), A.nPairs = nPairs;
    A.pf1 = d_pf1, A.pf2 = d_pf2, A.M1 = d_M1, A.M2 = d_M2;
    A.ref1 = (const int4*)d_ref1, A.ref2 = (const int4*)d_ref2;
    if (byRef) up_set_refs(A, h);
    A.histXY = h->xy, A.histR = h->R, A.histT = h->t, A.cen = h->cen;
    A.sigma = pixelErrVar;
    A.ok = d_ok, A.M = d_M, A.cov = d_cov;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].K || !cams[c].iK || !cams[c].trackSpan) {
            cs_set_error("%s: null pointer in camera %d (K, iK, trackSpan are read)", who, c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    CS_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    hist_centres(h, s);
    hipLaunchKernelGGL(k_check_unify, dim3((nPairs + 3) / 4), dim3(256), 0, s, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}
}  // namespace

extern "C" int cs_check_unify_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nPairs, const int* d_pf1,
                                  const int* d_pf2, const double* d_M1, const double* d_M2, double pixelErrVar, unsigned char* d_ok, double* d_M,
                                  double* d_cov) {
    return cu_launch("cs_check_unify_dev", h, hip_stream, cams, nPairs, d_pf1, d_pf2, nullptr, nullptr, d_M1, d_M2, pixelErrVar, d_ok, d_M, d_cov);
}
extern "C" int cs_check_unify_ref_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int nPairs,
                                      const cs_feat_ref* d_ref1, const cs_feat_ref* d_ref2, const double* d_M1, const double* d_M2,
                                      double pixelErrVar, unsigned char* d_ok, double* d_M, double* d_cov) {
    return cu_launch("cs_check_unify_ref_dev", h, hip_stream, cams, nPairs, nullptr, nullptr, d_ref1, d_ref2, d_M1, d_M2, pixelErrVar, d_ok, d_M, d_cov);
}

extern "C" int cs_register_decide_merge_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int P, int mapBase,
                                            const int* d_slot, const int* d_flags, const unsigned char* d_mergeable, unsigned char* d_mapFlags,
                                            int* d_pointFeat, double* d_mapPts, double* d_mapCov, double pixelErrVar, unsigned char* d_attached,
                                            unsigned char* d_regged, void* d_scratch, int* d_counts, int onlyCam) {
    return cs_register_decide_merge_list_dev(h, hip_stream, cams, P, mapBase, nullptr, 0, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, d_mapPts,
                                             d_mapCov, pixelErrVar, d_attached, d_regged, d_scratch, d_counts, onlyCam);
}
// ... walking a LIST of points (the frame's current points in map order, cs_register_list_current_dev; entries < 0 behind it) instead of
// all P rows of the whole-map tables: the one wave's loops are as long as the list, not as the map's capacity, and the checkUnify
// verdicts it may need are evaluated side by side before it starts.  d_scratch: cs_register_decide_merge_scratch_bytes(P, nList, nCams).
extern "C" size_t cs_register_decide_merge_scratch_bytes(int P, int nList, int nCams) {
    if (P < 0 || nList < 0 || nCams < 1) return 0;
    const size_t offOk = ((size_t)P + 15) & ~(size_t)15, offM = (offOk + (size_t)nList * nCams + 15) & ~(size_t)15;
    return offM + sizeof(double) * 12 * (size_t)nList * nCams;
}
extern "C" int cs_register_decide_merge_list_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int P, int mapBase,
                                                 const int* d_list, int nList, const int* d_slot, const int* d_flags,
                                                 const unsigned char* d_mergeable, unsigned char* d_mapFlags, int* d_pointFeat, double* d_mapPts,
                                                 double* d_mapCov, double pixelErrVar, unsigned char* d_attached, unsigned char* d_regged,
                                                 void* d_scratch, int* d_counts, int onlyCam) {
    if (d_list && (nList < 0 || mapBase != 0 || P > 32 * DM_DIRTY_WORDS)) {
        cs_set_error("cs_register_decide_merge_list_dev: a list needs nList >= 0, mapBase 0 (it holds map indices) and at most 65536 map points");
        return CS_ERR_INVALID;
    }
    if (!h || !cams || P < 0 || mapBase < 0 || h->nCams * 4 > 64 || onlyCam >= h->nCams ||
        (P > 0 && (!d_slot || !d_flags || !d_mergeable || !d_mapFlags || !d_pointFeat || !d_mapPts || !d_mapCov || !d_attached || !d_regged || !d_scratch))) {
        cs_set_error("cs_register_decide_merge_dev: bad arguments (at most 16 cameras)");
        return CS_ERR_INVALID;
    }
    if (h->count < 1) {
        cs_set_error("cs_register_decide_merge_dev: the history holds no frame");
        return CS_ERR_INVALID;
    }
    DmArgs A;
    memset(&A, 0, sizeof(A));
    A.cu.nCams = h->nCams, A.cu.N = h->N, A.cu.H = h->H, A.cu.head = h->head, A.cu.nHist = hist_walk(h);
    A.cu.histXY = h->xy, A.cu.histR = h->R, A.cu.histT = h->t, A.cu.cen = h->cen;
    A.cu.sigma = pixelErrVar;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].K || !cams[c].iK || !cams[c].trackSpan || !cams[c].slot2map) {
            cs_set_error("cs_register_decide_merge_dev: null pointer in camera %d (K, iK, trackSpan are read, slot2map is written)", c);
            return CS_ERR_INVALID;
        }
        A.cu.cam[c] = cams[c];
    }
    A.P = P, A.mapBase = mapBase, A.onlyCam = onlyCam < 0 ? -1 : onlyCam;
    A.slot = d_slot, A.flags = d_flags, A.mergeable = d_mergeable, A.mapFlags = d_mapFlags, A.pointFeat = d_pointFeat;
    A.mapPts = d_mapPts, A.mapCov = d_mapCov, A.attached = d_attached, A.regged = d_regged, A.inVec = (unsigned char*)d_scratch, A.counts = d_counts;
    A.list = d_list, A.nList = nList;
    A.featRef = h->mergeFeatRef, A.refStatic = h->mergeRefStatic, A.segPoolW = h->segPool, A.segCount = h->segCount;
    if (A.featRef) A.cu.segPool = h->segPool, A.cu.segCap = h->segCap, A.cu.curFrame = h->lastFrame, A.cu.stored = h->count < h->H ? h->count : h->H;
    A.debug = cs_debug_get(CS_DBG_MERGE_PRINT) == 1;   // (cs_debug_set("merge_print", 1): the kernel prints its own account)
    CS_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (P == 0) {
        if (d_counts) CS_HIP(hipMemsetAsync(d_counts, 0, 4 * sizeof(int), s));
        return CS_OK;
    }
    hist_centres(h, s);
    if (d_list) {
        // the whole tables' out-flags and the walk's "touched" marks (scratch [0, P)) cleared by a wide launch (the one wave below only
        // touches listed rows); behind the marks in the scratch: the pre-checked verdicts [nList][nCams] and their 12 doubles each
        unsigned char* scr = (unsigned char*)d_scratch;
        const size_t offOk = ((size_t)P + 15) & ~(size_t)15, offM = (offOk + (size_t)nList * h->nCams + 15) & ~(size_t)15;
        A.preOk = scr + offOk, A.preM = (double*)(scr + offM);
        cs_small::List ops;
        ops.fill(d_attached, 0, (size_t)P * h->nCams), ops.fill(d_regged, 0, (size_t)P), ops.fill(scr, 0, (size_t)P);
        CS_HIP(ops.run(s));
        if (nList > 0 && onlyCam < 0)
            hipLaunchKernelGGL(k_merge_precheck, dim3((nList * h->nCams + 3) / 4), dim3(256), 0, s, A);
        else
            A.preOk = nullptr;
    }
    hipLaunchKernelGGL(k_decide_merge, dim3(1), dim3(64), 0, s, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

static int cls_reserve_list(const cs_track_history* h, int nMap, hipStream_t s);

// selectFused: k_pose_update has already built the worklist of this parity and the centres (cs_pose_update_classify_frame_dev): the worker only
static int cls_run(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int* d_pointFeat,
                                          int nMap, const int* d_featFrame, const int* d_featFirst, int curFrame, double* d_mapPts,
                                          double* d_mapCov, unsigned char* d_mapFlags, unsigned char* d_newPt, int* d_staticFrameNum,
                                          const int* d_firstFrame, double pixelVar, int* d_counts, bool selectFused, int fusedPar) {
    if (!h || !cams || nMap < 0 ||
        (nMap > 0 && (!d_pointFeat || !d_mapPts || !d_mapCov || !d_mapFlags || !d_newPt || !d_staticFrameNum || !d_firstFrame))) {
        cs_set_error("cs_map_points_classify_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (h->count < 1 || curFrame != h->lastFrame) {
        cs_set_error("cs_map_points_classify_dev: the history's newest entry must be frame %d (it holds %d frame(s), the newest %d)", curFrame,
                     h->count, h->lastFrame);
        return CS_ERR_INVALID;
    }
    ClsArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = h->nCams, A.N = h->N, A.nMap = nMap, A.H = h->H, A.head = h->head, A.nHist = hist_walk(h), A.curFrame = curFrame;
    A.pointFeat = d_pointFeat, A.featFrame = d_featFrame, A.featFirst = d_featFirst;
    A.featRef = h->clsFeatRef, A.refStatic = h->clsRefStatic, A.segPool = h->segPool, A.segCap = h->segCap;
    A.stored = h->count < h->H ? h->count : h->H;
    if (A.featRef) A.featFrame = nullptr, A.featFirst = nullptr;
    A.histXY = h->xy, A.histR = h->R, A.histT = h->t, A.cen = h->cen;
    A.mapPts = d_mapPts, A.mapCov = d_mapCov, A.mapFlags = d_mapFlags, A.newPt = d_newPt, A.staticFrameNum = d_staticFrameNum;
    A.firstFrame = d_firstFrame;
    A.sigma = pixelVar;
    A.counts = d_counts;
    for (int c = 0; c < h->nCams; ++c) {
        if (!cams[c].K || !cams[c].iK || !cams[c].trackSpan || !cams[c].isStatic) {
            cs_set_error("cs_map_points_classify_dev: null pointer in camera %d (K, iK, trackSpan, isStatic; slot2map if given is written)", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    CS_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (nMap == 0) {
        if (d_counts) CS_HIP(hipMemsetAsync(d_counts, 0, 2 * sizeof(int), s));
        return CS_OK;
    }
    {
        const int rc_ = cls_reserve_list(h, nMap, s);
        if (rc_ != CS_OK) return rc_;
    }
    A.list = h->clsList;
    if (selectFused) {
        A.par = fusedPar;   // (the parity the fused kernel listed into; cls_reserve_list moved h->clsPar on)
    } else {
        A.par = h->clsPar;
        h->clsPar ^= 1;
        hist_centres(h, s);
        hipLaunchKernelGGL(k_classify_select, dim3((nMap + 255) / 256), dim3(256), 0, s, A);
    }
    hipLaunchKernelGGL(k_map_points_classify, dim3(CLS_WAVES / 4), dim3(256), 0, s, A);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

extern "C" int cs_map_points_classify_dev(const cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int* d_pointFeat,
                                          int nMap, const int* d_featFrame, const int* d_featFirst, int curFrame, double* d_mapPts,
                                          double* d_mapCov, unsigned char* d_mapFlags, unsigned char* d_newPt, int* d_staticFrameNum,
                                          const int* d_firstFrame, double pixelVar, int* d_counts) {
    return cls_run(h, hip_stream, cams, d_pointFeat, nMap, d_featFrame, d_featFirst, curFrame, d_mapPts, d_mapCov, d_mapFlags, d_newPt,
                   d_staticFrameNum, d_firstFrame, pixelVar, d_counts, false, 0);
}

// From the next call on cs_map_points_classify_dev / cs_pose_update_classify_frame_dev take the points' features as references (d_featRef
// [nMap][nCams] cs_feat_ref, the table cs_feat_ref_advance_dev keeps; d_refStatic [nMap][nCams] or NULL); NULL: back to d_pointFeat alone.
extern "C" int cs_track_history_set_classify_refs(cs_track_history* h, cs_feat_ref* d_featRef, unsigned char* d_refStatic) {
    if (!h) {
        cs_set_error("cs_track_history_set_classify_refs: null history");
        return CS_ERR_INVALID;
    }
    h->clsFeatRef = (int4*)d_featRef, h->clsRefStatic = d_featRef ? d_refStatic : nullptr;
    return CS_OK;
}

// From the next call on cs_register_decide_merge(_list)_dev takes the points' features as references (d_featRef [P][nCams] cs_feat_ref, read AND
// written; d_refStatic [P][nCams] or NULL); NULL: this frame's features alone.
extern "C" int cs_track_history_set_merge_refs(cs_track_history* h, cs_feat_ref* d_featRef, unsigned char* d_refStatic) {
    if (!h) {
        cs_set_error("cs_track_history_set_merge_refs: null history");
        return CS_ERR_INVALID;
    }
    h->mergeFeatRef = (int4*)d_featRef, h->mergeRefStatic = d_featRef ? d_refStatic : nullptr;
    return CS_OK;
}

// the worklist's storage for a map of nMap points (grown on first use / for a larger map: the only allocation, and it waits for the device)
static int cls_reserve_list(const cs_track_history* h, int nMap, hipStream_t s) {
    if (nMap > h->clsCap) {
        if (h->clsList) CS_HIP(hipFree(h->clsList));
        h->clsList = nullptr, h->clsCap = 0;
        CS_HIP(hipMalloc((void**)&h->clsList, sizeof(int) * (2 + (size_t)nMap)));
        CS_HIP(hipMemsetAsync(h->clsList, 0, 2 * sizeof(int), s));
        h->clsCap = nMap, h->clsPar = 0;
    }
    return CS_OK;
}

// cs_pose_update_frame_dev + cs_map_points_classify_dev of the same frame as TWO launches instead of four: the gate's lane of a map point
// also decides whether mapPointsClassify examines it (k_classify_select's test, on the flags the gate has just left), the camera centres
// by walk depth (k_ring_centres) are blocks of the same launch; then the classification's worker.  Same results as the two calls.
extern "C" int cs_pose_update_classify_frame_dev(cs_track_history* h, void* hip_stream, const cs_poseupdate_cam* cams, int* d_pointFeat, int nMap,
                                                 const double* d_R, const double* d_t, double* d_mapPts, double* d_mapCov, unsigned char* d_mapFlags,
                                                 int largeErr, double pixelErrVar, int frame, int maxLen, int minLen, int minOutNum, double maxEpiErr,
                                                 int* d_numNodes, int* d_numOut, int* d_numDyn, const int* d_featFrame, const int* d_featFirst,
                                                 unsigned char* d_newPt, int* d_staticFrameNum, const int* d_firstFrame, double pixelVarClassify,
                                                 int* d_clsCounts) {
    if (!h) {
        cs_set_error("cs_pose_update_classify_frame_dev: null history");
        return CS_ERR_INVALID;
    }
    (void)maxLen;
    int rc = pu_check_common("cs_pose_update_classify_frame_dev", h->nCams, 0, h->nCams, cams, h->N, d_R, d_t);
    if (rc != CS_OK) return rc;
    if (nMap < 1 || !d_pointFeat || !d_mapPts || !d_mapCov || !d_mapFlags || !d_newPt || !d_staticFrameNum || !d_firstFrame) {
        cs_set_error("cs_pose_update_classify_frame_dev: null map pointer (or an empty map: use the two calls)");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    rc = cls_reserve_list(h, nMap, s);
    if (rc != CS_OK) return rc;
    PuArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = h->nCams, A.N = h->N, A.cam0 = 0, A.nCamsRun = h->nCams, A.R = d_R, A.t = d_t;
    pu_fill_gate(A, d_pointFeat, nMap, d_mapPts, d_mapCov, d_mapFlags, largeErr, pixelErrVar, d_numNodes, d_numOut);
    pu_advance(h, frame);
    pu_fill_dyn(A, h, minLen, minOutNum, maxEpiErr, d_numDyn);
    const int par = h->clsPar;
    h->clsPar ^= 1;
    A.clsList = h->clsList, A.clsPar = par, A.clsCurFrame = frame, A.clsCounts = d_clsCounts, A.clsFeatFrame = d_featFrame;
    A.cenOut = h->cen;
    rc = pu_launch("cs_pose_update_classify_frame_dev", h->device, hip_stream, A, cams, true, true);
    if (rc != CS_OK) return rc;
    h->cenVersion = h->ringVersion;   // (the centres of this state of the ring are what the launch has just written)
    return cls_run(h, hip_stream, cams, d_pointFeat, nMap, d_featFrame, d_featFirst, frame, d_mapPts, d_mapCov, d_mapFlags, d_newPt, d_staticFrameNum,
                   d_firstFrame, pixelVarClassify, d_clsCounts, true, par);
}
