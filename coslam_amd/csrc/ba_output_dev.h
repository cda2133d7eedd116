// ba_output_dev.h -- what RobustBundleRTS::output() does with a finished bundle adjustment, on the device (included by ba.hip
// inside its anonymous namespace).
//
// Reference src/app/SL_CoSLAMRobustBA.cpp:273-316: the adjusted key poses go back into the CamPoseItems the key frames share with
// the tracker (:283-285) and into the fixed nodes of the camera graphs (:288-294), every adjusted point back into its MapPoint
// (:299) -- a point with ANY outlier measurement becomes false (:300-309) -- then the non-key frames' poses are relaxed over the
// camera graphs (:311-313: every frame from the window's first key frame to the NEWEST one, constructCameraGraphs :182-227) and
// every map point seen since is triangulated again (updateNewPosesPoints, :314).  The reference's BA thread does this under the
// lock it shares with tracking (src/app/SL_CoSLAM.cpp:1713-1720); here the solve's worker thread only PACKS the result into a
// record (k_ba_output_pack, behind the solve's last kernel on the solve's stream) and the stream that owns the map applies the
// record between two frames (cs_ba_output_apply_dev).  The record is self-contained so that, with the cameras sharded over
// several GPUs, the rank that solved a window can broadcast it and every rank applies the same bytes to its replica of the map.
//
// record (device memory, one slot of a small ring; 8-byte aligned):
//   int hdr[32]          [0] C  [1] P  [2] nObs  [3] nKf  [4] nCams  [5] sequence number  [6] ok  [7] sequence number + 1, stored
//                        by its own launch BEHIND the pack (k_ba_output_publish): what a stream that must not start before the
//                        record is complete polls (k_ba_output_wait)  [8 + j] frame of key frame j
//   double Rs[maxC][9]   key cameras, index = key frame (oldest first) x nCams + camera
//   double Ts[maxC][3]
//   double pts[maxP][3]
//   int pointMap[maxP]   map index of point i (RobustBundleRTS::int2MapPt)
//   uint8 ptOut[maxP]    1: at least one measurement of point i is an outlier
constexpr int BO_HDR_INTS = 32;

struct BoLayout {
    int maxC, maxP;
    size_t offRs, offTs, offPts, offMap, offOut, bytes;
};
static inline BoLayout bo_layout(int maxC, int maxP) {
    BoLayout L;
    L.maxC = maxC, L.maxP = maxP;
    L.offRs = sizeof(int) * BO_HDR_INTS;
    L.offTs = L.offRs + sizeof(double) * 9 * (size_t)maxC;
    L.offPts = L.offTs + sizeof(double) * 3 * (size_t)maxC;
    L.offMap = L.offPts + sizeof(double) * 3 * (size_t)maxP;
    L.offOut = L.offMap + sizeof(int) * (size_t)maxP;
    L.bytes = (L.offOut + (size_t)maxP + 255) & ~(size_t)255;
    return L;
}

struct BoPackArgs {
    int C, P, nObs, nKf, nCams, seq, ok;
    int kfFrame[16];
    const double *Rs, *Ts, *pts;
    const int *obs_ptr, *outlier, *pointMap;
    unsigned char* rec;
    BoLayout L;
};
__global__ __launch_bounds__(256) void k_ba_output_pack(BoPackArgs A) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    int* hdr = (int*)A.rec;
    if (q < BO_HDR_INTS) {
        int v = 0;
        if (q == 0) v = A.C;
        if (q == 1) v = A.P;
        if (q == 2) v = A.nObs;
        if (q == 3) v = A.nKf;
        if (q == 4) v = A.nCams;
        if (q == 5) v = A.seq;
        if (q == 6) v = A.ok;
        if (q >= 8 && q < 8 + 16) v = A.kfFrame[q - 8];
        if (q != 7) hdr[q] = v;   // (hdr[7] is the release word: k_ba_output_publish)
    }
    double* oR = (double*)(A.rec + A.L.offRs);
    double* oT = (double*)(A.rec + A.L.offTs);
    double* oM = (double*)(A.rec + A.L.offPts);
    int* oMap = (int*)(A.rec + A.L.offMap);
    unsigned char* oOut = A.rec + A.L.offOut;
    if (q < 9 * A.C) oR[q] = A.Rs[q];
    if (q < 3 * A.C) oT[q] = A.Ts[q];
    if (q < 3 * A.P) oM[q] = A.pts[q];
    if (q < A.P) {
        oMap[q] = A.pointMap[q];
        int any = 0;
        for (int j = A.obs_ptr[q]; j < A.obs_ptr[q + 1]; ++j) any |= A.outlier[j] > 0;
        oOut[q] = (unsigned char)any;
    }
}

// hdr[7] = seq + 1 behind the pack, visible to other streams' running kernels (agent scope)
__global__ void k_ba_output_publish(int* hdr, int seq) {
    __threadfence();
    __hip_atomic_store(hdr + 7, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane polls hdr[7] until the record `seq` (or a later one) is published: the stream behind this launch then reads a complete
// record without the host ever waiting for the solve.  The solve runs on ANOTHER stream (the workspace's worker): the two make
// progress side by side; a bound on the wait (wall-clock ticks of 10 ns) turns a solver that never publishes into an error word
// instead of a hung stream.
__global__ void k_ba_output_wait(const int* hdr, int seq, long long maxTicks, int* err) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(hdr + 7, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq + 1) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > maxTicks) {
            atomicAdd(err, 1);
            return;
        }
    }
}

// the points of a record into the map (SL_CoSLAMRobustBA.cpp:297-309): M <- pt3Ds[i]; any outlier measurement: setFalse
// (MapPoint::setFalse keeps bUncertain and leaves TYPE_MAP_FALSE: CS_MAP_DYNAMIC cleared, as cs_map_points_classify_dev does)
// Is the record the one the caller means to apply?  seq >= 0: its sequence number (hdr[5]) and its release word (hdr[7], stored
// behind the pack) must say so, and its key frames (hdr[8 + j]) must be the window's.  A wait that gave up (k_ba_output_wait's
// timeout) leaves a slot holding the record of nSlots solves ago -- complete, ok = 1, but of ANOTHER window: it applies nothing.
// (the window being applied: its request number and ALL of its key frames' numbers, hdr[8 + j] -- they need not be equally spaced)
struct BoWin {
    int seq, nKf;
    int kf[16];
};
__device__ __forceinline__ bool bo_record_is(const int* hdr, const BoWin& W) {
    if (!hdr[6]) return false;
    if (W.seq < 0) return true;
    if (hdr[5] != W.seq || hdr[7] != W.seq + 1) return false;
    bool same = true;
    for (int j = 0; j < 16; ++j)   // (a window requested before its ring had filled holds hdr[3] < nKf key frames: those it has must agree)
        if (j < W.nKf && j < hdr[3] && hdr[8 + j] != W.kf[j]) same = false;
    return same;
}
// one thread: counts a record that was refused although it claims to hold a result (err[1]; err[0] counts the waits that gave up)
__global__ void k_ba_output_check(const int* hdr, BoWin W, int* err) {
    if (hdr[6] && !bo_record_is(hdr, W)) atomicAdd(err + 1, 1);
}

__global__ __launch_bounds__(256) void k_ba_output_points(const unsigned char* __restrict__ rec, BoLayout L, int nMap, double* __restrict__ mapPts,
                                                          unsigned char* __restrict__ mapFlags, int* __restrict__ counts, BoWin W,
                                                          int writePts, int writeFalse) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int* hdr = (const int*)rec;
    if (!bo_record_is(hdr, W) || i >= hdr[1]) return;
    const int m = ((const int*)(rec + L.offMap))[i];
    if (m < 0 || m >= nMap) return;
    const double* p = (const double*)(rec + L.offPts) + 3 * (size_t)i;
    if (writePts) mapPts[3 * (size_t)m] = p[0], mapPts[3 * (size_t)m + 1] = p[1], mapPts[3 * (size_t)m + 2] = p[2];
    if (writeFalse && rec[L.offOut + i]) {
        const unsigned char fl = mapFlags[m];
        mapFlags[m] = (unsigned char)((fl & ~CS_MAP_DYNAMIC) | CS_MAP_FALSE);
        if (counts && !(fl & CS_MAP_FALSE)) atomicAdd(counts, 1);
    }
}

// the record's key poses into (a) the fixed nodes of the camera graphs' node arrays ([nCams][nNodes]: node of key frame j of
// camera c = c * nNodes + nodeOf[j], nodeOf[j] = the key frame's number - the first key frame's) and (b) the window ring's copies of those key frames (slotOf[j] < 0: the ring no longer
// holds it), so that the next window's parse starts from the adjusted key poses the way the reference's shared CamPoseItems do
struct BoPosesArgs {
    int nKf, nCams, nNodes;
    BoWin win;   // bo_record_is
    int nodeOf[16], slotOf[16];
    double *nodeR, *nodeT, *winR, *winT;
};
__global__ __launch_bounds__(256) void k_ba_output_poses(const unsigned char* __restrict__ rec, BoLayout L, BoPosesArgs A) {
    const int q = blockIdx.x * 256 + threadIdx.x, i = q / 12, e = q - 12 * i;
    const int* hdr = (const int*)rec;
    if (!bo_record_is(hdr, A.win) || i >= A.nKf * A.nCams || i >= hdr[0]) return;
    const int j = i / A.nCams, c = i - j * A.nCams;
    const double v = e < 9 ? ((const double*)(rec + L.offRs))[9 * (size_t)i + e] : ((const double*)(rec + L.offTs))[3 * (size_t)i + (e - 9)];
    const size_t node = (size_t)c * A.nNodes + (size_t)A.nodeOf[j];
    if (e < 9)
        A.nodeR[9 * node + e] = v;
    else
        A.nodeT[3 * node + (e - 9)] = v;
    if (A.winR && A.slotOf[j] >= 0) {
        const size_t w = (size_t)A.slotOf[j] * A.nCams + c;
        if (e < 9)
            A.winR[9 * w + e] = v;
        else
            A.winT[3 * w + (e - 9)] = v;
    }
}

// the newest node of every camera's chain = the camera's CURRENT pose (m_camPos.current()): what the next frame's pose solve starts from
__global__ __launch_bounds__(256) void k_ba_output_tail(int nCams, int nNodes, const double* __restrict__ newR, const double* __restrict__ newT,
                                                        double* __restrict__ Rcur, double* __restrict__ tcur) {
    const int q = blockIdx.x * 256 + threadIdx.x, c = q / 12, e = q - 12 * c;
    if (c >= nCams) return;
    const size_t node = (size_t)c * nNodes + (nNodes - 1);
    if (e < 9)
        Rcur[9 * (size_t)c + e] = newR[9 * node + e];
    else
        tcur[3 * (size_t)c + (e - 9)] = newT[3 * node + (e - 9)];
}

// staticSnap[m] = MapPoint::isLocalStatic() from the CS_MAP_* flag byte (cs_ba_solve_window_flags_async)
__global__ __launch_bounds__(256) void k_win_static_from_flags(int nMap, unsigned char* __restrict__ st) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < nMap) st[m] = (st[m] & (CS_MAP_DYNAMIC | CS_MAP_FALSE)) ? 0 : 1;
}
