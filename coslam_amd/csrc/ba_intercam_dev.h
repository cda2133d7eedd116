// ba_intercam_dev.h -- InterCamPoseEstimator::addMapPoints on the device (included by ba.hip inside its anonymous namespace).
//
// Reference src/app/SL_InterCamPoseEstimator.cpp:18-91.  The problem bundleAdjustRobust(0, ..., m_numStatic, ...) gets (:95):
//   cameras   the current pose of every camera (:29-37)
//   static    per camera chooseStaticFeatPts (src/app/SL_SingleSLAM.cpp:345-397), run again on the records as they stand NOW (the
//             frame's classification may have detached a feature or changed a point's type since the hand-back voted): per 40 x 40
//             block, over the tracks in slot order whose newest feature is static or belongs to a certainly static map point, the
//             first track with a map point, else the first of the longest; in block order every chosen feature that has a map
//             point: the point as the map holds it, ONE measurement (:41-52).
//   dynamic   chooseDynamicFeatPts per camera (SL_SingleSLAM.cpp:398-447: per block the track whose feature's map point is seen by
//             the most cameras -- the first such track in slot order -- among features of map points with numVisCam >= 2 that are
//             certainly dynamic or uncertain and new), the union of their map points in map order (a std::map keyed by MapPoint*,
//             :57-64), the first maxDyn + 1 of them (`if (k > maxDyn) continue`, :72), each with one measurement per camera that holds
//             a feature of THIS frame (:77-83).  numVisCam = cameras with a feature of this frame (MapPoint::updateVisCamNum).
// The flat problem lands in a staging record in the layout the window parse writes (k_win_fill), so the solver's set-up behind it --
// camera-pair lists, lane plan, segments -- is shared with cs_ba_solve_window_async.
struct IcCamArgs {
    const double* xy[16];
    const int* state[16];
    const int* slot2map[16];
    const int* trackSpan[16];
    const unsigned char* isStatic[16];
    const double* K[16];
};
struct IcStage {           // one staging record (device): what a request built, until the worker has copied it
    double *Ks, *Rs, *Ts, *pts, *obs_xy;
    int *obs_ptr, *obs_cam, *pointMap, *obs_pt, *obs_of, *totals;   // totals: P, nObs, maxObs, nStatic
    // scratch of the build
    double *stPts, *stXY;  // [nCams][ptsStride][3] / [2]
    int *stMap, *stCount;  // [nCams][ptsStride] / [nCams]
    unsigned char* dynMark;  // [nMap]
};
struct IcArgs {
    int nCams, N, W, H, nColBlk, nRowBlk, blkW, blkH, ptsStride, nMap, maxDyn;
    const double *R, *t, *mapPts;
    const unsigned char *mapFlags, *newPt;
    const int* pointFeat;  // [nMap][nCams]
    IcCamArgs cam;
    IcStage st;
};

constexpr int IC_MAX_BLOCKS = 1024;
// one workgroup per camera: (1) chooseStaticFeatPts' block vote and, in block order, the winners that carry a map point; (2) the
// block vote of chooseDynamicFeatPts, its winners' map points marked
// (1024 threads: the slots' dependent loads -- slot -> map point -> flags / the point's row -- two trips deep instead of eight)
__global__ __launch_bounds__(1024) void k_ic_gather(IcArgs A) {
    __shared__ unsigned long long skey[IC_MAX_BLOCKS];
    __shared__ unsigned key[IC_MAX_BLOCKS];
    __shared__ unsigned char sflag[IC_MAX_BLOCKS];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int N = A.N;
    const double* xy = A.cam.xy[c];
    const int* state = A.cam.state[c];
    const int* s2m = A.cam.slot2map[c];
    const int nBlk = A.nColBlk * A.nRowBlk;
    for (int q = tid; q < nBlk; q += 1024) skey[q] = 0, key[q] = 0;
    __syncthreads();
    // ---- static (:351-384): `tracks[bi]` replaced only while it holds an unmapped feature -- by the first mapped one, else by a longer one
    for (int s = tid; s < N; s += 1024) {
        const int st = state[s];
        if (st != 0 && st != 1) continue;
        int m = s2m[s];
        if (m >= A.nMap) m = -1;
        const bool certainStatic = m >= 0 && (A.mapFlags[m] & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN)) == 0;
        if (!(A.cam.isStatic[c][s] || certainStatic)) continue;   // :358-359
        const double x = xy[s], y = xy[N + s];
        const int bx = (int)(x / (double)A.blkW), by = (int)(y / (double)A.blkH);
        if (bx >= A.nColBlk || by >= A.nRowBlk || bx < 0 || by < 0) continue;
        const int f1 = A.cam.trackSpan[c][s], f2 = A.cam.trackSpan[c][N + s];
        const unsigned long long len = f1 >= 0 ? (unsigned long long)(f2 - f1 + 1) : 1ull;
        const unsigned long long order = (unsigned long long)(N - 1 - s);
        atomicMax(&skey[by * A.nColBlk + bx], m >= 0 ? ((1ull << 62) | order) : ((len << 24) | order));
    }
    __syncthreads();
    for (int b = tid; b < nBlk; b += 1024) sflag[b] = (skey[b] >> 62) ? 1 : 0;   // `if (!fp->mpt) continue` (:44-45)
    __syncthreads();
    for (int b = tid; b < nBlk; b += 1024) {
        if (!sflag[b]) continue;
        int k = 0;
        for (int q = 0; q < b; ++q) k += sflag[q];
        if (k >= A.ptsStride) continue;
        const int slot = N - 1 - (int)(skey[b] & 0xFFFFFFull), m = s2m[slot];
        const size_t o = (size_t)c * A.ptsStride + k;
        A.st.stPts[3 * o] = A.mapPts[3 * (size_t)m], A.st.stPts[3 * o + 1] = A.mapPts[3 * (size_t)m + 1], A.st.stPts[3 * o + 2] = A.mapPts[3 * (size_t)m + 2];
        A.st.stXY[2 * o] = xy[slot], A.st.stXY[2 * o + 1] = xy[N + slot];
        A.st.stMap[o] = m;
    }
    if (tid == 0) {
        int n = 0;
        for (int q = 0; q < nBlk; ++q) n += sflag[q];
        A.st.stCount[c] = n < A.ptsStride ? n : A.ptsStride;
    }
    // ---- dynamic: per block the feature whose map point the most cameras see; ties: the lowest slot (the first in slot order)
    for (int s = tid; s < N; s += 1024) {
        const int st = state[s];
        if (st != 0 && st != 1) continue;          // the track is empty (its tail is not a feature of this frame)
        const int m = s2m[s];
        if (m < 0 || m >= A.nMap) continue;        // :410
        int nVis = 0;
        for (int cc = 0; cc < A.nCams; ++cc) nVis += A.pointFeat[(size_t)m * A.nCams + cc] >= 0;
        if (nVis < 2) continue;                    // :412
        const unsigned char fl = A.mapFlags[m];
        const bool unc = (fl & CS_MAP_UNCERTAIN) != 0;
        const bool certainDyn = !unc && (fl & (CS_MAP_DYNAMIC | CS_MAP_FALSE)) == CS_MAP_DYNAMIC;
        if (!(certainDyn || (unc && A.newPt[m]))) continue;   // :414-415
        const double x = xy[s], y = xy[N + s];
        const int bx = (int)(x / (double)A.blkW), by = (int)(y / (double)A.blkH);
        if (bx >= A.nColBlk || by >= A.nRowBlk || bx < 0 || by < 0) continue;
        atomicMax(&key[by * A.nColBlk + bx], ((unsigned)nVis << 24) | (unsigned)(0xFFFFFF - s));
    }
    __syncthreads();
    for (int q = tid; q < nBlk; q += 1024) {
        const unsigned k = key[q];
        if (k) A.st.dynMark[s2m[0xFFFFFF - (int)(k & 0xFFFFFF)]] = 1;
    }
}

// one workgroup: the flat problem from the cameras' static lists and the marked dynamic points
__global__ __launch_bounds__(1024) void k_ic_assemble(IcArgs A) {
    __shared__ int sBase[17];
    __shared__ int sScan[1024];
    __shared__ int sDynMap[64], sDynObs[65];
    __shared__ int sNDyn;
    const int tid = threadIdx.x, C = A.nCams;
    if (tid == 0) {
        int b = 0;
        for (int c = 0; c < C; ++c) sBase[c] = b, b += A.st.stCount[c];
        sBase[C] = b;
    }
    // the dynamic points in map order: chunked scan over the marks
    const int per = (A.nMap + 1023) / 1024, lo = min(tid * per, A.nMap), hi = min(lo + per, A.nMap);
    int cnt = 0;
    for (int m = lo; m < hi; ++m) cnt += A.st.dynMark[m] != 0;
    sScan[tid] = cnt;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? sScan[tid - d] : 0;
        __syncthreads();
        sScan[tid] += v;
        __syncthreads();
    }
    int rank = sScan[tid] - cnt;   // marked points before this chunk
    const int keep = A.maxDyn + 1 < 64 ? A.maxDyn + 1 : 64;   // `if (k > maxDyn) continue`: points 0 .. maxDyn
    for (int m = lo; m < hi; ++m)
        if (A.st.dynMark[m]) {
            if (rank < keep) sDynMap[rank] = m;
            ++rank;
            A.st.dynMark[m] = 0;   // (left clean for the next request that builds into this staging record: no fill launch in front of it)
        }
    if (tid == 1023) sNDyn = sScan[1023] < keep ? sScan[1023] : keep;
    __syncthreads();
    const int nStatic = sBase[C], nDyn = sNDyn, P = nStatic + nDyn;
    // measurements of the dynamic points: the cameras holding a feature of this frame -- counted side by side, then strung together
    __shared__ int sDynCnt[64];
    if (tid < nDyn) {
        int n = 0;
        for (int c = 0; c < C; ++c) n += A.pointFeat[(size_t)sDynMap[tid] * C + c] >= 0;
        sDynCnt[tid] = n;
    }
    __syncthreads();
    if (tid == 0) {
        int o = nStatic;
        for (int k = 0; k < nDyn; ++k) sDynObs[k] = o, o += sDynCnt[k];
        sDynObs[nDyn] = o;
        int mx = nStatic > 0 ? 1 : 0;
        for (int k = 0; k < nDyn; ++k) mx = max(mx, sDynObs[k + 1] - sDynObs[k]);
        A.st.totals[0] = P, A.st.totals[1] = o, A.st.totals[2] = mx, A.st.totals[3] = nStatic;
    }
    __syncthreads();
    // cameras: K, the current poses
    for (int q = tid; q < C * 21; q += 1024) {
        const int c = q / 21, e = q - 21 * c;
        if (e < 9)
            A.st.Ks[9 * c + e] = A.cam.K[c][e], A.st.Rs[9 * c + e] = A.R[9 * c + e];
        else if (e < 12)
            A.st.Ts[3 * c + (e - 9)] = A.t[3 * c + (e - 9)];
    }
    for (int q = tid; q < P * C; q += 1024) A.st.obs_of[q] = -1;
    __syncthreads();
    // static points: one measurement each, in camera order
    for (int q = tid; q < C * A.ptsStride; q += 1024) {
        const int c = q / A.ptsStride, k = q - c * A.ptsStride;
        if (k >= A.st.stCount[c]) continue;
        const int i = sBase[c] + k;
        const size_t o = (size_t)c * A.ptsStride + k;
        A.st.pts[3 * i] = A.st.stPts[3 * o], A.st.pts[3 * i + 1] = A.st.stPts[3 * o + 1], A.st.pts[3 * i + 2] = A.st.stPts[3 * o + 2];
        A.st.obs_xy[2 * i] = A.st.stXY[2 * o], A.st.obs_xy[2 * i + 1] = A.st.stXY[2 * o + 1];
        A.st.obs_ptr[i] = i, A.st.obs_cam[i] = c, A.st.obs_pt[i] = i, A.st.pointMap[i] = A.st.stMap[o];
        A.st.obs_of[(size_t)i * C + c] = i;
    }
    // dynamic points
    if (tid < nDyn) {
        const int k = tid, m = sDynMap[k], i = nStatic + k;
        A.st.pts[3 * i] = A.mapPts[3 * (size_t)m], A.st.pts[3 * i + 1] = A.mapPts[3 * (size_t)m + 1], A.st.pts[3 * i + 2] = A.mapPts[3 * (size_t)m + 2];
        A.st.pointMap[i] = m;
        A.st.obs_ptr[i] = sDynObs[k];
        int o = sDynObs[k];
        for (int c = 0; c < C; ++c) {
            const int s = A.pointFeat[(size_t)m * C + c];
            if (s < 0) continue;
            A.st.obs_xy[2 * o] = A.cam.xy[c][s], A.st.obs_xy[2 * o + 1] = A.cam.xy[c][A.N + s];
            A.st.obs_cam[o] = c, A.st.obs_pt[o] = i;
            A.st.obs_of[(size_t)i * C + c] = o;
            ++o;
        }
    }
    if (tid == 0) A.st.obs_ptr[P] = sDynObs[nDyn];
}
