// ba_window_dev.h -- the bundle adjuster's inputs built ON THE DEVICE from the tracker's own records (included by ba.hip inside
// its anonymous namespace).
//
// Replaces RobustBundleRTS::addKeyFrames / addPoints / parseInputs (reference src/app/SL_CoSLAMRobustBA.cpp:37-78,109-165) and
// the CoSLAM::requestForBA walk that feeds them (src/app/SL_CoSLAM.cpp:1731-1784): the last K key frames of all cameras are the
// cameras of the problem (index = key frame x nCams + camera, oldest first: addKeyFrames' order), every STATIC map point with
// more than one feature point in those key frames is a point (in map order: parseInputs walks a std::map keyed by MapPoint*,
// i.e. by address; tests/cxx/ref_ba_dropin_test.cpp keeps the map in one array so that address order = index order), its
// measurements in camera order, one per key camera (a later feature point of the same (frame, camera) replaces an earlier one:
// vecFeatPts[camId] = fpt, :141).  The reference builds this from pointer-linked lists on the host under the BA mutex; here a
// key frame is snapshot as structure-of-arrays records (cs_ba_window_push_dev: the hand-back's xy / state / slot2map + the
// poses just solved) and the flat problem is three launches straight into the solver's workspace.
//   k_win_snapshot   per (camera, slot): the slot's undistorted pixel into the ring; pointFeat[m] = the LAST slot (list order
//                    = slot order, GPUKLT::addToFeaturePoints) whose feature of this frame carries map point m
//   k_win_count      per map point: feature points over the window's key cameras
//   k_win_scan       one workgroup: which points stay (> 1 feature point), their index, their first measurement
//   k_win_fill       per kept point: pt3Ds entry, obs_ptr, the Meas2D list in camera order (+ the measurement -> point and the
//                    dense (point, camera) -> measurement tables of the solver); per camera K, R, t
// and the camera-pair lists of the Schur kernel (built on the host for uploaded problems) on the device too:
//   k_pairs_count / k_pairs_fill   one wave per camera pair, ballot-compaction over the points in index order.

struct WinDev {
    int nCams, nKf, N, nMap;
    int count;                 // key frames held (<= nKf)
    int slotOf[16];            // ring slot of key frame j (oldest first), j < count
    const double* xy;          // [nKf][nCams][2N]
    const int* pf;             // [nKf][nCams][nMap]
    const double* K;           // [nKf][nCams][9]
    const double* R;           // [nKf][nCams][9]
    const double* t;           // [nKf][nCams][3]
    const unsigned char* mapStatic;  // [nMap] or null: MapPoint::isLocalStatic()
    const double* mapPts;      // [nMap][3]
    int* cnt;                  // [nMap] feature points per map point over the window
    int* ptIndex;              // [nMap] index among the kept points, or -1
    int* obsStart;             // [nMap] first measurement of a kept point
    int* totals;               // [4]: P, nObs, maxObs, reserved
};

struct WinSnapArgs {
    int nCams, N, nMap;
    const double* xy[16];
    const int* state[16];
    const int* slot2map[16];
    double* xyOut;   // [nCams][2N]
    int* pfOut;      // [nCams][nMap], preset to -1
};
__global__ __launch_bounds__(256) void k_win_snapshot(WinSnapArgs A) {
    const int c = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    if (s >= A.N) return;
    A.xyOut[(size_t)c * 2 * A.N + s] = A.xy[c][s];
    A.xyOut[(size_t)c * 2 * A.N + A.N + s] = A.xy[c][A.N + s];
    const int st = A.state[c][s], m = A.slot2map[c][s];
    if ((st == 0 || st == 1) && m >= 0 && m < A.nMap) atomicMax(A.pfOut + (size_t)c * A.nMap + m, s);
}

// one WAVE per map point, lane = (key frame, camera) entry of the window (64 entries at a time): the point's feature count is a
// ballot (a lane per point walking 40 entries one load after the other: 14 us in the loop)
__global__ __launch_bounds__(256) void k_win_count(WinDev Wd) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= Wd.nMap) return;
    int n = 0;
    if (!Wd.mapStatic || Wd.mapStatic[m]) {
        const int C = Wd.count * Wd.nCams;
        for (int e0 = 0; e0 < C; e0 += 64) {
            const int e = e0 + lane;
            bool in = false;
            if (e < C) {
                const int j = e / Wd.nCams, c = e - j * Wd.nCams;
                in = Wd.pf[((size_t)Wd.slotOf[j] * Wd.nCams + c) * Wd.nMap + m] >= 0;
            }
            n += __popcll(__builtin_amdgcn_ballot_w64(in));
        }
    }
    if (lane == 0) Wd.cnt[m] = n;
}

// exclusive scans over the map points in index order (one workgroup of 1024: chunks of consecutive points per thread)
__global__ __launch_bounds__(1024) void k_win_scan(WinDev Wd) {
    __shared__ int sP[1024], sO[1024], sM[1024];
    const int tid = threadIdx.x, per = (Wd.nMap + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < Wd.nMap) ? lo + per : Wd.nMap;
    int p = 0, o = 0, mx = 0;
    for (int m = lo; m < hi; ++m) {
        const int n = Wd.cnt[m];
        if (n > 1) {  // parseInputs: nfpts > 1 (:120-121)
            p += 1;
            o += n;
            mx = n > mx ? n : mx;
        }
    }
    sP[tid] = p, sO[tid] = o, sM[tid] = mx;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan of the per-thread totals
        const int a = tid >= d ? sP[tid - d] : 0, b = tid >= d ? sO[tid - d] : 0, c = tid >= d ? sM[tid - d] : 0;
        __syncthreads();
        sP[tid] += a, sO[tid] += b, sM[tid] = c > sM[tid] ? c : sM[tid];
        __syncthreads();
    }
    int bp = sP[tid] - p, bo = sO[tid] - o;
    for (int m = lo; m < hi; ++m) {
        const int n = Wd.cnt[m];
        if (n > 1) {
            Wd.ptIndex[m] = bp++;
            Wd.obsStart[m] = bo;
            bo += n;
        } else {
            Wd.ptIndex[m] = -1;
            Wd.obsStart[m] = 0;
        }
    }
    if (tid == 1023) {
        Wd.totals[0] = sP[1023];
        Wd.totals[1] = sO[1023];
        Wd.totals[2] = sM[1023];
        Wd.totals[3] = 0;
    }
}

struct WinFillOut {
    double *Ks, *Rs, *Ts, *pts, *obs_xy;
    int *obs_ptr, *obs_cam, *pointMap;
    int *obs_pt, *obs_of;   // the measurement's point; the dense (point, camera) -> measurement table (-1: none)
};
// one WAVE per map point, lane = (key frame, camera) entry: the measurements of a kept point in camera order (:146-151) are the
// set lanes of a ballot in lane order, a lane's measurement index = the point's start + the set lanes below it
__global__ __launch_bounds__(256) void k_win_fill(WinDev Wd, WinFillOut O) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int C = Wd.count * Wd.nCams;
    if (g < C) {  // the key cameras: K, R, t as CamPoseItem holds them (addKeyCamera, :80-89)
        const int j = g / Wd.nCams, c = g - j * Wd.nCams;
        const size_t src = (size_t)Wd.slotOf[j] * Wd.nCams + c;
        for (int q = 0; q < 9; ++q) {
            O.Ks[9 * g + q] = Wd.K[9 * src + q];
            O.Rs[9 * g + q] = Wd.R[9 * src + q];
        }
        for (int q = 0; q < 3; ++q) O.Ts[3 * g + q] = Wd.t[3 * src + q];
    }
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= Wd.nMap) return;
    const int i = Wd.ptIndex[m];
    if (m == Wd.nMap - 1 && lane == 0) {  // closing entry of obs_ptr
        const int P = Wd.totals[0];
        O.obs_ptr[P] = Wd.totals[1];
    }
    if (i < 0) return;
    int o = Wd.obsStart[m];
    if (lane == 0) {
        O.pointMap[i] = m;
        for (int q = 0; q < 3; ++q) O.pts[3 * (size_t)i + q] = Wd.mapPts[3 * (size_t)m + q];
        O.obs_ptr[i] = o;
    }
    for (int e0 = 0; e0 < C; e0 += 64) {
        const int e = e0 + lane;
        int sl = -1;
        size_t src = 0;
        if (e < C) {
            const int j = e / Wd.nCams, c = e - j * Wd.nCams;
            src = (size_t)Wd.slotOf[j] * Wd.nCams + c;
            sl = Wd.pf[src * Wd.nMap + m];
        }
        const bool in = sl >= 0;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
        const int mine = o + __popcll(mask & ((1ull << lane) - 1ull));
        if (e < C) O.obs_of[(size_t)i * C + e] = in ? mine : -1;
        if (in) {
            O.obs_pt[mine] = i;
            O.obs_cam[mine] = e;
            O.obs_xy[2 * (size_t)mine] = Wd.xy[src * 2 * Wd.N + sl];
            O.obs_xy[2 * (size_t)mine + 1] = Wd.xy[src * 2 * Wd.N + Wd.N + sl];
        }
        o += __popcll(mask);
    }
}

// ---- camera-pair lists on the device -------------------------------------------------------------------------------------
// pair (ca <= cb), id = ca C - ca (ca - 1) / 2 + (cb - ca): the measurements {oa, ob, point} of the points both cameras see,
// ascending point index (ca == cb: every measurement of the camera) -- what cs_ba_upload builds on the host.  One wave per
// pair walks the points 64 at a time through the dense (point, camera) table.
__global__ __launch_bounds__(64) void k_pairs_count(int C, const int* totals, const int* obs_of, int* pairCnt) {
    const int pid = blockIdx.x, lane = threadIdx.x, P = totals[0];  // (the number of kept points is still on its way to the host)
    int ca = 0, rest = pid;
    while (rest >= C - ca) {
        rest -= C - ca;
        ++ca;
    }
    const int cb = ca + rest;
    int n = 0;
    for (int i0 = 0; i0 < P; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < P && obs_of[(size_t)i * C + ca] >= 0 && obs_of[(size_t)i * C + cb] >= 0;
        n += __popcll(__builtin_amdgcn_ballot_w64(in));
    }
    if (lane == 0) pairCnt[pid] = n;
}
__global__ __launch_bounds__(1024) void k_pairs_scan(int nPairs, const int* pairCnt, int* pairPtr, int* total) {
    __shared__ int sS[1024];
    const int tid = threadIdx.x, per = (nPairs + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < nPairs) ? lo + per : nPairs;
    int s = 0;
    for (int q = lo; q < hi; ++q) s += pairCnt[q];
    sS[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int a = tid >= d ? sS[tid - d] : 0;
        __syncthreads();
        sS[tid] += a;
        __syncthreads();
    }
    int b = sS[tid] - s;
    for (int q = lo; q < hi; ++q) {
        pairPtr[q] = b;
        b += pairCnt[q];
    }
    if (tid == 1023) {
        pairPtr[nPairs] = sS[1023];
        *total = sS[1023];
    }
}
__global__ __launch_bounds__(64) void k_pairs_fill(int C, int P, const int* obs_of, const int* pairPtr, int4* pairEnt) {
    const int pid = blockIdx.x, lane = threadIdx.x;
    int ca = 0, rest = pid;
    while (rest >= C - ca) {
        rest -= C - ca;
        ++ca;
    }
    const int cb = ca + rest;
    int base = pairPtr[pid];
    for (int i0 = 0; i0 < P; i0 += 64) {
        const int i = i0 + lane;
        int oa = -1, ob = -1;
        if (i < P) {
            oa = obs_of[(size_t)i * C + ca];
            ob = obs_of[(size_t)i * C + cb];
        }
        const bool in = oa >= 0 && ob >= 0;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
        if (in) pairEnt[base + __popcll(mask & ((1ull << lane) - 1ull))] = make_int4(oa, ob, i, 0);
        base += __popcll(mask);
    }
}

// the camera-indexed measurement lists (cam_ptr [C + 1], cam_obs [nObs]: per camera its measurements in ascending order) that the
// small-order solver's sliced Schur kernels walk -- what cs_ba_upload builds on the host.  One workgroup, a wave per camera (round
// robin), ballot compaction over the measurements 64 at a time.  Only enqueued for reduced systems of order <= 36.
__global__ __launch_bounds__(1024) void k_cam_lists(int C, const int* __restrict__ totals, const int* __restrict__ obs_cam, int* __restrict__ cam_ptr,
                                                    int* __restrict__ cam_obs) {
    __shared__ int cnt[257];
    const int nObs = totals[1], wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = wv; c < C; c += 16) {
        int n = 0;
        for (int o0 = 0; o0 < nObs; o0 += 64) {
            const int o = o0 + lane;
            n += __popcll(__builtin_amdgcn_ballot_w64(o < nObs && obs_cam[o] == c));
        }
        if (lane == 0) cnt[c] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0;
        for (int c = 0; c < C; ++c) {
            const int n = cnt[c];
            cnt[c] = a, cam_ptr[c] = a;
            a += n;
        }
        cam_ptr[C] = a;
    }
    __syncthreads();
    for (int c = wv; c < C; c += 16) {
        int base = cnt[c];
        for (int o0 = 0; o0 < nObs; o0 += 64) {
            const int o = o0 + lane;
            const bool in = o < nObs && obs_cam[o] == c;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
            if (in) cam_obs[base + __popcll(m & ((1ull << lane) - 1ull))] = o;
            base += __popcll(m);
        }
    }
}
