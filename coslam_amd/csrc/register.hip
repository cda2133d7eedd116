// register.hip -- the search step of CoSLAM's map-point registration for all points x all cameras in one launch, gfx950
// (SURVEY.md 8f-2).
//
// Replaces, inside the three registration loops of the reference
//   CoSLAM::curStaticPointRegInGroup        src/app/SL_CoSLAM.cpp:731-757
//   CoSLAM::curDynamicPointRegInGroup       src/app/SL_CoSLAM.cpp:955-980
//   CoSLAM::activeMapPointRegisterInGroup   src/app/SL_CoSLAM.cpp:1118-1145
// the per (map point, camera) statements that are the same in all three -- isAtCameraBack, project, the image test,
// getProjectionCovMat, searchMahaNearestFeatPt (src/app/SL_SingleSLAM.cpp:1141-1164: a serial walk over the camera's
// feature list of the current frame) -- plus the candidate's own term of staticCheckMergability (SL_CoSLAM.cpp:714-729).
// The reference runs them point by point, camera by camera, under the BA mutex: P x C x N distance evaluations on one
// host core per frame (1500 x 8 x 2000 = 24 M).  What the loops do with a candidate afterwards (NCC comparison, the walk
// over the candidate's earlier frames, pointer updates, refineMapPoint, checkUnify) stays with the caller.
//
// Layout: one workgroup per (tile of 64 map points, camera); lane = point, the camera's feature list (the hand-back's SoA
// records x[N], y[N], state[N]) staged in LDS once and walked with broadcast reads by 16 waves, 1/16 of the list each; the
// waves' minima are merged on (distance, slot) -- exactly the serial loop's "strict <, first wins".  The first version ran
// one wave per pair with the lanes striding the list out of L2: 12000 waves x 2000 x 20 B = 480 MB of L1 fills per launch,
// 59 us in the benchmark loop; staging cuts that to 15 MB.  A pair the reference skips (feature of this frame already
// attached, behind the camera, outside the image) only idles its lane.
//
// searchMahaNearestFeatPt scales the inverse covariance by 1 / maxDist and never compares the distance with a threshold:
// the nearest feature in that metric wins however far it is.  Reproduced as is; the scaled distance is returned so the
// caller can gate.  isAtCameraBack / project / getProjectionCovMat / mat22Inv / mahaDist2 are un-vendored LibVisualSLAM:
// definitions in DESIGN.md, same arithmetic order as the test oracle, no FMA contraction.
#include "cs_common.h"

#include <cfloat>

#pragma clang fp contract(off)

namespace {

constexpr int RG_MAX_CAMS = 16;

struct RgArgs {
    int nCams, N, W, H, nPass, cam0;   // cameras cam0 .. cam0 + gridDim.y - 1 of the nCams-wide tables
    cs_register_pass pass[2];  // blockIdx.z: the passes of a frame share ONE launch (cs_register_search_passes_dev)
    cs_register_cam cam[RG_MAX_CAMS];
};

struct Proj {
    double u, v, w;
    double KR[9];
};

__device__ __forceinline__ void projection_cov(const Proj& q, const double* __restrict__ cov, double sigma, double var[4]) {
    const double ww = q.w * q.w;
    double J[6], JC[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        J[j] = (q.KR[j] * q.w - q.u * q.KR[6 + j]) / ww;
        J[3 + j] = (q.KR[3 + j] * q.w - q.v * q.KR[6 + j]) / ww;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) JC[3 * i + j] = (J[3 * i] * cov[j] + J[3 * i + 1] * cov[3 + j]) + J[3 * i + 2] * cov[6 + j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double s = (JC[3 * i] * J[3 * j] + JC[3 * i + 1] * J[3 * j + 1]) + JC[3 * i + 2] * J[3 * j + 2];
            var[2 * i + j] = (i == j) ? s + sigma * sigma : s;
        }
}

__device__ __forceinline__ void mat22_inv(const double A[4], double iA[4]) {
    const double det = A[0] * A[3] - A[1] * A[2];
    iA[0] = A[3] / det;
    iA[1] = -A[1] / det;
    iA[2] = -A[2] / det;
    iA[3] = A[0] / det;
}

__device__ __forceinline__ double maha_dist2(double mx, double my, double bx, double by, const double ivar[4]) {
    const double dx = mx - bx, dy = my - by;
    return dx * (ivar[0] * dx + ivar[1] * dy) + dy * (ivar[2] * dx + ivar[3] * dy);
}

// The kernel has to FIT beside the persistent tracker, whose workgroups hold two waves of 160 VGPRs on every SIMD and 128 of a compute
// unit's 160 KB of LDS for as long as its main kernel runs: 192 VGPRs per SIMD lane and 32 KB of LDS are what is left.  8 waves of 96
// VGPRs (two per SIMD) and a 512-feature stage (8 KB + 6 KB of tables) fit; round 3's 16 waves (four per SIMD: 384 VGPRs) with the
// whole list staged (44 KB) did not -- a search that became ready before the tracker's main kernel had retired waited for it.
// Measured in the loop, alternating on one box (profiles/r04_ab_runs.txt): 16 waves / whole list 2141-2148 frames/s, 8 / 1024 2159-2167,
// 8 / 512 2173-2179, 8 / 256 2159, 16 / 512 2134-2141, 4 / 512 2128-2140.
constexpr int RG_WAVES = 8;       // waves per workgroup: each scans 1 / 8 of the staged features for the block's 64 points
constexpr int RG_ROWS = RG_WAVES > 6 ? RG_WAVES : 6;   // rows of the minima table (it first carries the 6 projection values)
constexpr int RG_CHUNK = 512;     // features staged in LDS at a time; the list is scanned chunk by chunk

// One workgroup = 64 map points x one camera.  Lane = point: its projection and scaled inverse covariance live in
// registers.  The camera's feature list is staged in LDS once per workgroup (x, y as doubles; a slot that is not in this
// frame's list is staged as NaN, so its distance compares false and it can never win -- no state test in the loop), and
// every wave walks its share of it with BROADCAST reads: one LDS read serves 64 points.  Per wave the walk is in slot
// order with a strict <, the waves' minima are merged lexicographically on (distance, slot): the serial loop's answer.
// (Round 4 built the alternative -- features binned into 32-pixel cells, a point looks at 3 x 3 cells and falls back to the full scan
// when it cannot prove the answer global: 33 x fewer distance evaluations, bit-identical tables, 18.6 + 10.4 us alone -- and measured it
// in the loop: 2114 / 2118 frames/s against this kernel's 2168 / 2159.  Scattered reads and a second launch cost more next to the tracker
// than 50 M broadcast-fed evaluations.  Deleted; profiles/r04_ab_runs.txt.)
__global__ __launch_bounds__(64 * RG_WAVES) void k_register_search(RgArgs A) {
    extern __shared__ double lds[];
    CS_POSE_STREAM_PRIO();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = A.cam0 + blockIdx.y;
    const cs_register_pass& Q = A.pass[blockIdx.z];
    // Q.list: the pass's points are list[0 .. P) (map indices, < 0: no point) -- the tables stay indexed by the MAP index, so a compact
    // list of the frame's current points (cs_register_list_current_dev) costs one workgroup per 64 LISTED points and camera
    const int j = blockIdx.x * 64 + lane;
    if (Q.list && (int)(blockIdx.x * 64) < Q.P && Q.list[blockIdx.x * 64] < 0) return;   // (a compact list: this tile lies behind its end)
    const int p = Q.list ? (j < Q.P ? Q.list[j] : -1) : (j < Q.P ? j : -1);
    const cs_register_cam& C = A.cam[c];
    const int N = A.N;
    const int CH = N < RG_CHUNK ? N : RG_CHUNK;
    double* sx = lds;
    double* sy = lds + CH;
    double* cd = lds + 2 * CH;                        // [RG_ROWS][64]
    int* ci = (int*)(cd + RG_ROWS * 64);              // [RG_WAVES][64]
    const size_t o = (size_t)(p < 0 ? 0 : p) * A.nCams + c;
    int outSlot = -1, outFlags = 0;
    double m0 = 0, m1 = 0, var[4] = {0, 0, 0, 0}, outDist = 0;
    double ivar[4] = {0, 0, 0, 0};
    bool search = false;
    Proj q;
    if (wave == 0 && p >= 0 && Q.pointFeat[o] < 0) {  // SL_CoSLAM.cpp:737-738: no feature of this frame attached in this camera yet
        const double *K = C.K, *R = C.R, *t = C.t, *M = Q.M + 3 * (size_t)p;
        const double X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0];
        const double Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1];
        const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
        if (Z < 0.0) {  // :740-742 isAtCameraBack
            outSlot = -2;
        } else {
            q.u = (K[0] * X + K[1] * Y) + K[2] * Z;
            q.v = (K[3] * X + K[4] * Y) + K[5] * Z;
            q.w = (K[6] * X + K[7] * Y) + K[8] * Z;
            m0 = q.u / q.w;  // :744-745 project
            m1 = q.v / q.w;
            if (m0 < 0 || m0 >= (double)A.W || m1 < 0 || m1 >= (double)A.H) {  // :746-748
                outSlot = -3;
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) q.KR[3 * i + j] = (K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j]) + K[3 * i + 2] * R[6 + j];
                projection_cov(q, Q.cov + 9 * (size_t)p, Q.sigmaSearch, var);  // :750-753
                mat22_inv(var, ivar);                                          // SL_SingleSLAM.cpp:1148-1149
                // (the certainly dynamic points of a pass that serves both registrations: their own scale, SL_CoSLAM.cpp:973)
                const bool dynPt = Q.mapFlags && (Q.mapFlags[p] & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN)) == CS_MAP_DYNAMIC;
                const double sc = 1 / (dynPt ? Q.maxDistDynamic : Q.maxDist);
#pragma unroll
                for (int k = 0; k < 4; ++k) ivar[k] = ivar[k] * sc;
                search = true;
            }
        }
    }
    if (__syncthreads_or(search)) {
        // the projection and the scaled inverse covariance were worked out by wave 0 only: hand them to the other waves
        double* sq = cd;  // [6][64], overwritten by the minima after the scan
        if (wave == 0) {
            sq[lane] = m0;
            sq[64 + lane] = m1;
#pragma unroll
            for (int k = 0; k < 4; ++k) sq[(2 + k) * 64 + lane] = ivar[k];
        }
        __syncthreads();
        if (wave != 0) {
            m0 = sq[lane];
            m1 = sq[64 + lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) ivar[k] = sq[(2 + k) * 64 + lane];
        }
        // ---- searchMahaNearestFeatPt, SL_SingleSLAM.cpp:1141-1164 ----
        const double* __restrict__ xs = C.xy;
        const double* __restrict__ ys = C.xy + N;
        const int* __restrict__ st = C.state;
        double dMin = DBL_MAX;
        int iMin = 0x7fffffff;
        for (int base = 0; base < N; base += CH) {
            const int n = (N - base) < CH ? (N - base) : CH;
            if (base) __syncthreads();
            for (int i = threadIdx.x; i < n; i += 64 * RG_WAVES) {
                const int s = st[base + i];
                const bool in = (s == 0 || s == 1);
                sx[i] = in ? xs[base + i] : __builtin_nan("");
                sy[i] = in ? ys[base + i] : __builtin_nan("");
            }
            __syncthreads();
            const int per = (n + RG_WAVES - 1) / RG_WAVES;
            const int lo = wave * per, hi = (lo + per) < n ? (lo + per) : n;
            for (int i = lo; i < hi; ++i) {
                const double d = maha_dist2(m0, m1, sx[i], sy[i], ivar);
                if (d < dMin) {
                    dMin = d;
                    iMin = base + i;
                }
            }
        }
        cd[wave * 64 + lane] = dMin;  // (every wave read its copy of sq before the staging barrier above)
        ci[wave * 64 + lane] = iMin;
        __syncthreads();
        if (wave == 0 && search) {
#pragma unroll 4
            for (int w = 1; w < RG_WAVES; ++w) {
                const double d2 = cd[w * 64 + lane];
                const int i2 = ci[w * 64 + lane];
                if (d2 < dMin || (d2 == dMin && i2 < iMin)) {
                    dMin = d2;
                    iMin = i2;
                }
            }
            if (iMin == 0x7fffffff) {
                outSlot = -4;
            } else {
                outSlot = iMin;
                outDist = dMin;
                if (C.slot2map[iMin] < 0) outFlags |= 1;               // :759 pFeat->mpt == 0
                if (C.isDynamic ? C.isDynamic[iMin] != 0 : (C.isStatic && C.isStatic[iMin] == 0)) outFlags |= 2;  // :758 pFeat->type
                double v2[4], iv[4];                                   // staticCheckMergability, the candidate itself (:716-725)
                projection_cov(q, Q.cov + 9 * (size_t)p, Q.sigmaMerge, v2);
                mat22_inv(v2, iv);
                if (!(maha_dist2(m0, m1, xs[iMin], ys[iMin], iv) > 1.0)) outFlags |= 4;
            }
        }
    }
    if (wave == 0 && p >= 0) {
        Q.slot[o] = outSlot;
        Q.flags[o] = outFlags;
        Q.dist[o] = outDist;
        Q.m[2 * o] = m0;
        Q.m[2 * o + 1] = m1;
#pragma unroll
        for (int k = 0; k < 4; ++k) Q.var[4 * o + k] = var[k];
    }
}

// ---- the frame's CURRENT map points as a compact list ------------------------------------------------------------------------------
// CoSLAM::currentMapPointsRegister walks curMapPts: the points with a feature of this frame in at least one camera (mapStateUpdate,
// src/app/SL_CoSLAM.cpp:1176-1194, moves every other point off the list; :734 / :958 ask numVisCam > 0 again) -- wherever they sit
// in the map, the points genNewMapPoints has just appended included.  One workgroup: a point is listed when it lies below the
// live count, is not false (neither registration loop visits a false point) and holds a feature of this frame; the list keeps the
// map's order (the order the reference's walks visit the points in).  list[count .. nMap) = -1; the rows of the points that
// left the list since the previous call get slot = -1 in `slotTable` (nMap x nCams, optional; it starts at -1 everywhere and only
// listed rows are ever written), so that the tables never carry a stale candidate.
__global__ __launch_bounds__(1024) void k_register_list(int nCams, int nMap, const int* __restrict__ mapCount, const int* __restrict__ pointFeat,
                                                        const unsigned char* __restrict__ mapFlags, int* __restrict__ list,
                                                        int* __restrict__ listCount, int* __restrict__ slotTable, int listCap,
                                                        int* __restrict__ overflow) {
    // pass 1, coalesced: thread t looks at the points t, t + 1024, ... (a wave = 64 consecutive rows of pointFeat), the verdicts go into
    // a bitmap in LDS (and the unlisted points' rows of slotTable to -1); pass 2: thread t owns `per` consecutive points of the bitmap
    // (the list keeps the map's order): count, ONE scan over the block, write.  (Earlier forms: 15 rounds of scan over 1024 points,
    // 35 us; a thread reading its own 15 rows of pointFeat, 64 cache lines per wave load, 68 us.)
    __shared__ unsigned long long bits[1024];   // 65536 points
    __shared__ int waveSum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int live = mapCount ? (*mapCount < nMap ? *mapCount : nMap) : nMap;
    const bool vec4 = (nCams & 3) == 0 && (reinterpret_cast<uintptr_t>(pointFeat) & 15) == 0;
    const bool vec4s = (nCams & 3) == 0 && (reinterpret_cast<uintptr_t>(slotTable) & 15) == 0;
    const int oldCount = listCount ? (*listCount < nMap ? (*listCount < 0 ? 0 : *listCount) : nMap) : nMap;   // `list` still holds the previous call's
    for (int w = tid; w < (nMap + 63) / 64; w += 1024) bits[w] = 0ull;
    __syncthreads();
    // (eight trips' loads in flight together: the rows are independent, a trip by itself is one dependent round of memory latency)
    for (int pb = wv * 64; pb < live; pb += 8 * 1024) {
        int4 v0[8], v1[8];
        unsigned char fl[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = pb + 1024 * k + lane;
            v0[k] = v1[k] = make_int4(-1, -1, -1, -1), fl[k] = CS_MAP_FALSE;
            if (p < live) {
                fl[k] = mapFlags ? mapFlags[p] : 0;
                if (vec4 && nCams <= 8) {   // (rows of 4 or 8 ints on a 16-byte boundary: one or two 16-byte loads)
                    const int4* row = reinterpret_cast<const int4*>(pointFeat + (size_t)p * nCams);
                    v0[k] = row[0];
                    if (nCams == 8) v1[k] = row[1];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p0 = pb + 1024 * k, p = p0 + lane;
            if (p0 >= live) break;   // (uniform over the wave)
            bool in = false;
            if (p < live && !(fl[k] & CS_MAP_FALSE)) {
                if (vec4 && nCams <= 8) {
                    in = v0[k].x >= 0 || v0[k].y >= 0 || v0[k].z >= 0 || v0[k].w >= 0 || v1[k].x >= 0 || v1[k].y >= 0 || v1[k].z >= 0 || v1[k].w >= 0;
                } else if (vec4) {
                    const int4* row = reinterpret_cast<const int4*>(pointFeat + (size_t)p * nCams);
                    for (int c = 0; c < nCams / 4; ++c) {
                        const int4 v = row[c];
                        in |= v.x >= 0 || v.y >= 0 || v.z >= 0 || v.w >= 0;
                    }
                } else {
                    for (int c = 0; c < nCams; ++c) in |= pointFeat[(size_t)p * nCams + c] >= 0;
                }
            }
            const unsigned long long b = __builtin_amdgcn_ballot_w64(in);
            if (lane == 0) bits[p0 >> 6] = b;
        }
    }
    __syncthreads();
    // the rows of the points that were on the list a call ago and are not any more lose their candidates (every other unlisted row
    // lost them when it left the list, or never had any: the table starts at -1) -- a few hundred rows instead of the whole map's
    if (slotTable) {
        for (int q = tid; q < oldCount; q += 1024) {
            const int p = list[q];
            if (p < 0 || p >= nMap || ((bits[p >> 6] >> (p & 63)) & 1ull)) continue;
            if (vec4s) {
                int4* row = reinterpret_cast<int4*>(slotTable + (size_t)p * nCams);
                for (int c = 0; c < nCams / 4; ++c) row[c] = make_int4(-1, -1, -1, -1);
            } else {
                for (int c = 0; c < nCams; ++c) slotTable[(size_t)p * nCams + c] = -1;
            }
        }
    }
    __syncthreads();   // (the previous list has been read: it may be overwritten)
    const int per = (nMap + 1023) / 1024;
    const int q0 = tid * per, q1 = (q0 + per) < nMap ? (q0 + per) : nMap;
    int cnt = 0;
    for (int p = q0; p < q1; ++p) cnt += (int)((bits[p >> 6] >> (p & 63)) & 1ull);
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) waveSum[wv] = inc;
    __syncthreads();
    int off = inc - cnt, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wv) off += waveSum[w];
        total += waveSum[w];
    }
    // listCap: the passes behind this list (search, running mergability, merge walk, candidate records) cover its first listCap
    // entries.  A current point beyond the cap is NOT listed this frame: its row of slotTable is cleared, so that the decision -- which
    // visits every point with a feature -- finds no candidate for it instead of an older frame's; *overflow counts such points.
    const int kept = total < listCap ? total : listCap;
    for (int p = q0; p < q1; ++p)
        if ((bits[p >> 6] >> (p & 63)) & 1ull) {
            if (off < listCap) list[off] = p;
            else if (slotTable)
                for (int c = 0; c < nCams; ++c) slotTable[(size_t)p * nCams + c] = -1;
            ++off;
        }
    for (int q = kept + tid; q < nMap; q += 1024) list[q] = -1;
    if (tid == 0 && listCount) *listCount = kept;
    if (tid == 0 && overflow && total > listCap) atomicAdd(overflow, total - listCap);
}

int check_args(const char* who, int nCams, const cs_register_cam* cams, int N, int W, int H, int P, double sigmaSearch,
               double maxDist, double sigmaMerge) {
    if (nCams < 1 || nCams > RG_MAX_CAMS || !cams || N < 1 || W < 1 || H < 1 || P < 0 || !(maxDist > 0) || !(sigmaSearch >= 0) ||
        !(sigmaMerge >= 0)) {
        cs_set_error("%s: bad arguments (1..%d cameras, N >= 1, maxDist > 0)", who, RG_MAX_CAMS);
        return CS_ERR_INVALID;
    }
    return CS_OK;
}

}  // namespace

extern "C" int cs_register_list_current_dev(int device, void* hip_stream, int nCams, int nMap, const int* d_mapCount, const int* d_pointFeat,
                                            const unsigned char* d_mapFlags, int* d_list, int* d_listCount, int* d_slotTable) {
    return cs_register_list_current_cap_dev(device, hip_stream, nCams, nMap, d_mapCount, d_pointFeat, d_mapFlags, d_list, d_listCount, d_slotTable,
                                            nMap, nullptr);
}

extern "C" int cs_register_list_current_cap_dev(int device, void* hip_stream, int nCams, int nMap, const int* d_mapCount, const int* d_pointFeat,
                                                const unsigned char* d_mapFlags, int* d_list, int* d_listCount, int* d_slotTable, int listCap,
                                                int* d_overflow) {
    if (nCams < 1 || nCams > RG_MAX_CAMS || nMap < 0 || nMap > 65536 || listCap < 0 || (nMap > 0 && (!d_pointFeat || !d_list))) {
        cs_set_error("cs_register_list_current_dev: bad arguments (at most 65536 map points)");
        return CS_ERR_INVALID;
    }
    if (nMap == 0) return CS_OK;
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_register_list, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, nCams, nMap, d_mapCount, d_pointFeat, d_mapFlags, d_list,
                       d_listCount, d_slotTable, listCap < nMap ? listCap : nMap, d_overflow);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_register_search_passes_dev(int device, void* hip_stream, int nCams, const cs_register_cam* cams, int N, int W, int H,
                                             int nPass, const cs_register_pass* passes) {
    return cs_register_search_passes_range_dev(device, hip_stream, nCams, 0, nCams, cams, N, W, H, nPass, passes);
}

// cameras cam0 .. cam0 + nCamsRun - 1 only: their columns of the nCams-wide tables (a rank that holds every camera's records but
// searches only for the cameras it owns)
extern "C" int cs_register_search_passes_range_dev(int device, void* hip_stream, int nCams, int cam0, int nCamsRun, const cs_register_cam* cams,
                                                   int N, int W, int H, int nPass, const cs_register_pass* passes) {
    if (cam0 < 0 || nCamsRun < 0 || cam0 + nCamsRun > nCams) {
        cs_set_error("cs_register_search_passes_range_dev: camera range %d + %d of %d", cam0, nCamsRun, nCams);
        return CS_ERR_INVALID;
    }
    if (nPass < 1 || nPass > 2 || !passes) {
        cs_set_error("cs_register_search_passes_dev: 1 or 2 passes");
        return CS_ERR_INVALID;
    }
    RgArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = nCams, A.N = N, A.W = W, A.H = H, A.nPass = nPass, A.cam0 = cam0;
    int maxP = 0;
    for (int k = 0; k < nPass; ++k) {
        const cs_register_pass& q = passes[k];
        int rc = check_args("cs_register_search_passes_dev", nCams, cams, N, W, H, q.P, q.sigmaSearch, q.maxDist, q.sigmaMerge);
        if (rc != CS_OK) return rc;
        if (q.P > 0 && (!q.M || !q.cov || !q.pointFeat || !q.slot || !q.m || !q.var || !q.dist || !q.flags)) {
            cs_set_error("cs_register_search_passes_dev: null pointer in pass %d", k);
            return CS_ERR_INVALID;
        }
        if (q.mapFlags && !(q.maxDistDynamic > 0)) {
            cs_set_error("cs_register_search_passes_dev: pass %d has mapFlags but maxDistDynamic <= 0", k);
            return CS_ERR_INVALID;
        }
        A.pass[k] = q;
        if (q.P > maxP) maxP = q.P;
    }
    if (maxP == 0 || nCamsRun == 0) return CS_OK;
    for (int c = cam0; c < cam0 + nCamsRun; ++c) {
        const cs_register_cam& q = cams[c];
        if (!q.K || !q.R || !q.t || !q.xy || !q.state || !q.slot2map) {
            cs_set_error("cs_register_search_passes_dev: null pointer in camera %d", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = q;
    }
    CS_HIP(hipSetDevice(device));
    const int CH = N < RG_CHUNK ? N : RG_CHUNK;
    const size_t ldsBytes = (size_t)2 * CH * sizeof(double) + (size_t)RG_ROWS * 64 * sizeof(double) + (size_t)RG_WAVES * 64 * sizeof(int);
    hipLaunchKernelGGL(k_register_search, dim3((unsigned)((maxP + 63) / 64), (unsigned)nCamsRun, (unsigned)nPass), dim3(64 * RG_WAVES), ldsBytes,
                       (hipStream_t)hip_stream, A);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_register_search_dev(int device, void* hip_stream, int nCams, const cs_register_cam* cams, int N, int W, int H,
                                      int P, const double* d_M, const double* d_cov, const int* d_pointFeat, double sigmaSearch,
                                      double maxDist, double sigmaMerge, int* d_slot, double* d_m, double* d_var, double* d_dist,
                                      int* d_flags) {
    int rc = check_args("cs_register_search_dev", nCams, cams, N, W, H, P, sigmaSearch, maxDist, sigmaMerge);
    if (rc != CS_OK) return rc;
    if (P == 0) return CS_OK;
    if (!d_M || !d_cov || !d_pointFeat || !d_slot || !d_m || !d_var || !d_dist || !d_flags) {
        cs_set_error("cs_register_search_dev: null pointer");
        return CS_ERR_INVALID;
    }
    cs_register_pass q;
    memset(&q, 0, sizeof(q));
    q.P = P, q.sigmaSearch = sigmaSearch, q.maxDist = maxDist, q.sigmaMerge = sigmaMerge;
    q.M = d_M, q.cov = d_cov, q.pointFeat = d_pointFeat, q.slot = d_slot, q.m = d_m, q.var = d_var, q.dist = d_dist, q.flags = d_flags;
    return cs_register_search_passes_dev(device, hip_stream, nCams, cams, N, W, H, 1, &q);
}

namespace {

// ---- the decision behind the current-static search: who attaches which feature -------------------------------------------------------
// CoSLAM::curStaticPointsRegInGroup / curStaticPointRegInGroup with bMerge == false (src/app/SL_CoSLAM.cpp:854-898, 731-830) walk the
// points one after the other: for every camera o, the certainly static points with a feature of this frame in o, in map order; each
// walks the cameras in order and attaches the nearest feature where that is unmapped and mergeable, and STOPS at the first feature
// that already carries a map point (:789-790) -- including one an earlier point has just taken.  The only coupling between the walks
// is that: a feature belongs to the FIRST walk step that reaches it unmapped and can take it.  With step order
// ((first camera in which the point has a feature) x P + point) x C + camera, owner[feature] = the smallest order among the steps
// that (a) are reached -- no earlier camera of the same walk met a feature mapped on arrival: mapped before the pass, or owned by a
// step of smaller order -- and (b) may attach.  That is a recursion along a total order, so it has ONE solution, the sequential
// result; Jacobi sweeps (every walk re-evaluated against the previous sweep's owners) reach it in as many sweeps as the longest
// chain of walks cutting each other short -- two or three here.  One workgroup, a thread per point (its <= 16 candidate features and
// their flags in registers), sweeps separated by workgroup barriers; the owners live in a caller-supplied scratch (L2-resident).
// A point's later visits (it appears once per camera in which it has a feature) find what its first visit left and change nothing.
constexpr int RD_MAX_CAMS = 16;
struct RdArgs {
    int nCams, N, P, mapBase, nSweeps, onlyCam, kinds;
    const int* slot;                 // [P][nCams] the search's candidates
    const int* flags;                // [P][nCams] bit 1: the candidate is dynamic
    const unsigned char* mergeable;  // [P][nCams] 1: mergeable over the whole track
    const unsigned char* mapFlags;   // [P] CS_MAP_* of the pass's points
    int* pointFeat;                  // [P][nCams] in / out
    int* slot2map[RD_MAX_CAMS];      // [N] per camera, in / out
    unsigned char* attached;         // [P][nCams] out
    unsigned char* regged;           // [P] out: refineMapPoint is due (:889-893)
    int* code;                       // scratch [nCams][P]: the walk's view of (point, camera), see RD_* below
    int* base;                       // scratch [P]: order of the point's walk (x nCams), -1: the point is not visited
    int* owner[3];                   // scratch 3 x [nCams * N]: the sweeps rotate through them
    int* counts;                     // [4] out: features attached, points regged, sweeps, converged
    int* unconverged;                // the scratch's last word: the NUMBER of calls whose sweeps did not settle (never cleared here)
    int* timeouts;                   // the word before it: the number of those calls that ended on a grid-barrier TIME-OUT (the launch's workgroups
                                     // were not co-resident within 20 ms: nothing was attached) -- apart from `sweeps ran out`
    int* callFlag;                   // the word before it: this call has been counted
    int* changed;                    // scratch [RD_MAX_SWEEPS]: sweep k changed an owner (the self-settling launch, k_decide_settle)
    int* bar;                        // scratch [1]: its grid barrier's arrival counter
    // the second visits' first list built by the walks themselves (cs_register_decide_kinds_rounds_dev; null: not asked for): a point that
    // registered and holds a feature in a later camera's loop appends itself -- what k_revisit_list(firstRound) would find
    int* rvLists;                    // [nRounds][rvCap]: cleared to -1 by the prepare launch, list 0 filled by the settle launch
    int* rvCounts;                   // [nRounds]: cleared; [0] = points appended (may exceed rvCap: the rest is not visited again)
    int rvCap, nRounds;
    int* visitLoop;                  // [P]: the loop of a registered point's visit
    int* nextLoop;                   // [P]: the loop of its next visit
};
constexpr int RD_MAX_SWEEPS = 64;
// code of (point, camera): -1 the walk passes the camera by; else the candidate feature camera * N + slot in the low bits and
constexpr int RD_INIT_MAPPED = 1 << 29, RD_CAN_MERGE = 1 << 28, RD_FEAT = (1 << 28) - 1;
constexpr int RD_INF = 0x7fffffff;

// entry-parallel (coalesced over the P x nCams tables): what every walk will see of (point, camera); the owners start at "nobody"
__global__ __launch_bounds__(256) void k_decide_prepare(RdArgs A) {
    const int k = blockIdx.x * 256 + threadIdx.x, C = A.nCams, nFeat = C * A.N;
    for (int f = k; f < nFeat; f += gridDim.x * 256) A.owner[0][f] = RD_INF, A.owner[1][f] = RD_INF, A.owner[2][f] = RD_INF;
    if (k < 4 && A.counts) A.counts[k] = k == 2 ? A.nSweeps : (k == 3 ? 1 : 0);   // (converged: cleared by the last launch when not)
    if (k == 0) *A.callFlag = 0, *A.bar = 0;
    if (k < RD_MAX_SWEEPS) A.changed[k] = 0;
    if (A.rvLists) {
        for (int f = k; f < A.nRounds * A.rvCap; f += gridDim.x * 256) A.rvLists[f] = -1;
        if (k < A.nRounds) A.rvCounts[k] = 0;
    }
    if (k >= A.P * C) return;
    const int p = k / C, i = k - p * C;
    A.attached[k] = 0;
    if (i == 0) A.regged[p] = 0;
    int code = -1;
    // the point's kind: 0 certainly static, 1 certainly dynamic (its walk takes DYNAMIC features only, :981), -1 not visited
    const unsigned char fl = A.mapFlags[p] & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN);
    const int kind = (fl == 0 && (A.kinds & 1)) ? 0 : ((fl == CS_MAP_DYNAMIC && (A.kinds & 2)) ? 1 : -1);
    if (A.pointFeat[k] < 0 && kind >= 0) {   // (else :736-737: the point has a feature of this frame there)
        const int s = A.slot[k];
        if (s >= 0 && s < A.N && ((A.flags[k] >> 1) & 1) == kind) {   // (else: nothing found / a feature of the other type, :757 / :981)
            code = i * A.N + s;
            if (A.slot2map[i][s] >= 0) code |= RD_INIT_MAPPED;
            if (A.mergeable[k] == 1) code |= RD_CAN_MERGE;
        }
    }
    A.code[(size_t)i * A.P + p] = code;
    if (i == 0) {   // the order of the point's walk = ((first camera in which it has a feature of this frame) x P + point) x nCams
        int ofirst = -1;
        for (int q = C - 1; q >= 0; --q)
            if (A.pointFeat[(size_t)p * C + q] >= 0) ofirst = q;
        // onlyCam >= 0: ONE camera's loop of the reference (:864-869): the points with a feature of this frame in that camera, map order
        if (A.onlyCam >= 0) ofirst = A.pointFeat[(size_t)p * C + A.onlyCam] >= 0 ? 0 : -1;
        A.base[p] = (ofirst >= 0 && kind >= 0) ? (ofirst * A.P + p) * C : -1;
    }
}
// one Jacobi sweep, thread per point: its walk against the owners of the previous sweep (prev), claims into next; `clear` is the
// buffer the FOLLOWING sweep will claim into.  mode 1: the owners in prev are final -- attach instead of claiming.
__global__ __launch_bounds__(256) void k_decide_sweep(RdArgs A, const int* __restrict__ prev, int* __restrict__ next, int* __restrict__ clear,
                                                      const int* __restrict__ prev2, int mode) {
    const int p = blockIdx.x * 256 + threadIdx.x, C = A.nCams, nFeat = C * A.N;
    if (clear)
        for (int f = p; f < nFeat; f += gridDim.x * 256) clear[f] = RD_INF;
    if (mode == 1 && prev2) {   // converged: the last two sweeps agree on every owner
        int ch = 0;
        for (int f = p; f < nFeat; f += gridDim.x * 256) ch |= prev[f] != prev2[f];
        if (ch) {
            if (A.counts) A.counts[3] = 0;
            if (atomicExch(A.callFlag, 1) == 0) atomicAdd(A.unconverged, 1);
        }
    }
    if (p >= A.P) return;
    const int base = A.base[p];
    if (base < 0) return;
    int code[RD_MAX_CAMS], own[RD_MAX_CAMS];
#pragma unroll
    for (int i = 0; i < RD_MAX_CAMS; ++i) code[i] = i < C ? A.code[(size_t)i * A.P + p] : -1;
#pragma unroll
    for (int i = 0; i < RD_MAX_CAMS; ++i) own[i] = code[i] >= 0 ? prev[code[i] & RD_FEAT] : RD_INF;   // (independent loads)
    bool go = true, reg = false;
    int nAtt = 0;
#pragma unroll
    for (int i = 0; i < RD_MAX_CAMS; ++i) {
        if (go && code[i] >= 0) {
            const int ord = base + i, f = code[i] & RD_FEAT;
            if ((code[i] & RD_INIT_MAPPED) || own[i] < ord) {
                go = false;                                    // mapped on arrival: the walk ends (:789-790)
            } else if (code[i] & RD_CAN_MERGE) {
                if (mode == 0) {
                    atomicMin(&next[f], ord);                  // it takes the feature unless an earlier step does
                } else if (own[i] == ord) {
                    const int s = f - i * A.N;
                    A.slot2map[i][s] = A.mapBase + p;          // the feature and its predecessors on the track (:771-775): slot2map is per track
                    A.pointFeat[(size_t)p * C + i] = s;         // MapPoint::addFeature
                    A.attached[(size_t)p * C + i] = 1;
                    reg = true, ++nAtt;
                }
            }
        }
    }
    if (mode == 1 && reg) {
        A.regged[p] = 1;
        if (A.counts) atomicAdd(A.counts, nAtt), atomicAdd(A.counts + 1, 1);
    }
}

// ---- the sweeps as ONE launch that stops when they have settled ------------------------------------------------------------------------
// nSweeps launches with a fixed count either waste launches or, on a frame whose walks cut each other short in a longer chain than
// the count allows, end on an answer that is not the sequential one (counted, not repaired -- ADVICE r04).  Here every workgroup of
// ONE launch sweeps until a sweep changes no owner: sweep, grid barrier, compare the sweep's owners with the previous sweep's, grid
// barrier, stop or go on -- as many sweeps as the frame needs (two on a quiet frame), never more than RD_MAX_SWEEPS, then the
// attach.  The grid is P / 256 workgroups of 44 registers and no LDS: co-resident beside anything (the barrier spins on an arrival
// counter in the scratch; a wait of more than 20 ms -- a grid that is NOT co-resident -- gives up and counts the call as unsettled).
// Owners are read and cleared with agent-scope accesses (they are claimed by atomicMin in L2; a cached line of an earlier sweep must
// not be read back).
__device__ __forceinline__ bool rd_grid_barrier(int* ctr, int target) {
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(ctr, 1);
        const long long t0 = wall_clock64();
        int good = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 2000000LL) {
                good = 0;
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}
__device__ __forceinline__ int rd_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rd_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void k_decide_settle(RdArgs A) {
    const int p = blockIdx.x * 256 + threadIdx.x, C = A.nCams, nFeat = C * A.N, nB = gridDim.x, stride = nB * 256;
    const int base = p < A.P ? A.base[p] : -1;
    int code[RD_MAX_CAMS];
#pragma unroll
    for (int i = 0; i < RD_MAX_CAMS; ++i) code[i] = (base >= 0 && i < C) ? A.code[(size_t)i * A.P + p] : -1;
    int arrivals = 0, k = 0;
    bool settled = false, alive = true;
    const int* fin = A.owner[0];
    for (; k < RD_MAX_SWEEPS; ++k) {
        const int* prev = A.owner[k % 3];
        int* next = A.owner[(k + 1) % 3];
        int* clear = A.owner[(k + 2) % 3];
        for (int f = p; f < nFeat; f += stride) rd_st(clear + f, RD_INF);
        if (base >= 0) {
            bool go = true;
#pragma unroll
            for (int i = 0; i < RD_MAX_CAMS; ++i) {
                if (go && code[i] >= 0) {
                    const int ord = base + i, f = code[i] & RD_FEAT;
                    if ((code[i] & RD_INIT_MAPPED) || rd_ld(prev + f) < ord) {
                        go = false;
                    } else if (code[i] & RD_CAN_MERGE) {
                        atomicMin(&next[f], ord);
                    }
                }
            }
        }
        arrivals += nB;
        if (!rd_grid_barrier(A.bar, arrivals)) {
            alive = false;
            break;
        }
        int ch = 0;
        for (int f = p; f < nFeat; f += stride) ch |= rd_ld(next + f) != rd_ld(prev + f);
        if (__syncthreads_or(ch) && threadIdx.x == 0) atomicOr(A.changed + k, 1);
        arrivals += nB;
        if (!rd_grid_barrier(A.bar, arrivals)) {
            alive = false;
            break;
        }
        fin = next;
        if (rd_ld(A.changed + k) == 0) {
            settled = true;
            ++k;
            break;
        }
    }
    if (p == 0 && A.counts) A.counts[2] = k, A.counts[3] = settled ? 1 : 0;
    if (!settled && p == 0) atomicAdd(A.unconverged, 1);
    if (!alive && p == 0) atomicAdd(A.timeouts, 1);
    if (!alive || base < 0) return;
    // the owners in `fin` are final (or the best the sweeps reached): attach (k_decide_sweep's mode 1)
    bool go = true, reg = false;
    int nAtt = 0;
#pragma unroll
    for (int i = 0; i < RD_MAX_CAMS; ++i) {
        if (go && code[i] >= 0) {
            const int ord = base + i, f = code[i] & RD_FEAT, own = rd_ld(fin + f);
            if ((code[i] & RD_INIT_MAPPED) || own < ord) {
                go = false;
            } else if ((code[i] & RD_CAN_MERGE) && own == ord) {
                const int s2 = f - i * A.N;
                A.slot2map[i][s2] = A.mapBase + p;
                A.pointFeat[(size_t)p * C + i] = s2;
                A.attached[(size_t)p * C + i] = 1;
                reg = true, ++nAtt;
            }
        }
    }
    if (reg) {
        A.regged[p] = 1;
        if (A.counts) atomicAdd(A.counts, nAtt), atomicAdd(A.counts + 1, 1);
        if (A.rvLists && A.onlyCam < 0) {   // k_revisit_list's first round, by the walk itself
            const int last = (base / C) / A.P;   // the loop of this visit: the first camera that held a feature before the pass (k_decide_prepare)
            A.visitLoop[p] = last;
            int b = -1;
            for (int c = C - 1; c > last; --c)
                if (A.pointFeat[(size_t)p * C + c] >= 0) b = c;
            if (b >= 0) {
                const int q = atomicAdd(A.rvCounts, 1);
                if (q < A.rvCap) A.rvLists[q] = p, A.nextLoop[p] = b;
                else atomicAdd(A.rvCounts + A.nRounds, 1);   // (never cleared here: the points beyond the lists over the run)
            }
        }
    }
}

// ---- the SECOND VISITS of the reference's order (round 6) -----------------------------------------------------------------------------
// CoSLAM::curStaticPointsRegInGroup builds every camera's visiting list afresh (src/app/SL_CoSLAM.cpp:864-869) and refines the points that
// gained a feature at the end of every camera's loop (:889-893): a point that registers in camera a's loop is visited AGAIN in the loop of
// every later camera b in which it holds a feature of this frame -- the one it has just gained included -- projected with its refined
// position.  A later visit of a point that did NOT register changes nothing (same position, same candidates, and a candidate can only have
// gone from unmapped to mapped since, which ends a walk earlier): the single pass above is the reference's run up to the second visits of
// the points it registered.  Those are played here, in rounds: round r visits the points that registered in round r - 1 (round 0 = the
// single pass) in their next loop -- behind a search and a mergability pass over JUST those points at their refined positions -- and a
// refine of the ones that registered again follows.  One workgroup, a thread per listed point; the visits of a round are ordered among
// themselves like all walks ((loop x P + point) x C + camera, Jacobi sweeps over the features they compete for).
// What a round cannot do is take back what a LATER-ordered visit of the single pass did with a feature this visit would have reached
// first: such a frame is counted (conflicts) and left as the single pass decided.  tools/r06_exact_vs_single.py measures what remains.
struct RvListArgs {
    int nCams, P, cap, firstRound;
    const int* pointFeat;              // [P][nCams]
    const unsigned char* attached;     // [P][nCams]: attached in THIS frame (the single pass's and the rounds' so far)
    unsigned char* regIn;              // [P]: registered in the previous round (read; cleared unless keepIn)
    int keepIn;
    unsigned char* regOutClear;        // [P] or null: the array this round's walks will mark -- cleared here (it may hold an earlier frame's marks)
    int* visitLoop;                    // [P]: the loop of the point's latest registering visit
    int* nextLoop;                     // [P]: the loop of the visit listed here
    int* list;                         // [cap] out, padded with -1
    int* counts;                       // [4]: listed, overflow (points beyond cap), -, -
};
__global__ __launch_bounds__(1024) void k_revisit_list(RvListArgs A) {
    __shared__ int sCount;
    const int tid = threadIdx.x, C = A.nCams;
    if (tid == 0) sCount = 0;
    __syncthreads();
    for (int p = tid; p < A.P; p += 1024) {
        if (A.regOutClear) A.regOutClear[p] = 0;
        if (!A.regIn[p]) continue;
        if (!A.keepIn) A.regIn[p] = 0;
        int last;
        if (A.firstRound) {   // the loop of its first visit: the first camera that held a feature of this frame BEFORE the pass attached any
            last = -1;
            for (int c = C - 1; c >= 0; --c)
                if (A.pointFeat[(size_t)p * C + c] >= 0 && !A.attached[(size_t)p * C + c]) last = c;
            A.visitLoop[p] = last;
        } else {
            last = A.visitLoop[p];
        }
        if (last < 0) continue;
        int b = -1;
        for (int c = C - 1; c > last; --c)
            if (A.pointFeat[(size_t)p * C + c] >= 0) b = c;
        if (b < 0) continue;   // no later loop holds the point: it is not visited again
        const int k = atomicAdd(&sCount, 1);
        if (k < A.cap) A.list[k] = p, A.nextLoop[p] = b;
    }
    __syncthreads();
    const int n = sCount < A.cap ? sCount : A.cap;
    for (int k = n + tid; k < A.cap; k += 1024) A.list[k] = -1;
    if (tid == 0 && A.counts) {
        A.counts[0] = n;
        if (sCount > A.cap) atomicAdd(A.counts + 1, sCount - A.cap);
    }
}

struct RvArgs {
    int nCams, N, P, cap, mapBase, kinds;
    const int* list;                 // [cap]: the points visited again (k_revisit_list)
    const int* nextLoop;             // [P]
    int* visitLoop;                  // [P]
    const int* slot;                 // the search's tables over the listed rows, [P][nCams]
    const int* flags;
    const unsigned char* mergeable;
    const unsigned char* mapFlags;
    int* pointFeat;
    int* slot2map[RD_MAX_CAMS];
    unsigned char* attached;         // [P][nCams]: set where this round attaches
    unsigned char* regOut;           // [P]: set for the points that registered in this round
    int* owner[3];                   // the decision's owner arrays (scratch): only the entries of this round's candidates are touched
    const int* curList;              // the frame's current points (cs_register_list_current_dev) and their count: who else wanted a feature
    const int* curCount;
    int curCap;
    int* counts;                     // [4] (accumulating): features attached, points registered, conflicts, sweeps that did not settle
    const int* listCount;            // null, or k_revisit_list's count: 0 = nobody is visited again (the usual round): leave at once
    int* nextList;                   // null, or the NEXT round's list [cap] (cleared to -1 before): a point that registered here and holds a
    int* nextCount;                  // feature in a later loop appends itself (what k_revisit_list would find), *nextCount counts them
    int* nextLoopW;                  // = nextLoop, written for the appended points
    int* overflow;                   // null, or where the points beyond the next list are counted
    int debug;                       // cs_debug_set("merge_print", 1): the launch prints where its time went
};
constexpr int RV_MAX_ROWS = 1024;
// MC: the cameras the row arrays are sized for (8 or MC: 16 cameras' worth of registers per thread spill at 1024 threads)
template <int MC>
__global__ __launch_bounds__(1024) void k_revisit_decide(RvArgs A) {
    __shared__ int sChanged, sAttCam[256], sAttSl[256], sAttKey[256], sNAtt;
    if (A.listCount && *A.listCount == 0) return;   // (uniform: before any barrier)
    const int j = threadIdx.x, C = A.nCams;
    const long long tD0 = A.debug ? wall_clock64() : 0;
    long long tD1 = 0, tD2 = 0, tD3 = 0;
    const int p = j < A.cap ? A.list[j] : -1;
    int code[MC], base = -1, kind = -1, nConf = 0;
#pragma unroll
    for (int i = 0; i < MC; ++i) code[i] = -1;
    if (j == 0) sNAtt = 0;
    // the conflict count's scan (at the end) compares every current point's candidates with what this round attached: the candidate rows do
    // not change in this launch, so a thread asks for its (up to two) current points' rows NOW -- the loads travel while the walks are built
    // and swept -- and the scan is register compares (it was 13 us of three dependent rounds of loads behind the attach, as much as the walks)
    const int nCurPre = *A.curCount < A.curCap ? *A.curCount : A.curCap;
    int preQ[2], preS[2][MC];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = j + u * (int)blockDim.x;
        preQ[u] = e < nCurPre ? A.curList[e] : -1;
        if (preQ[u] >= A.P) preQ[u] = -1;
#pragma unroll
        for (int i = 0; i < MC; ++i) preS[u][i] = preQ[u] >= 0 ? A.slot[(size_t)preQ[u] * C + (i < C ? i : 0)] : -1;
    }
    // the row's loads in three rounds (the point's flags, its loop and every camera's entry together; then who owns the candidates; then those
    // owners' state) instead of up to four dependent loads per camera one camera after the other: the launch is a handful of rows' latency
    int pfv[MC], slv[MC], flv[MC], own[MC];
    unsigned char mgv[MC];
    {
        const bool pv = p >= 0 && p < A.P;
        const size_t pr = pv ? (size_t)p : 0;
        const unsigned char fl = A.mapFlags[pr] & (CS_MAP_DYNAMIC | CS_MAP_FALSE | CS_MAP_UNCERTAIN);
        const int nl = A.nextLoop[pr];
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const size_t k = pr * C + (i < C ? i : 0);
            pfv[i] = A.pointFeat[k], slv[i] = A.slot[k], flv[i] = A.flags[k], mgv[i] = A.mergeable[k];
        }
        if (pv) {
            kind = (fl == 0 && (A.kinds & 1)) ? 0 : ((fl == CS_MAP_DYNAMIC && (A.kinds & 2)) ? 1 : -1);
            if (kind >= 0) base = (nl * A.P + p) * C;
        }
    }
    if (base >= 0) {
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const bool cand = i < C && pfv[i] < 0 && slv[i] >= 0 && slv[i] < A.N && ((flv[i] >> 1) & 1) == kind;   // :736-737; nothing found / the other type
            if (!cand) slv[i] = -1;
            own[i] = cand ? A.slot2map[i][slv[i]] - A.mapBase : -1;
        }
        int oAtt[MC], oPf[MC], oLoop[MC];
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const bool look = slv[i] >= 0 && own[i] >= 0 && own[i] < A.P;
            const size_t ko = (size_t)(look ? own[i] : 0) * C + (i < C ? i : 0);
            oAtt[i] = look ? A.attached[ko] : 0, oPf[i] = look ? A.pointFeat[ko] : -1, oLoop[i] = look ? A.visitLoop[look ? own[i] : 0] : 0;
        }
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            if (slv[i] < 0) continue;
            const int sl = slv[i];
            int c = i * A.N + sl;
            if (own[i] >= 0) {
                c |= RD_INIT_MAPPED;
                // mapped NOW.  Was it mapped when this visit takes place?  Not if a later-ordered visit of this frame attached it.
                if (own[i] < A.P && oAtt[i] && oPf[i] == sl && (oLoop[i] * A.P + own[i]) * C + i > base + i) ++nConf;
            } else if (mgv[i] == 1) {
                c |= RD_CAN_MERGE;
            }
            code[i] = c;
        }
#pragma unroll
        for (int i = 0; i < MC; ++i)
            if (code[i] >= 0) rd_st(A.owner[0] + (code[i] & RD_FEAT), RD_INF), rd_st(A.owner[1] + (code[i] & RD_FEAT), RD_INF), rd_st(A.owner[2] + (code[i] & RD_FEAT), RD_INF);
    }
    __syncthreads();   // (ONE workgroup: its barrier orders the agent-scope accesses above -- an agent-scope fence here writes the XCD's L2 back, tracker's lines and all: 30-70 us a launch)
    if (A.debug) tD1 = wall_clock64();
    // Jacobi sweeps among the round's visits (k_decide_settle's recursion, one workgroup)
    int k = 0;
    const int* fin = A.owner[0];
    bool settled = false;
    for (; k < 32; ++k) {
        const int* prev = A.owner[k % 3];
        int* next = A.owner[(k + 1) % 3];
        int* clear = A.owner[(k + 2) % 3];
        if (j == 0) sChanged = 0;
        if (base >= 0) {
#pragma unroll
            for (int i = 0; i < MC; ++i)
                if (code[i] >= 0) rd_st(clear + (code[i] & RD_FEAT), RD_INF);
        }
        __syncthreads();
        if (base >= 0) {
            bool go = true;
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                if (go && code[i] >= 0) {
                    const int ord = base + i, f = code[i] & RD_FEAT;
                    if ((code[i] & RD_INIT_MAPPED) || rd_ld(prev + f) < ord) go = false;
                    else if (code[i] & RD_CAN_MERGE) atomicMin(&next[f], ord);
                }
            }
        }
        __syncthreads();
        if (base >= 0) {
            int ch = 0;
#pragma unroll
            for (int i = 0; i < MC; ++i)
                if (code[i] >= 0) ch |= rd_ld(next + (code[i] & RD_FEAT)) != rd_ld(prev + (code[i] & RD_FEAT));
            if (ch) sChanged = 1;
        }
        __syncthreads();
        fin = next;
        const int chg = sChanged;
        __syncthreads();
        if (!chg) {
            settled = true;
            break;
        }
    }
    if (A.debug) tD2 = wall_clock64();
    // attach (the owners in `fin` are final)
    bool reg = false;
    int nAtt = 0;
    if (base >= 0) {
        bool go = true;
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            if (go && code[i] >= 0) {
                const int ord = base + i, f = code[i] & RD_FEAT, own = rd_ld(fin + f);
                if ((code[i] & RD_INIT_MAPPED) || own < ord) {
                    go = false;
                } else if ((code[i] & RD_CAN_MERGE) && own == ord) {
                    const int s2 = f - i * A.N;
                    A.slot2map[i][s2] = A.mapBase + p;
                    A.pointFeat[(size_t)p * C + i] = s2;
                    A.attached[(size_t)p * C + i] = 1;
                    reg = true, ++nAtt;
                    const int q = atomicAdd(&sNAtt, 1);
                    if (q < 256) sAttCam[q] = i, sAttSl[q] = s2, sAttKey[q] = ord;
                }
            }
        }
        if (reg) {
            const int last = A.nextLoop[p];
            A.regOut[p] = 1, A.visitLoop[p] = last;
            if (A.nextList) {
                int b = -1;
                for (int c = C - 1; c > last; --c)
                    if (A.pointFeat[(size_t)p * C + c] >= 0) b = c;
                if (b >= 0) {
                    const int q = atomicAdd(A.nextCount, 1);
                    if (q < A.cap) A.nextList[q] = p, A.nextLoopW[p] = b;
                    else if (A.overflow) atomicAdd(A.overflow, 1);
                }
            }
        }
    }
    __syncthreads();
    // a feature attached here was unmapped until now: a LATER-ordered visit of this frame that had it as its candidate walked past it (it
    // could not take it) and went on to other cameras -- in the reference's order that walk ends at it.  Counted where that walk attached
    // something behind it (what it did there would not have happened).
    if (A.debug) tD3 = wall_clock64();
    const int nA = sNAtt < 256 ? sNAtt : 256;
    if (nA > 0) {
        const int nCur = nCurPre;
        for (int e = j, u = 0; e < nCur; e += (int)blockDim.x, ++u) {
            const int q = u < 2 ? preQ[u < 2 ? u : 0] : A.curList[e];
            if (q < 0 || q >= A.P) continue;
            int qs[MC];
#pragma unroll
            for (int i = 0; i < MC; ++i) qs[i] = u == 0 ? preS[0][i] : (u == 1 ? preS[1][i] : A.slot[(size_t)q * C + (i < C ? i : 0)]);
            // does ANY of the round's attachments name one of q's candidates?  (almost never: only then is q's row of features looked at)
            bool any = false;
            for (int a = 0; a < nA && !any; ++a) {
                const int i = sAttCam[a], sl = sAttSl[a];
#pragma unroll
                for (int t = 0; t < MC; ++t) any |= t == i && qs[t] == sl;
            }
            if (!any) continue;
            int qp[MC];
            unsigned qatt = 0;
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const size_t kq = (size_t)q * C + (i < C ? i : 0);
                qp[i] = A.pointFeat[kq];
                if (i < C && A.attached[kq]) qatt |= 1u << i;
            }
            int lq = -1;   // the loop of q's (first) visit in this frame
#pragma unroll
            for (int i = MC - 1; i >= 0; --i)
                if (i < C && qp[i] >= 0 && !((qatt >> i) & 1u)) lq = i;
            if (lq < 0) continue;
            for (int a = 0; a < nA; ++a) {
                const int i = sAttCam[a], sl = sAttSl[a];
                int qsi = -1, qpi = 0;
#pragma unroll
                for (int t = 0; t < MC; ++t)
                    if (t == i) qsi = qs[t], qpi = qp[t];
                if (qsi != sl || qpi >= 0) continue;
                if ((lq * A.P + q) * C + i <= sAttKey[a]) continue;
                if ((qatt >> (i + 1)) != 0u) ++nConf;   // it attached something in a camera behind the one it walked past
            }
        }
    }
    if (A.debug && j == 0)
        printf("k_revisit_decide: rows %d; build %lld us, %d sweeps %lld us, attach %lld us (%d attached), scan %lld us\n", A.listCount ? *A.listCount : -1,
               (tD1 - tD0) / 100, k + 1, (tD2 - tD1) / 100, (tD3 - tD2) / 100, nA, (wall_clock64() - tD3) / 100);
    if (A.counts) {
        if (nAtt) atomicAdd(A.counts, nAtt);
        if (reg) atomicAdd(A.counts + 1, 1);
        if (nConf) atomicAdd(A.counts + 2, nConf);
        if (j == 0 && !settled) atomicAdd(A.counts + 3, 1);
    }
}

}  // namespace

extern "C" int cs_register_revisit_list_dev(int device, void* hip_stream, int nCams, int P, int cap, int firstRound, const int* d_pointFeat,
                                            const unsigned char* d_attached, unsigned char* d_regIn, int keepIn, unsigned char* d_regOutClear,
                                            int* d_visitLoop, int* d_nextLoop, int* d_list, int* d_counts) {
    if (nCams < 1 || nCams > RD_MAX_CAMS || P < 1 || cap < 1 || cap > RV_MAX_ROWS || !d_pointFeat || !d_attached || !d_regIn || d_regOutClear == d_regIn || !d_visitLoop || !d_nextLoop ||
        !d_list) {
        cs_set_error("cs_register_revisit_list_dev: bad arguments (1..%d cameras, 1..%d rows)", RD_MAX_CAMS, RV_MAX_ROWS);
        return CS_ERR_INVALID;
    }
    RvListArgs A;
    A.nCams = nCams, A.P = P, A.cap = cap, A.firstRound = firstRound ? 1 : 0, A.pointFeat = d_pointFeat, A.attached = d_attached;
    A.regIn = d_regIn, A.keepIn = keepIn ? 1 : 0, A.regOutClear = d_regOutClear, A.visitLoop = d_visitLoop, A.nextLoop = d_nextLoop, A.list = d_list, A.counts = d_counts;
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_revisit_list, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, A);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_register_revisit_decide_dev(int device, void* hip_stream, int nCams, int N, int P, int cap, int mapBase, int kinds, const int* d_list,
                                              const int* d_nextLoop, int* d_visitLoop, const int* d_slot, const int* d_flags,
                                              const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat,
                                              int* const* d_slot2map, unsigned char* d_attached, unsigned char* d_regOut, void* d_decideScratch,
                                              const int* d_curList, const int* d_curCount, int curCap, int* d_counts, const int* d_listCount) {
    return cs_register_revisit_decide_next_dev(device, hip_stream, nCams, N, P, cap, mapBase, kinds, d_list, (int*)d_nextLoop, d_visitLoop, d_slot, d_flags,
                                               d_mergeable, d_mapFlags, d_pointFeat, d_slot2map, d_attached, d_regOut, d_decideScratch, d_curList, d_curCount,
                                               curCap, d_counts, d_listCount, nullptr, nullptr, nullptr);
}

extern "C" int cs_register_revisit_decide_next_dev(int device, void* hip_stream, int nCams, int N, int P, int cap, int mapBase, int kinds, const int* d_list,
                                                   int* d_nextLoop, int* d_visitLoop, const int* d_slot, const int* d_flags,
                                                   const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat,
                                                   int* const* d_slot2map, unsigned char* d_attached, unsigned char* d_regOut, void* d_decideScratch,
                                                   const int* d_curList, const int* d_curCount, int curCap, int* d_counts, const int* d_listCount,
                                                   int* d_nextList, int* d_nextCount, int* d_overflow) {
    if ((d_nextList != nullptr) != (d_nextCount != nullptr)) {
        cs_set_error("cs_register_revisit_decide_next_dev: the next round's list and its count go together");
        return CS_ERR_INVALID;
    }
    if (nCams < 1 || nCams > RD_MAX_CAMS || N < 1 || P < 1 || cap < 1 || cap > RV_MAX_ROWS || kinds < 1 || kinds > 3 || !d_list || !d_nextLoop || !d_visitLoop ||
        !d_slot || !d_flags || !d_mergeable || !d_mapFlags || !d_pointFeat || !d_slot2map || !d_attached || !d_regOut || !d_decideScratch ||
        !d_curList || !d_curCount || (long long)nCams * N > RD_FEAT || (long long)nCams * P * nCams > 0x7fffffffLL) {
        cs_set_error("cs_register_revisit_decide_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    RvArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = nCams, A.N = N, A.P = P, A.cap = cap, A.mapBase = mapBase, A.kinds = kinds, A.list = d_list, A.nextLoop = d_nextLoop, A.visitLoop = d_visitLoop;
    A.slot = d_slot, A.flags = d_flags, A.mergeable = d_mergeable, A.mapFlags = d_mapFlags, A.pointFeat = d_pointFeat;
    for (int c = 0; c < nCams; ++c) A.slot2map[c] = d_slot2map[c];
    A.attached = d_attached, A.regOut = d_regOut, A.curList = d_curList, A.curCount = d_curCount, A.curCap = curCap, A.counts = d_counts, A.listCount = d_listCount;
    A.nextList = d_nextList, A.nextCount = d_nextCount, A.nextLoopW = d_nextLoop, A.overflow = d_overflow;
    A.debug = cs_debug_get(CS_DBG_MERGE_PRINT) == 1;
    int* scr = (int*)d_decideScratch + (size_t)nCams * P + P;   // (cs_register_decide_kinds_dev's carve-up: code | base | owner x 3 | ...)
    for (int k = 0; k < 3; ++k) A.owner[k] = scr, scr += (size_t)nCams * N;
    CS_HIP(hipSetDevice(device));
    if (nCams <= 8)
        hipLaunchKernelGGL(k_revisit_decide<8>, dim3(1), dim3(cap <= 256 ? 256 : (cap + 63) / 64 * 64), 0, (hipStream_t)hip_stream, A);
    else
        hipLaunchKernelGGL(k_revisit_decide<RD_MAX_CAMS>, dim3(1), dim3(cap <= 256 ? 256 : (cap + 63) / 64 * 64), 0, (hipStream_t)hip_stream, A);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
extern "C" size_t cs_register_decide_scratch_bytes(int nCams, int N, int P) {
    if (nCams < 1 || N < 1 || P < 0) return 0;
    return sizeof(int) * ((size_t)nCams * P + (size_t)P + 3 * (size_t)nCams * N + RD_MAX_SWEEPS + 4);
}

extern "C" int cs_register_decide_static_dev(int device, void* hip_stream, int nCams, int N, int P, int mapBase, const int* d_slot, const int* d_flags,
                                             const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat,
                                             int* const* d_slot2map /* host array of nCams device pointers */, unsigned char* d_attached,
                                             unsigned char* d_regged, void* d_scratch, int nSweeps, int* d_counts) {
    return cs_register_decide_static_cam_dev(device, hip_stream, nCams, N, P, mapBase, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, d_slot2map,
                                             d_attached, d_regged, d_scratch, nSweeps, d_counts, -1);
}

extern "C" int cs_register_decide_static_cam_dev(int device, void* hip_stream, int nCams, int N, int P, int mapBase, const int* d_slot,
                                                 const int* d_flags, const unsigned char* d_mergeable, const unsigned char* d_mapFlags,
                                                 int* d_pointFeat, int* const* d_slot2map, unsigned char* d_attached, unsigned char* d_regged,
                                                 void* d_scratch, int nSweeps, int* d_counts, int onlyCam) {
    return cs_register_decide_kinds_dev(device, hip_stream, nCams, N, P, mapBase, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, d_slot2map,
                                        d_attached, d_regged, d_scratch, nSweeps, d_counts, onlyCam, 1);
}

extern "C" int cs_register_decide_kinds_dev(int device, void* hip_stream, int nCams, int N, int P, int mapBase, const int* d_slot, const int* d_flags,
                                            const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat,
                                            int* const* d_slot2map, unsigned char* d_attached, unsigned char* d_regged, void* d_scratch,
                                            int nSweeps, int* d_counts, int onlyCam, int kinds) {
    return cs_register_decide_kinds_rounds_dev(device, hip_stream, nCams, N, P, mapBase, d_slot, d_flags, d_mergeable, d_mapFlags, d_pointFeat, d_slot2map,
                                               d_attached, d_regged, d_scratch, nSweeps, d_counts, onlyCam, kinds, nullptr, 0, 0, nullptr, nullptr, nullptr);
}

extern "C" int cs_register_decide_kinds_rounds_dev(int device, void* hip_stream, int nCams, int N, int P, int mapBase, const int* d_slot, const int* d_flags,
                                                   const unsigned char* d_mergeable, const unsigned char* d_mapFlags, int* d_pointFeat,
                                                   int* const* d_slot2map, unsigned char* d_attached, unsigned char* d_regged, void* d_scratch,
                                                   int nSweeps, int* d_counts, int onlyCam, int kinds, int* d_rvLists, int rvCap, int nRounds,
                                                   int* d_rvCounts, int* d_visitLoop, int* d_nextLoop) {
    if (d_rvLists && (rvCap < 1 || rvCap > RV_MAX_ROWS || nRounds < 1 || nRounds > 8 || !d_rvCounts || !d_visitLoop || !d_nextLoop || nSweeps != 0 || onlyCam >= 0)) {
        cs_set_error("cs_register_decide_kinds_rounds_dev: the second visits' lists need 1..%d rows, 1..8 rounds, their counters, visitLoop / nextLoop, "
                     "the self-settling launch (nSweeps 0) and all cameras' loops", RV_MAX_ROWS);
        return CS_ERR_INVALID;
    }
    if (onlyCam >= nCams || kinds < 1 || kinds > 3) {
        cs_set_error("cs_register_decide_kinds_dev: camera %d of %d, kinds %d (1: static, 2: dynamic, 3: both)", onlyCam, nCams, kinds);
        return CS_ERR_INVALID;
    }
    if (nCams < 1 || nCams > RD_MAX_CAMS || N < 1 || P < 0 || (long long)nCams * N > RD_FEAT || (long long)nCams * P * nCams > 0x7fffffffLL ||
        mapBase < 0 || nSweeps < 0 || nSweeps > 256 || !d_slot || !d_flags || !d_mergeable || !d_mapFlags || !d_pointFeat || !d_slot2map ||
        !d_attached || !d_regged || !d_scratch) {
        cs_set_error("cs_register_decide_static_dev: bad arguments (1..%d cameras, 0..256 sweeps)", RD_MAX_CAMS);
        return CS_ERR_INVALID;
    }
    RdArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = nCams, A.N = N, A.P = P, A.mapBase = mapBase, A.nSweeps = nSweeps, A.onlyCam = onlyCam < 0 ? -1 : onlyCam, A.kinds = kinds;
    A.slot = d_slot, A.flags = d_flags, A.mergeable = d_mergeable, A.mapFlags = d_mapFlags, A.pointFeat = d_pointFeat;
    for (int c = 0; c < nCams; ++c) {
        if (!d_slot2map[c]) {
            cs_set_error("cs_register_decide_static_dev: null slot2map of camera %d", c);
            return CS_ERR_INVALID;
        }
        A.slot2map[c] = d_slot2map[c];
    }
    A.attached = d_attached, A.regged = d_regged, A.counts = d_counts;
    A.rvLists = d_rvLists, A.rvCap = rvCap, A.nRounds = nRounds, A.rvCounts = d_rvCounts, A.visitLoop = d_visitLoop, A.nextLoop = d_nextLoop;
    int* scr = (int*)d_scratch;
    A.code = scr, scr += (size_t)nCams * P;
    A.base = scr, scr += P;
    for (int k = 0; k < 3; ++k) A.owner[k] = scr, scr += (size_t)nCams * N;
    A.changed = scr, scr += RD_MAX_SWEEPS;
    A.bar = scr, A.callFlag = scr + 1, A.timeouts = scr + 2, A.unconverged = scr + 3;
    CS_HIP(hipSetDevice(device));
    if (P == 0) return CS_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    const int gE = (P * nCams + 255) / 256, gP = (P + 255) / 256;
    hipLaunchKernelGGL(k_decide_prepare, dim3(gE), dim3(256), 0, s, A);
    if (nSweeps == 0) {   // as many sweeps as the frame needs, ONE launch (k_decide_settle)
        hipLaunchKernelGGL(k_decide_settle, dim3(gP), dim3(256), 0, s, A);
        CS_CHECK_LAUNCH();
        return CS_OK;
    }
    // sweep k reads owner[k % 3], claims into owner[(k + 1) % 3] and clears owner[(k + 2) % 3] for the sweep after it
    for (int k = 0; k < nSweeps; ++k)
        hipLaunchKernelGGL(k_decide_sweep, dim3(gP), dim3(256), 0, s, A, (const int*)A.owner[k % 3], A.owner[(k + 1) % 3], A.owner[(k + 2) % 3],
                           (const int*)nullptr, 0);
    hipLaunchKernelGGL(k_decide_sweep, dim3(gP), dim3(256), 0, s, A, (const int*)A.owner[nSweeps % 3], (int*)nullptr, (int*)nullptr,
                       (const int*)A.owner[(nSweeps + 2) % 3], 1);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

namespace {

// ---- the candidates of a rank's own cameras to every rank (cameras sharded over GPUs) ------------------------------------------------
// pack: columns cam0 .. cam0 + nOwn - 1 of the P x nCams tables slot / flags / mergeable into one send record of 3 * nOwn * P ints
// ([table][own camera][point]); unpack: the gathered records of all ranks (rank r owns cameras r * nOwn ..) back into the tables.
__global__ __launch_bounds__(256) void k_candidates_pack(int P, int nCams, int cam0, int nOwn, const int* __restrict__ slot, const int* __restrict__ flags,
                                                         const unsigned char* __restrict__ merg, int* __restrict__ send) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nOwn * P) return;
    const int i = q / P, p = q - i * P;
    const size_t k = (size_t)p * nCams + cam0 + i;
    send[q] = slot[k], send[nOwn * P + q] = flags[k], send[2 * nOwn * P + q] = merg[k];
}
__global__ __launch_bounds__(256) void k_candidates_unpack(int P, int nCams, int nOwn, int skipRank, const int* __restrict__ recv, int* __restrict__ slot,
                                                           int* __restrict__ flags, unsigned char* __restrict__ merg) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nCams * P) return;
    const int g = q / P, p = q - g * P, r = g / nOwn, i = g - r * nOwn;
    if (r == skipRank) return;
    const int* rec = recv + (size_t)r * 3 * nOwn * P;
    const size_t k = (size_t)p * nCams + g;
    slot[k] = rec[i * P + p], flags[k] = rec[nOwn * P + i * P + p], merg[k] = (unsigned char)rec[2 * nOwn * P + i * P + p];
}

}  // namespace

namespace {
// the same over a LIST of rows (cs_register_list_current_dev: identical on every rank, since every rank holds the same pointFeat table): row j of
// the record = map point list[j], j < cap; entries of rows beyond the list travel as -1 / 0 / 255.  The record stays cap rows long whatever
// the map's capacity is.
__global__ __launch_bounds__(256) void k_candidates_pack_list(int cap, int nCams, int cam0, int nOwn, const int* __restrict__ list,
                                                              const int* __restrict__ slot, const int* __restrict__ flags,
                                                              const unsigned char* __restrict__ merg, int* __restrict__ send) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nOwn * cap) return;
    const int i = q / cap, j = q - i * cap, p = list[j];
    const size_t k = (size_t)(p < 0 ? 0 : p) * nCams + cam0 + i;
    send[q] = p < 0 ? -1 : slot[k], send[nOwn * cap + q] = p < 0 ? 0 : flags[k], send[2 * nOwn * cap + q] = p < 0 ? 255 : merg[k];
}
__global__ __launch_bounds__(256) void k_candidates_unpack_list(int cap, int nCams, int nOwn, int skipRank, const int* __restrict__ list,
                                                                const int* __restrict__ recv, int* __restrict__ slot, int* __restrict__ flags,
                                                                unsigned char* __restrict__ merg) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nCams * cap) return;
    const int g = q / cap, j = q - g * cap, r = g / nOwn, i = g - r * nOwn, p = list[j];
    if (r == skipRank || p < 0) return;
    const int* rec = recv + (size_t)r * 3 * nOwn * cap;
    const size_t k = (size_t)p * nCams + g;
    slot[k] = rec[i * cap + j], flags[k] = rec[nOwn * cap + i * cap + j], merg[k] = (unsigned char)rec[2 * nOwn * cap + i * cap + j];
}
}  // namespace

extern "C" int cs_register_candidates_pack_list_dev(int device, void* hip_stream, int cap, int nCams, int cam0, int nOwn, const int* d_list,
                                                    const int* d_slot, const int* d_flags, const unsigned char* d_mergeable, int* d_send) {
    if (cap < 0 || nCams < 1 || cam0 < 0 || nOwn < 1 || cam0 + nOwn > nCams || !d_list || !d_slot || !d_flags || !d_mergeable || !d_send) {
        cs_set_error("cs_register_candidates_pack_list_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (cap == 0) return CS_OK;
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_candidates_pack_list, dim3((nOwn * cap + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, cap, nCams, cam0, nOwn, d_list,
                       d_slot, d_flags, d_mergeable, d_send);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
extern "C" int cs_register_candidates_unpack_list_dev(int device, void* hip_stream, int cap, int nCams, int nOwn, int skipRank, const int* d_list,
                                                      const int* d_recv, int* d_slot, int* d_flags, unsigned char* d_mergeable) {
    if (cap < 0 || nCams < 1 || nOwn < 1 || nCams % nOwn || !d_list || !d_recv || !d_slot || !d_flags || !d_mergeable) {
        cs_set_error("cs_register_candidates_unpack_list_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (cap == 0) return CS_OK;
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_candidates_unpack_list, dim3((nCams * cap + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, cap, nCams, nOwn, skipRank,
                       d_list, d_recv, d_slot, d_flags, d_mergeable);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_register_candidates_pack_dev(int device, void* hip_stream, int P, int nCams, int cam0, int nOwn, const int* d_slot,
                                               const int* d_flags, const unsigned char* d_mergeable, int* d_send) {
    if (P < 1 || nCams < 1 || cam0 < 0 || nOwn < 1 || cam0 + nOwn > nCams || !d_slot || !d_flags || !d_mergeable || !d_send) {
        cs_set_error("cs_register_candidates_pack_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_candidates_pack, dim3((nOwn * P + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, P, nCams, cam0, nOwn, d_slot, d_flags,
                       d_mergeable, d_send);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
extern "C" int cs_register_candidates_unpack_dev(int device, void* hip_stream, int P, int nCams, int nOwn, int skipRank, const int* d_recv, int* d_slot,
                                                 int* d_flags, unsigned char* d_mergeable) {
    if (P < 1 || nCams < 1 || nOwn < 1 || nCams % nOwn || !d_recv || !d_slot || !d_flags || !d_mergeable) {
        cs_set_error("cs_register_candidates_unpack_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_candidates_unpack, dim3((nCams * P + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, P, nCams, nOwn, skipRank, d_recv,
                       d_slot, d_flags, d_mergeable);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

namespace {
}  // namespace

// Host-pointer form for the reference's loops: one upload, one launch, one read-back.  cams[c] holds HOST pointers.
extern "C" int cs_register_search(int device, int nCams, const cs_register_cam* cams, int N, int W, int H, int P, const double* M,
                                  const double* cov, const int* pointFeat, double sigmaSearch, double maxDist, double sigmaMerge,
                                  int* slot, double* m, double* var, double* dist, int* flags) {
    int rc = check_args("cs_register_search", nCams, cams, N, W, H, P, sigmaSearch, maxDist, sigmaMerge);
    if (rc != CS_OK) return rc;
    if (P == 0) return CS_OK;
    if (!M || !cov || !pointFeat || !slot || !m || !var || !dist || !flags) {
        cs_set_error("cs_register_search: null pointer");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(device));
    const size_t pairs = (size_t)P * nCams;
    // one staging block: per camera K R t | xy | state | slot2map | isDynamic, then M | cov | pointFeat, then the outputs
    const size_t camBytes = (21 + 2 * (size_t)N) * 8 + 2 * (size_t)N * 4 + (((size_t)N + 7) & ~(size_t)7);
    const size_t inBytes = camBytes * nCams + (size_t)P * 12 * 8 + ((pairs * 4 + 7) & ~(size_t)7);
    const size_t outBytes = pairs * (2 + 4 + 1) * 8 + 2 * ((pairs * 4 + 7) & ~(size_t)7);
    char *h = nullptr, *d = nullptr;
    CS_HIP(hipHostMalloc((void**)&h, inBytes + outBytes, hipHostMallocDefault));
    if (hipMalloc((void**)&d, inBytes + outBytes) != hipSuccess) {
        (void)hipHostFree(h);
        cs_set_error("cs_register_search: out of device memory");
        return CS_ERR_HIP;
    }
    cs_register_cam dc[RG_MAX_CAMS];
    size_t off = 0;
    for (int c = 0; c < nCams; ++c) {
        const cs_register_cam& q = cams[c];
        if (!q.K || !q.R || !q.t || !q.xy || !q.state || !q.slot2map) {
            (void)hipHostFree(h);
            (void)hipFree(d);
            cs_set_error("cs_register_search: null pointer in camera %d", c);
            return CS_ERR_INVALID;
        }
        memcpy(h + off, q.K, 72);
        dc[c].K = (const double*)(d + off);
        memcpy(h + off + 72, q.R, 72);
        dc[c].R = (const double*)(d + off + 72);
        memcpy(h + off + 144, q.t, 24);
        dc[c].t = (const double*)(d + off + 144);
        size_t o2 = off + 168;
        memcpy(h + o2, q.xy, 2 * (size_t)N * 8);
        dc[c].xy = (const double*)(d + o2);
        o2 += 2 * (size_t)N * 8;
        memcpy(h + o2, q.state, (size_t)N * 4);
        dc[c].state = (const int*)(d + o2);
        o2 += (size_t)N * 4;
        memcpy(h + o2, q.slot2map, (size_t)N * 4);
        dc[c].slot2map = (const int*)(d + o2);
        o2 += (size_t)N * 4;
        dc[c].isStatic = nullptr;
        if (q.isDynamic) {
            memcpy(h + o2, q.isDynamic, (size_t)N);
            dc[c].isDynamic = (const unsigned char*)(d + o2);
        } else if (q.isStatic) {   // (staged as its complement)
            for (int i = 0; i < N; ++i) h[o2 + i] = q.isStatic[i] ? 0 : 1;
            dc[c].isDynamic = (const unsigned char*)(d + o2);
        } else {
            dc[c].isDynamic = nullptr;
        }
        off += camBytes;
    }
    const size_t oM = off, oC = oM + (size_t)P * 24, oF = oC + (size_t)P * 72;
    memcpy(h + oM, M, (size_t)P * 24);
    memcpy(h + oC, cov, (size_t)P * 72);
    memcpy(h + oF, pointFeat, pairs * 4);
    const size_t pad4 = (pairs * 4 + 7) & ~(size_t)7;
    const size_t om = inBytes, ov = om + pairs * 16, od = ov + pairs * 32, os = od + pairs * 8, ofl = os + pad4;
    hipStream_t s = nullptr;
    hipError_t e = hipMemcpyAsync(d, h, inBytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        rc = cs_register_search_dev(device, s, nCams, dc, N, W, H, P, (const double*)(d + oM), (const double*)(d + oC),
                                    (const int*)(d + oF), sigmaSearch, maxDist, sigmaMerge, (int*)(d + os), (double*)(d + om),
                                    (double*)(d + ov), (double*)(d + od), (int*)(d + ofl));
        if (rc == CS_OK) e = hipMemcpyAsync(h + inBytes, d + inBytes, outBytes, hipMemcpyDeviceToHost, s);
        if (rc == CS_OK && e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (rc == CS_OK && e == hipSuccess) {
        memcpy(m, h + om, pairs * 16);
        memcpy(var, h + ov, pairs * 32);
        memcpy(dist, h + od, pairs * 8);
        memcpy(slot, h + os, pairs * 4);
        memcpy(flags, h + ofl, pairs * 4);
    }
    (void)hipHostFree(h);
    (void)hipFree(d);
    if (rc != CS_OK) return rc;
    if (e != hipSuccess) {
        cs_set_error("cs_register_search: %s", hipGetErrorString(e));
        return CS_ERR_HIP;
    }
    return CS_OK;
}
