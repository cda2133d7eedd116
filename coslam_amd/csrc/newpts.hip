// newpts.hip -- from the inter-camera NCC matrices to NEW MAP POINTS, on the device.
//
// Replaces what NewMapPtsNCC::run / output do behind getEpiNccMat (reference src/app/SL_NewMapPointsInterCam.cpp):
//   matchBetween's tail (:295-316)  the seeds of a camera pair (getSeedsBetween, :97-127: the map points both cameras see in this
//                                   frame), the disparity guide and the greedy one-to-one matching of the pair's candidates
//   featTracksFromMatches (:631-690) the matches of consecutive camera pairs chained into tracks, numbered in the order
//                                   (pair, feature of the pair's first camera)
//   reconstructTracks (:194-270)    per track of >= minLen views: triangulateMultiView, every view within maxRpErr pixels of the
//                                   re-projection and in front of its camera, getTriangulateCovMat, reprojErr of every feature, the
//                                   point's type (more than one DYNAMIC feature: dynamic, else uncertain)
//   output (:163-192)               decidePointType (:22-93: an uncertain point becomes locally static unless one of its features lies
//                                   within 20 pixels of a feature of a dynamic map point), the features take the point, the point
//                                   joins the current list
// The map is structure-of-arrays with spare capacity; new points are appended in track order behind *mapCount.
// Un-vendored LibVisualSLAM, OUR definitions (the oracle's, operation for operation):
//   greedyNCCMatch(ncc)                 the valid entries in order of falling score (ties: smaller row, then smaller column), an entry
//                                       is a match when neither its row nor its column has one yet
//   getDisparityMat(c1, c2, s1, s2, maxDisp)   entry (i, j) = | (c2_j - c1_i) - (s2_k - s1_k) | for the seed k nearest to c1_i in image
//                                       1 (first of equals), invalid when larger than maxDisp
//   greedyGuidedNCCMatch(ncc, disp)     greedyNCCMatch over the entries whose disparity entry is valid
//   reprojErrorSingle(K, R, t, M, m)    Euclidean pixel distance of m from the projection; dist2 likewise
// triangulateMultiView / getTriangulateCovMat / project / isAtCameraBack as in poseupdate.hip.
#include "cs_common.h"
#include "small_ops.h"

#pragma clang fp contract(off)

namespace {

constexpr int NP_MAX_CAMS = 16, NP_MAX_CAND = 2048, NP_MAX_SEEDS = 512, NP_MAX_TRACKS = 4096, NP_MAX_N = 32768;
constexpr int NP_MAX_DYN = 4096;       // features of certain dynamic points decidePointType's mask is drawn from (frame sizes up to 4011 x 4011)
constexpr int NP_SEGS = 8;             // a pair's seed scan is cut into this many map segments, a workgroup each (k_np_prep)
constexpr int NP_DYN_PER_CAM = 1024;   // features of certain dynamic points per camera that decidePointType's mask takes (k_np_prep)

struct NpArgs {
    int nCams, N, mapCap, curFrame, pairCap, minLen, W, H;
    double maxDisp, maxRpErr, sigma;
    const cs_ncc_pair* pairs[NP_MAX_CAMS];   // candidates of camera pair (a, a + 1): cs_ncc_epi_pairs_group_dev's list
    const int* pairCount[NP_MAX_CAMS];
    cs_poseupdate_cam cam[NP_MAX_CAMS];      // K, iK, xy, state, slot2map (written), reprojErr (written), isStatic
    const double *R, *t;                     // [nCams][9], [nCams][3]: the cameras' current poses
    double *mapPts, *mapCov;                 // [mapCap][3], [mapCap][9]
    unsigned char *mapFlags, *newPt;         // [mapCap]
    int *firstFrame, *pointFeat;             // [mapCap], [mapCap][nCams]
    int* mapCount;                           // [1] points in use: new ones are appended here
    int* matchIdx;                           // scratch [nCams - 1][N]: feature of camera a -> its match in camera a + 1 (matched rows only)
    unsigned* rowMask;                       // scratch [nCams - 1][N / 32 words]: bit i = feature i of camera a has a match in a + 1
    unsigned* colMask;                       // scratch [nCams - 1][N / 32 words]: bit j = feature j of camera a + 1 is such a match
    int* seedIdx;                            // scratch [nCams - 1][NP_SEGS][NP_MAX_SEEDS]: k_np_prep's seeds of a pair, map indices in map order
    int* seedCnt;                            // scratch [nCams - 1][NP_SEGS]
    unsigned* dynList;                       // scratch [nCams][NP_DYN_PER_CAM]: rounded positions of this frame's features on CERTAIN dynamic points
    int* dynCnt;                             // scratch [nCams]
    int* counts;                             // [4 + nCams] out: new points, tracks, tracks >= minLen, flags (bit 0: a candidate list
                                             // overflowed, bit 1: the map is full, bit 2: more than NP_MAX_DYN dynamic features), then the
                                             // matches of every pair
};

__device__ __forceinline__ bool np_before(double sa, unsigned ia, double sb, unsigned ib) {   // falling score, then rising (row, column)
    return sa > sb || (sa == sb && ia < ib);
}

// ---- preparation, many workgroups side by side (both were serial scans inside the one-workgroup kernels below: 48 / 15 us) -----------
// Workgroups [0, (nCams - 1) * NP_SEGS): getSeedsBetween (:97-127) of camera pair a for ONE segment of the map -- the certain, not
// false points with a feature of this frame in BOTH cameras (two features: numVisCam >= 2 holds), in map order (ballots per wave,
// the waves' totals through LDS), as map indices; k_np_match strings the segments together.
// Workgroups behind them, one per camera: the features of this frame that belong to CERTAIN dynamic map points -- what
// decidePointType's mask is drawn from (:38-59); this run's new dynamic points are added by k_np_reconstruct itself.
__global__ __launch_bounds__(256) void k_np_prep(NpArgs A) {
    __shared__ int wTot[4];
    __shared__ int sBase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, N = A.N, C = A.nCams, nP = C - 1;
    if (blockIdx.x == 0 && A.counts && tid < 4 + C) A.counts[tid] = 0;   // (the kernels behind this one add to them; was a launch of its own)
    if ((int)blockIdx.x < nP * NP_SEGS) {
        const int a = blockIdx.x / NP_SEGS, seg = blockIdx.x % NP_SEGS, b = a + 1;
        const int cap = *A.mapCount < A.mapCap ? *A.mapCount : A.mapCap;
        const int per = ((cap + NP_SEGS - 1) / NP_SEGS + 255) / 256 * 256;
        const int m0 = seg * per, m1 = min(m0 + per, cap);
        int* out = A.seedIdx + ((size_t)a * NP_SEGS + seg) * NP_MAX_SEEDS;
        if (tid == 0) sBase = 0;
        __syncthreads();
        for (int c0 = m0; c0 < m1 && sBase < NP_MAX_SEEDS; c0 += 256) {
            const int m = c0 + tid;
            bool in = false;
            if (m < m1) {
                const unsigned char fl = A.mapFlags[m];
                in = !(fl & (CS_MAP_FALSE | CS_MAP_UNCERTAIN)) && A.pointFeat[(size_t)m * C + a] >= 0 && A.pointFeat[(size_t)m * C + b] >= 0;
            }
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(in);
            if (lane == 0) wTot[wv] = __popcll(bal);
            __syncthreads();
            int k = sBase;
            for (int w = 0; w < wv; ++w) k += wTot[w];
            k += __popcll(bal & ((1ull << lane) - 1ull));
            if (in && k < NP_MAX_SEEDS) out[k] = m;
            __syncthreads();
            if (tid == 0) sBase = min(sBase + wTot[0] + wTot[1] + wTot[2] + wTot[3], NP_MAX_SEEDS);
            __syncthreads();
        }
        if (tid == 0) A.seedCnt[a * NP_SEGS + seg] = sBase;
        return;
    }
    const int c = blockIdx.x - nP * NP_SEGS;
    if (c >= C) return;
    if (tid == 0) sBase = 0;
    __syncthreads();
    for (int sl = tid; sl < N; sl += 256) {
        const int st = A.cam[c].state[sl], m = A.cam[c].slot2map[sl];
        if ((st == 0 || st == 1) && m >= 0 && m < A.mapCap && A.mapFlags[m] == CS_MAP_DYNAMIC) {
            const int x = (int)(A.cam[c].xy[sl] + 0.5), y = (int)(A.cam[c].xy[N + sl] + 0.5);
            if (x + 20 >= 0 && x - 20 < A.W && y + 20 >= 0 && y - 20 < A.H) {   // (a square that misses the image marks nothing)
                const int k = atomicAdd(&sBase, 1);
                if (k < NP_DYN_PER_CAM) A.dynList[(size_t)c * NP_DYN_PER_CAM + k] = ((unsigned)(y + 64) << 12) | (unsigned)(x + 64);
            }
        }
    }
    __syncthreads();
    if (tid == 0) A.dynCnt[c] = sBase;   // (beyond NP_DYN_PER_CAM: k_np_reconstruct raises flag bit 2)
}

// one workgroup per camera pair: seeds, disparity guide, the candidates sorted, the greedy walk
__global__ __launch_bounds__(256) void k_np_match(NpArgs A) {
    __shared__ double sKey[NP_MAX_CAND];
    __shared__ unsigned sIdx[NP_MAX_CAND];
    __shared__ double sSeed[NP_MAX_SEEDS][4];   // s1.x, s1.y, d.x, d.y (the first NP_MAX_SEEDS seeds in map order)
    __shared__ unsigned sRow[NP_MAX_N / 32], sCol[NP_MAX_N / 32];
    __shared__ int sNSeeds, sNCand;
    const int a = blockIdx.x, b = a + 1, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, N = A.N, C = A.nCams;
    const int nWords = (N + 31) / 32;
    for (int q = tid; q < nWords; q += 256) sRow[q] = 0, sCol[q] = 0;
    // the pair's seeds: k_np_prep's segments strung together (map order), the first NP_MAX_SEEDS of them
    {
        int off[NP_SEGS + 1];
        off[0] = 0;
#pragma unroll
        for (int g = 0; g < NP_SEGS; ++g) off[g + 1] = off[g] + A.seedCnt[a * NP_SEGS + g];
        const int total = off[NP_SEGS] < NP_MAX_SEEDS ? off[NP_SEGS] : NP_MAX_SEEDS;
        if (tid == 0) sNSeeds = total, sNCand = 0;
        for (int k = tid; k < total; k += 256) {
            int g = 0, base = 0;
#pragma unroll
            for (int q = 1; q < NP_SEGS; ++q)
                if (k >= off[q]) g = q, base = off[q];
            const int m = A.seedIdx[((size_t)a * NP_SEGS + g) * NP_MAX_SEEDS + (k - base)];
            const int s1 = A.pointFeat[(size_t)m * C + a], s2 = A.pointFeat[(size_t)m * C + b];
            const double x1 = A.cam[a].xy[s1], y1 = A.cam[a].xy[N + s1], x2 = A.cam[b].xy[s2], y2 = A.cam[b].xy[N + s2];
            sSeed[k][0] = x1, sSeed[k][1] = y1, sSeed[k][2] = x2 - x1, sSeed[k][3] = y2 - y1;
        }
    }
    __syncthreads();
    const int nSeeds = sNSeeds;
    // the pair's candidates: the disparity guide (when there are seeds: a WAVE per candidate, the seeds over the lanes, the nearest
    // -- the first of equals -- by a reduction), then into the sort arrays
    int nAll = *A.pairCount[a];
    if (nAll > A.pairCap) {   // (the NCC stage kept only pairCap of its passing pairs, in no fixed order: say so)
        nAll = A.pairCap;
        if (tid == 0 && A.counts) atomicOr(A.counts + 3, 1);
    }
    for (int q = wv; q < nAll; q += 4) {
        const cs_ncc_pair p = A.pairs[a][q];
        bool ok = p.i >= 0 && p.i < N && p.j >= 0 && p.j < N;   // (uniform over the wave)
        if (ok && nSeeds > 0) {
            const double x1 = A.cam[a].xy[p.i], y1 = A.cam[a].xy[N + p.i];
            double best = 1.0e300;
            int bk = 0x7fffffff;
            for (int k = lane; k < nSeeds; k += 64) {
                const double dx = sSeed[k][0] - x1, dy = sSeed[k][1] - y1, d2 = dx * dx + dy * dy;
                if (d2 < best) best = d2, bk = k;
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double ob = __shfl_xor(best, off, 64);
                const int ok_ = __shfl_xor(bk, off, 64);
                if (ob < best || (ob == best && ok_ < bk)) best = ob, bk = ok_;
            }
            if (bk == 0x7fffffff) bk = 0;   // (every distance NaN or >= 1e300: the scalar walk's index stays 0)
            const double ex = (A.cam[b].xy[p.j] - x1) - sSeed[bk][2];
            const double ey = (A.cam[b].xy[N + p.j] - y1) - sSeed[bk][3];
            ok = sqrt(ex * ex + ey * ey) <= A.maxDisp;
        }
        if (ok && lane == 0) {
            const int k = atomicAdd(&sNCand, 1);
            if (k < NP_MAX_CAND) sKey[k] = p.ncc, sIdx[k] = ((unsigned)p.i << 16) | (unsigned)p.j;
        }
    }
    __syncthreads();
    int L = sNCand;
    if (L > NP_MAX_CAND) {
        L = NP_MAX_CAND;
        if (tid == 0 && A.counts) atomicOr(A.counts + 3, 1);
    }
    // bitonic sort over the next power of two (padding sorts last)
    int L2 = 1;
    while (L2 < L) L2 <<= 1;
    for (int q = L + tid; q < L2; q += 256) sKey[q] = -1.0e300, sIdx[q] = 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= L2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < L2; q += 256) {
                const int x = q ^ j;
                if (x > q) {
                    const bool up = (q & k) == 0;
                    const double ka = sKey[q], kb = sKey[x];
                    const unsigned ia = sIdx[q], ib = sIdx[x];
                    if (np_before(kb, ib, ka, ia) == up) sKey[q] = kb, sIdx[q] = ib, sKey[x] = ka, sIdx[x] = ia;
                }
            }
            __syncthreads();
        }
    // the greedy walk: one lane, the list is short (a few hundred entries); the matches leave as two bit masks (rows of camera a,
    // columns of camera a + 1) and the matched rows' partners
    if (tid == 0) {
        int n = 0;
        int* match = A.matchIdx + (size_t)a * N;
        for (int q = 0; q < L; ++q) {
            const unsigned id = sIdx[q], i = id >> 16, j = id & 0xFFFFu;
            if ((sRow[i >> 5] >> (i & 31)) & 1u) continue;
            if ((sCol[j >> 5] >> (j & 31)) & 1u) continue;
            sRow[i >> 5] |= 1u << (i & 31), sCol[j >> 5] |= 1u << (j & 31);
            match[i] = (int)j;
            ++n;
        }
        if (A.counts) A.counts[4 + a] = n;
    }
    __syncthreads();
    for (int q = tid; q < nWords; q += 256) A.rowMask[(size_t)a * nWords + q] = sRow[q], A.colMask[(size_t)a * nWords + q] = sCol[q];
}

// ---- featTracksFromMatches + reconstructTracks + output: one workgroup, EIGHT LANES PER TRACK ------------------------------------------
// (Round 5's form gave a track to a lane: its views in a private array -- scratch memory --, every view's projections, divisions and
// square roots one after the other, and decidePointType's scan of the dynamic features' list by that one lane: 134 us on average for
// at most three tracks, profiles/r05_headline_bench_kernel_stats.md.)  A track's views are the cameras a0, a0 + 1, ..: lane r of the
// track's group holds views r and r + 8 in registers, computes their terms side by side, and the sums the reference takes view after
// view (:205-246) are taken in THAT order by fetching the terms lane by lane -- same additions, same order, same bits.  32 tracks per
// round of the 256 threads.
constexpr int NP_LPT = 8;                       // lanes per track
constexpr int NP_TPB = 256 / NP_LPT;            // tracks per round
static_assert(NP_MAX_CAMS <= 2 * NP_LPT, "a lane holds two views of a track");

struct NpTerm {   // what ONE view adds to the sums
    double n[6], g[3];
};
__device__ __forceinline__ double np_pick(double a, double b, bool second) { return second ? b : a; }
__device__ __forceinline__ int np_pick_i(int a, int b, bool second) { return second ? b : a; }

__global__ __launch_bounds__(256) void k_np_reconstruct(NpArgs A) {
    __shared__ int sBase, sNTracks, sWave[4];
    __shared__ unsigned sTrk[NP_MAX_TRACKS];             // first camera of the track << 16 | its slot there
    __shared__ int sDynOff[NP_MAX_CAMS + 2];
    const int tid = threadIdx.x, N = A.N, C = A.nCams, nP = C - 1;
    const int lane = tid & 63, wv = tid >> 6, r = tid & (NP_LPT - 1), gbase = lane & ~(NP_LPT - 1);
    if (tid == 0) sBase = 0, sNTracks = 0;
    __syncthreads();
    // ---- the tracks' starts in the order (pair, feature): (a, i) matched and not the continuation of a track of pair a - 1 (some
    // feature of camera a - 1 matched to i: then the match EXTENDS that track, :668-676) -- from the pairs' bit masks, 32 features a word
    const int nWords = (N + 31) / 32, totalW = nP * nWords;
    for (int w0 = 0; w0 < totalW; w0 += 256) {
        const int w = w0 + tid;
        unsigned bits = 0;
        int a = 0, q = 0;
        if (w < totalW) {
            a = w / nWords, q = w - a * nWords;
            bits = A.rowMask[w];
            if (a > 0) bits &= ~A.colMask[(size_t)(a - 1) * nWords + q];
        }
        const int cnt = __popc(bits);
        // inclusive scan of cnt over the block: inside the wave by shuffles, the waves' totals through LDS
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        if (lane == 63) sWave[wv] = inc;
        __syncthreads();
        int before = inc - cnt, total = 0;
        for (int k = 0; k < 4; ++k) {
            if (k < wv) before += sWave[k];
            total += sWave[k];
        }
        int rank = sNTracks + before;
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            if (rank < NP_MAX_TRACKS) sTrk[rank] = ((unsigned)a << 16) | (unsigned)(32 * q + bit);
            ++rank;
        }
        __syncthreads();
        if (tid == 0) sNTracks = min(sNTracks + total, NP_MAX_TRACKS);
        __syncthreads();
    }
    const int nTracks = sNTracks;
    int nLong = 0;
    // ---- reconstructTracks (:194-270), eight lanes per track
    for (int t0 = 0; t0 < nTracks; t0 += NP_TPB) {
        const int tk = t0 + tid / NP_LPT;
        // the chain: camera c's slot s matched into camera c + 1 (every lane of the group walks it; view k lives in lane k & 7)
        int nv = 0, c0 = 0, vs0 = -1, vs1 = -1;
        if (tk < nTracks) {
            int c = (int)(sTrk[tk] >> 16), sl = (int)(sTrk[tk] & 0xFFFFu);
            c0 = c;
            for (;;) {
                if ((nv & (NP_LPT - 1)) == r) (nv < NP_LPT ? vs0 : vs1) = sl;
                ++nv;
                if (c >= nP || !((A.rowMask[(size_t)c * nWords + (sl >> 5)] >> (sl & 31)) & 1u)) break;
                sl = A.matchIdx[(size_t)c * N + sl];
                ++c;
            }
        }
        const bool lng = tk < nTracks && nv >= A.minLen;
        if (lng && r == 0) ++nLong;
        bool valid = false;
        double M[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, err0 = 0, err1 = 0;
        unsigned char fl = 0;
        // (the whole wave goes through the phases together -- the shuffles need every lane -- with `lng` guarding the work)
        {
            // phase A: a view's terms of triangulateMultiView's normal equations (normPoint, the two rows, :205-216)
            NpTerm T0, T1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                NpTerm& T = kk ? T1 : T0;
#pragma unroll
                for (int q = 0; q < 6; ++q) T.n[q] = 0;
#pragma unroll
                for (int q = 0; q < 3; ++q) T.g[q] = 0;
                const int k = r + NP_LPT * kk, sl = kk ? vs1 : vs0;
                if (!lng || k >= nv) continue;
                const int cam = c0 + k;
                const cs_poseupdate_cam& Cm = A.cam[cam];
                const double* R = A.R + 9 * cam;
                const double* t = A.t + 3 * cam;
                const double mx = Cm.xy[sl], my = Cm.xy[N + sl];
                const double* iK = Cm.iK;
                const double w = (iK[6] * mx + iK[7] * my) + iK[8];
                const double x = ((iK[0] * mx + iK[1] * my) + iK[2]) / w, y = ((iK[3] * mx + iK[4] * my) + iK[5]) / w;   // normPoint
                const double a0[3] = {R[0] - x * R[6], R[1] - x * R[7], R[2] - x * R[8]}, a1[3] = {R[3] - y * R[6], R[4] - y * R[7], R[5] - y * R[8]};
                const double b0 = x * t[2] - t[0], b1 = y * t[2] - t[1];
                T.n[0] = a0[0] * a0[0] + a1[0] * a1[0];
                T.n[1] = a0[0] * a0[1] + a1[0] * a1[1];
                T.n[2] = a0[0] * a0[2] + a1[0] * a1[2];
                T.n[3] = a0[1] * a0[1] + a1[1] * a1[1];
                T.n[4] = a0[1] * a0[2] + a1[1] * a1[2];
                T.n[5] = a0[2] * a0[2] + a1[2] * a1[2];
#pragma unroll
                for (int q = 0; q < 3; ++q) T.g[q] = a0[q] * b0 + a1[q] * b1;
            }
            // the sums view after view, as the reference takes them
            double Nn[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            for (int k = 0; k < NP_MAX_CAMS; ++k) {
                if (!__builtin_amdgcn_ballot_w64(lng && k < nv)) break;
                const int src = gbase + (k & (NP_LPT - 1));
                const bool second = k >= NP_LPT;
                double tn[6], tg[3];
#pragma unroll
                for (int q = 0; q < 6; ++q) tn[q] = __shfl(np_pick(T0.n[q], T1.n[q], second), src, 64);
#pragma unroll
                for (int q = 0; q < 3; ++q) tg[q] = __shfl(np_pick(T0.g[q], T1.g[q], second), src, 64);
                if (lng && k < nv) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) Nn[q] = Nn[q] + tn[q];
#pragma unroll
                    for (int q = 0; q < 3; ++q) g[q] = g[q] + tg[q];
                }
            }
            double cf[6];
            if (lng) {
                cf[0] = Nn[3] * Nn[5] - Nn[4] * Nn[4], cf[1] = Nn[2] * Nn[4] - Nn[1] * Nn[5], cf[2] = Nn[1] * Nn[4] - Nn[2] * Nn[3];
                cf[3] = Nn[0] * Nn[5] - Nn[2] * Nn[2], cf[4] = Nn[1] * Nn[2] - Nn[0] * Nn[4], cf[5] = Nn[0] * Nn[3] - Nn[1] * Nn[1];
                const double det = (Nn[0] * cf[0] + Nn[1] * cf[1]) + Nn[2] * cf[2];
                M[0] = ((cf[0] * g[0] + cf[1] * g[1]) + cf[2] * g[2]) / det;
                M[1] = ((cf[1] * g[0] + cf[3] * g[1]) + cf[4] * g[2]) / det;
                M[2] = ((cf[2] * g[0] + cf[4] * g[1]) + cf[5] * g[2]) / det;
            }
            // phase B: every view within maxRpErr of the re-projection and in front of the camera (:226-236); its J^T J for the covariance
            double S0[6], S1[6];
            bool out = false, dyn0 = false, dyn1 = false;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                double* Sx = kk ? S1 : S0;
#pragma unroll
                for (int q = 0; q < 6; ++q) Sx[q] = 0;
                const int k = r + NP_LPT * kk, sl = kk ? vs1 : vs0;
                if (!lng || k >= nv) continue;
                const int cam = c0 + k;
                const cs_poseupdate_cam& Cm = A.cam[cam];
                const double* K = Cm.K;
                const double* R = A.R + 9 * cam;
                const double* t = A.t + 3 * cam;
                const double X = ((R[0] * M[0] + R[1] * M[1]) + R[2] * M[2]) + t[0];
                const double Y = ((R[3] * M[0] + R[4] * M[1]) + R[5] * M[2]) + t[1];
                const double Z = ((R[6] * M[0] + R[7] * M[1]) + R[8] * M[2]) + t[2];
                const double u = (K[0] * X + K[1] * Y) + K[2] * Z, vv = (K[3] * X + K[4] * Y) + K[5] * Z, w = (K[6] * X + K[7] * Y) + K[8] * Z;
                const double dx = Cm.xy[sl] - u / w, dy = Cm.xy[N + sl] - vv / w;
                const double e = sqrt(dx * dx + dy * dy);
                (kk ? err1 : err0) = e;
                if (e > A.maxRpErr || Z < 0) out = true;
                if (Cm.isStatic && !Cm.isStatic[sl]) (kk ? dyn1 : dyn0) = true;       // TYPE_FEATPOINT_DYNAMIC (:250-251)
                double KR[9], J[6];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) KR[3 * i + j] = (K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j]) + K[3 * i + 2] * R[6 + j];
                const double ww = w * w;
#pragma unroll
                for (int j = 0; j < 3; ++j) J[j] = (KR[j] * w - u * KR[6 + j]) / ww, J[3 + j] = (KR[3 + j] * w - vv * KR[6 + j]) / ww;
                Sx[0] = J[0] * J[0] + J[3] * J[3];
                Sx[1] = J[0] * J[1] + J[3] * J[4];
                Sx[2] = J[0] * J[2] + J[3] * J[5];
                Sx[3] = J[1] * J[1] + J[4] * J[4];
                Sx[4] = J[1] * J[2] + J[4] * J[5];
                Sx[5] = J[2] * J[2] + J[5] * J[5];
            }
            double S[6] = {0, 0, 0, 0, 0, 0};
            int nDyn = (dyn0 ? 1 : 0) + (dyn1 ? 1 : 0);
            for (int k = 0; k < NP_MAX_CAMS; ++k) {
                if (!__builtin_amdgcn_ballot_w64(lng && k < nv)) break;
                const int src = gbase + (k & (NP_LPT - 1));
                const bool second = k >= NP_LPT;
                double ts[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) ts[q] = __shfl(np_pick(S0[q], S1[q], second), src, 64);
                if (lng && k < nv) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) S[q] = S[q] + ts[q];
                }
            }
            // any view an outlier / the number of DYNAMIC features: over the group's lanes
#pragma unroll
            for (int o = 1; o < NP_LPT; o <<= 1) {
                out |= __shfl_xor((int)out, o, 64) != 0;
                nDyn += __shfl_xor(nDyn, o, 64);
            }
            if (lng && !out) {
                valid = true;
                cf[0] = S[3] * S[5] - S[4] * S[4], cf[1] = S[2] * S[4] - S[1] * S[5], cf[2] = S[1] * S[4] - S[2] * S[3];
                cf[3] = S[0] * S[5] - S[2] * S[2], cf[4] = S[1] * S[2] - S[0] * S[4], cf[5] = S[0] * S[3] - S[1] * S[1];
                const double dS = (S[0] * cf[0] + S[1] * cf[1]) + S[2] * cf[2], s2 = A.sigma * A.sigma;
                cov[0] = (cf[0] / dS) * s2, cov[1] = (cf[1] / dS) * s2, cov[2] = (cf[2] / dS) * s2;
                cov[3] = cov[1], cov[4] = (cf[3] / dS) * s2, cov[5] = (cf[4] / dS) * s2;
                cov[6] = cov[2], cov[7] = cov[5], cov[8] = (cf[5] / dS) * s2;
                // more than one DYNAMIC feature: dynamic (:253-254); else uncertain (:263-264; decidePointType below may make it certain static)
                fl = nDyn > 1 ? CS_MAP_DYNAMIC : CS_MAP_UNCERTAIN;
            }
        }
        // the valid tracks of this round take consecutive map indices in track order
        const unsigned long long lead = __builtin_amdgcn_ballot_w64(valid && r == 0);
        if (lane == 0) sWave[wv] = __popcll(lead);
        __syncthreads();
        int before = sBase + __popcll(lead & ((1ull << gbase) - 1ull)), total = 0;
        for (int k = 0; k < 4; ++k) {
            if (k < wv) before += sWave[k];
            total += sWave[k];
        }
        const int m = *A.mapCount + before;
        if (valid) {
            if (m < A.mapCap) {
                if (r < 3) A.mapPts[3 * (size_t)m + r] = M[r];
                // (cov[] is indexed by a lane-dependent subscript below: select instead, the array stays in registers)
                double cr = cov[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) cr = r == q ? cov[q] : cr;
                A.mapCov[9 * (size_t)m + r] = cr;
                if (r == 0) A.mapCov[9 * (size_t)m + 8] = cov[8], A.mapFlags[m] = fl, A.newPt[m] = 1, A.firstFrame[m] = A.curFrame;
                // MapPoint::addFeature: camera c0 + k's column takes view k's slot, every other column -1 (ONE store per column)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int c = r + NP_LPT * kk, k = c - c0;   // column c; the view there, if any
                    const int kSafe = (k >= 0 && k < nv) ? k : 0;
                    const int sv = __shfl(np_pick_i(vs0, vs1, kSafe >= NP_LPT), gbase + (kSafe & (NP_LPT - 1)), 64);
                    if (c < C) A.pointFeat[(size_t)m * C + c] = (k >= 0 && k < nv) ? sv : -1;
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int k = r + NP_LPT * kk, sl = kk ? vs1 : vs0;
                    if (k >= nv) continue;
                    const_cast<int*>(A.cam[c0 + k].slot2map)[sl] = m;   // the feature (and its track) takes the point (:168-172)
                    if (A.cam[c0 + k].reprojErr) A.cam[c0 + k].reprojErr[sl] = kk ? err1 : err0;   // fp->reprojErr (:247-248)
                }
            } else if (A.counts && r == 0) {
                atomicOr(A.counts + 3, 2);
            }
        }
        __syncthreads();
        if (tid == 0) sBase += total;
        __syncthreads();
    }
    // (nLong: counted by the groups' first lanes)
    if (nLong && A.counts) atomicAdd(A.counts + 2, nLong);
    __syncthreads();
    const int first = *A.mapCount;
    int added = sBase;
    if (first + added > A.mapCap) added = A.mapCap - first;
    // ---- decidePointType (:25-91): a new UNCERTAIN point none of whose features lies within 20 pixels (in rounded coordinates, both
    // axes) of a feature of this frame that belongs to a CERTAIN dynamic map point becomes certain static -- setLocalStatic()
    // clears bUncertain (src/slam/SL_MapPoint.cpp:104-109).  The dynamic points of this very run count: reconstructTracks'
    // addFeature already gave their features the point (src/slam/SL_MapPoint.cpp:58-69), so the scan below runs over the records
    // as they stand now.  (The barrier above makes this workgroup's writes to the map and to slot2map visible to all its threads.)
    bool anyUncertain = false;
    for (int q = tid; q < added; q += 256) anyUncertain |= A.mapFlags[first + q] == CS_MAP_UNCERTAIN;
    if (__syncthreads_or(anyUncertain ? 1 : 0)) {
        __shared__ int sNDyn;
        unsigned* sDyn = sTrk;   // the tracks' table is done with: [NP_MAX_DYN] (y + 64) << 12 | (x + 64), camera after camera
        static_assert(NP_MAX_TRACKS >= NP_MAX_DYN && NP_MAX_CAMS <= 256 && NP_MAX_N <= 65536, "the dynamic features' list reuses the track table");
        // the existing certain dynamic points' features: k_np_prep's per-camera lists, camera c at [sDynOff[c], sDynOff[c + 1])
        int off = 0;
        bool over = false;
        for (int c = 0; c < C; ++c) {
            int n = A.dynCnt[c];
            if (n > NP_DYN_PER_CAM) n = NP_DYN_PER_CAM, over = true;
            if (off + n > NP_MAX_DYN) n = NP_MAX_DYN - off, over = true;
            if (tid == 0) sDynOff[c] = off;
            for (int k = tid; k < n; k += 256) sDyn[off + k] = A.dynList[(size_t)c * NP_DYN_PER_CAM + k];
            off += n;
        }
        if (tid == 0) sDynOff[C] = off, sNDyn = off;
        __syncthreads();
        // ... and this run's new dynamic points (rare), behind them with their camera in the top byte
        for (int q = tid; q < added; q += 256) {
            const int m = first + q;
            if (A.mapFlags[m] != CS_MAP_DYNAMIC) continue;
            for (int c = 0; c < C; ++c) {
                const int sl = A.pointFeat[(size_t)m * C + c];
                if (sl < 0) continue;
                const int x = (int)(A.cam[c].xy[sl] + 0.5), y = (int)(A.cam[c].xy[N + sl] + 0.5);
                if (x + 20 >= 0 && x - 20 < A.W && y + 20 >= 0 && y - 20 < A.H) {
                    const int k = atomicAdd(&sNDyn, 1);
                    if (k < NP_MAX_DYN) sDyn[k] = ((unsigned)c << 24) | ((unsigned)(y + 64) << 12) | (unsigned)(x + 64);
                }
            }
        }
        __syncthreads();
        int nDynF = sNDyn;
        if (nDynF > NP_MAX_DYN) nDynF = NP_MAX_DYN, over = true;
        if (over && tid == 0 && A.counts) atomicOr(A.counts + 3, 4);
        const int offNew = sDynOff[C];
        // eight lanes per new point: a camera's list and this run's tail in strides of eight
        for (int q0 = 0; q0 < added; q0 += NP_TPB) {
            const int q = q0 + tid / NP_LPT, m = first + q;
            const bool unc = q < added && A.mapFlags[m] == CS_MAP_UNCERTAIN;
            bool hit = false;
            if (unc) {
                for (int c = 0; c < C; ++c) {
                    const int sl = A.pointFeat[(size_t)m * C + c];
                    if (sl < 0) continue;
                    const int x = (int)(A.cam[c].xy[sl] + 0.5), y = (int)(A.cam[c].xy[N + sl] + 0.5);
                    if (x < 0 || x >= A.W || y < 0 || y >= A.H) continue;
                    for (int k = sDynOff[c] + r; k < sDynOff[c + 1]; k += NP_LPT) {
                        const unsigned d = sDyn[k];
                        const int dx = (int)(d & 0xFFFu) - 64 - x, dy = (int)((d >> 12) & 0xFFFu) - 64 - y;
                        hit |= dx >= -20 && dx <= 20 && dy >= -20 && dy <= 20;
                    }
                    for (int k = offNew + r; k < nDynF; k += NP_LPT) {
                        const unsigned d = sDyn[k];
                        if ((int)(d >> 24) != c) continue;
                        const int dx = (int)(d & 0xFFFu) - 64 - x, dy = (int)((d >> 12) & 0xFFFu) - 64 - y;
                        hit |= dx >= -20 && dx <= 20 && dy >= -20 && dy <= 20;
                    }
                }
            }
#pragma unroll
            for (int o = 1; o < NP_LPT; o <<= 1) hit |= __shfl_xor((int)hit, o, 64) != 0;
            if (unc && !hit && r == 0) A.mapFlags[m] = 0;
        }
    }
    if (tid == 0) {
        if (A.counts) A.counts[0] = added, A.counts[1] = nTracks;
        *A.mapCount = first + added;
    }
}

// valid[i] = slot i can become a new map point: a feature of this frame on a track of at least minTrack + 1 frames (getTrackedFeatPts
// (.., 3), src/app/SL_SingleSLAM.cpp:173-184), unmapped or mapped to a FALSE point (NewMapPtsNCC::addSlam, SL_NewMapPointsInterCam.h:120-128)
__global__ __launch_bounds__(256) void k_np_candidates(int n, int N, const int* __restrict__ state, const int* __restrict__ slot2map,
                                                       const int* __restrict__ trackSpan, const unsigned char* __restrict__ mapFlags, int mapCap,
                                                       int minTrack, int* __restrict__ valid, size_t validStride) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const int c = q / N, s = q - c * N;
    const int st = state[q], m = slot2map[q];
    const int f1 = trackSpan[(size_t)c * 2 * N + s], f2 = trackSpan[(size_t)c * 2 * N + N + s];
    const bool tracked = (st == 0 || st == 1) && f1 >= 0 && f2 - f1 >= minTrack;
    const bool freeOrFalse = m < 0 || (m < mapCap && (mapFlags[m] & CS_MAP_FALSE));
    valid[(size_t)c * validStride + s] = tracked && freeOrFalse ? 1 : 0;
}

}  // namespace

extern "C" size_t cs_newpts_scratch_bytes(int nCams, int N) {
    if (nCams < 2 || N < 1) return 0;
    const size_t nP = (size_t)nCams - 1, nW = ((size_t)N + 31) / 32;
    return sizeof(int) * (nP * ((size_t)N + 2 * nW) + nP * NP_SEGS * (NP_MAX_SEEDS + 1) + (size_t)nCams * (NP_DYN_PER_CAM + 1));
}

extern "C" int cs_ncc_candidate_mask_dev(int device, void* hip_stream, int nCams, int N, const int* d_state, const int* d_slot2map,
                                         const int* d_trackSpan, const unsigned char* d_mapFlags, int mapCap, int minTrack, int* d_valid,
                                         size_t validStride) {
    if (nCams < 1 || N < 1 || !d_state || !d_slot2map || !d_trackSpan || !d_mapFlags || !d_valid || (validStride && validStride < (size_t)N)) {
        cs_set_error("cs_ncc_candidate_mask_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_np_candidates, dim3((nCams * N + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, nCams * N, N, d_state, d_slot2map,
                       d_trackSpan, d_mapFlags, mapCap, minTrack, d_valid, validStride ? validStride : (size_t)N);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_newpts_from_pairs_dev(int device, void* hip_stream, int nCams, int N, const cs_poseupdate_cam* cams,
                                        const cs_ncc_pair* const* d_pairs, const int* const* d_pairCount, int pairCap, const double* d_R,
                                        const double* d_t, double* d_mapPts, double* d_mapCov, unsigned char* d_mapFlags, unsigned char* d_newPt,
                                        int* d_firstFrame, int* d_pointFeat, int mapCap, int* d_mapCount, int curFrame, double maxDisp,
                                        double maxRpErr, double pixelErrVar, int minLen, int W, int H, void* d_scratch, int* d_counts) {
    // (the dynamic features' rounded positions travel as (y + 64) << 12 | (x + 64), 12 bits each, for positions within 20 px of the frame:
    // x + 20 + 64 <= 4095)
    if (W < 1 || H < 1 || W > 4011 || H > 4011) {
        cs_set_error("cs_newpts_from_pairs_dev: frame size 1..4011");
        return CS_ERR_INVALID;
    }
    if (nCams < 2 || nCams > NP_MAX_CAMS || N < 1 || N > NP_MAX_N || !cams || !d_pairs || !d_pairCount || pairCap < 1 || !d_R || !d_t || !d_mapPts ||
        !d_mapCov || !d_mapFlags || !d_newPt || !d_firstFrame || !d_pointFeat || mapCap < 1 || !d_mapCount || !d_scratch || minLen < 2) {
        cs_set_error("cs_newpts_from_pairs_dev: bad arguments (2..%d cameras, N <= %d)", NP_MAX_CAMS, NP_MAX_N);
        return CS_ERR_INVALID;
    }
    NpArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = nCams, A.N = N, A.mapCap = mapCap, A.curFrame = curFrame, A.pairCap = pairCap, A.minLen = minLen, A.W = W, A.H = H;
    A.maxDisp = maxDisp, A.maxRpErr = maxRpErr, A.sigma = pixelErrVar;
    for (int a = 0; a + 1 < nCams; ++a) {
        if (!d_pairs[a] || !d_pairCount[a]) {
            cs_set_error("cs_newpts_from_pairs_dev: null candidate list of camera pair %d", a);
            return CS_ERR_INVALID;
        }
        A.pairs[a] = d_pairs[a], A.pairCount[a] = d_pairCount[a];
    }
    for (int c = 0; c < nCams; ++c) {
        if (!cams[c].K || !cams[c].iK || !cams[c].xy || !cams[c].state || !cams[c].slot2map) {
            cs_set_error("cs_newpts_from_pairs_dev: null pointer in camera %d (K, iK, xy, state, slot2map)", c);
            return CS_ERR_INVALID;
        }
        A.cam[c] = cams[c];
    }
    A.R = d_R, A.t = d_t, A.mapPts = d_mapPts, A.mapCov = d_mapCov, A.mapFlags = d_mapFlags, A.newPt = d_newPt, A.firstFrame = d_firstFrame;
    A.pointFeat = d_pointFeat, A.mapCount = d_mapCount, A.matchIdx = (int*)d_scratch;
    A.rowMask = (unsigned*)d_scratch + (size_t)(nCams - 1) * N, A.colMask = A.rowMask + (size_t)(nCams - 1) * ((N + 31) / 32);
    A.seedIdx = (int*)(A.colMask + (size_t)(nCams - 1) * ((N + 31) / 32));
    A.seedCnt = A.seedIdx + (size_t)(nCams - 1) * NP_SEGS * NP_MAX_SEEDS;
    A.dynList = (unsigned*)(A.seedCnt + (size_t)(nCams - 1) * NP_SEGS);
    A.dynCnt = (int*)(A.dynList + (size_t)nCams * NP_DYN_PER_CAM);
    A.counts = d_counts;
    CS_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(k_np_prep, dim3((nCams - 1) * NP_SEGS + nCams), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_np_match, dim3(nCams - 1), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_np_reconstruct, dim3(1), dim3(256), 0, s, A);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
