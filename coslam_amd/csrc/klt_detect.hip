// klt_detect.hip -- min-eigenvalue corner detector, non-max suppression, compaction, top-K and slot
// filling, gfx950.
//
// Replaces KLT_Detector::detectCorners / extractCorners (src/tracking/CGKLT/v3d_gpuklt.cpp:423-588 and
// Shaders/klt_detector_{pass1,pass2,nonmax,discriminator,build_histpyr,traverse_histpyr}.cg) and the CPU
// selection / slot-fill loops of KLT_SequenceTracker::{detect,redetect} (v3d_gpuklt.cpp:650-805).
//
// Design: the reference needs 5 full-screen passes, a log2(512)-level HistoPyramid, two synchronous
// glReadPixels and a CPU nth_element per frame.  Here
//   (cornerness)      the separable 7x7 structure tensor + min eigenvalue is computed by the level-0 pyramid kernel
//                     out of the tile it has just produced (klt_pyramid.hip, k_pyr_level0_corner),
//   k_post_track      per slot: status + the "present feature" -1e30 scatter (only behind the per-pass / no-gain
//                     trackers: the persistent gain tracker does this in its epilogue),
//   k_nonmax_compact  both separable non-max passes out of one LDS tile, survivors appended to a
//                     candidate list with one wave-aggregated atomic (replaces the HistoPyramid),
//   k_select_fill     one workgroup: rank sorts of the candidates (HistoPyramid order is Morton order of the pixel,
//                     top-K is by cornerness), then a scan over the dead slots writes dest[] and the next feature list.
// Nothing is read back mid-frame; the counts the reference reads with glReadPixels stay in HBM and
// kernels that depend on them early-exit on the device value.
#include "klt_front_dev.h"

#pragma clang fp contract(off)


namespace {

// ------------------------------------------------------------------ present-feature scatter
__device__ __forceinline__ void suppress_at(float* corner, int W, int H, float s, float t) {
    if (!(s >= 0.0f && t >= 0.0f)) return;
    float fx = floorf(s * (float)W), fy = floorf(t * (float)H);
    if (fx >= (float)W || fy >= (float)H) return;
    corner[(size_t)(int)fy * W + (int)fx] = -1e30f;  // v3d_gpuklt.cpp:444-447
}

__global__ void k_suppress_list(float* corner, int W, int H, int n, const float* __restrict__ list3) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) suppress_at(corner, W, H, list3[3 * k], list3[3 * k + 1]);
}

// v3d_gpuklt.cpp:872-888 (status loop of track()) fused with :744-752 + :475-500 (present list scatter)
__global__ void k_post_track(const float* __restrict__ feat, int N, cs_klt_feature* __restrict__ dest, int* ctr,
                             float* corner, int W, int H, int doSuppress) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float X = feat[3 * i], Y = feat[3 * i + 1], gain = feat[3 * i + 2];
    if (X >= 0) {
        dest[i].status = 0;
        dest[i].pos[0] = X;
        dest[i].pos[1] = Y;
        dest[i].gain = gain;
        dest[i].fed = -1;
        if (doSuppress) suppress_at(corner, W, H, X, Y);
    } else {
        dest[i].status = -1;
        dest[i].fed = -1;
    }
}

__global__ void k_clear_dest(cs_klt_feature* dest, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        dest[i].status = -1;
        dest[i].fed = -1;
    }
}

// ------------------------------------------------------------------ non-max + compaction
constexpr int NTW = 64, NTH = 16;

__device__ __forceinline__ unsigned part1by1(unsigned v) {
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

struct CsNonmaxArgs {
    const float* in;
    int W, H, d;
    float* out;
    CsCand* cand;
    int maxCand;
    int* ctr;
};

// one workgroup (256 threads) of the non-max + compaction stage: tile (bx, by)
__device__ __forceinline__ void nonmax_body(const CsNonmaxArgs& Z, int bx, int by, int tid, float* smem) {
    const float* __restrict__ in = Z.in;
    float* __restrict__ out = Z.out;
    CsCand* __restrict__ cand = Z.cand;
    int* ctr = Z.ctr;
    const int W = Z.W, H = Z.H, d = Z.d, maxCand = Z.maxCand;
    const int RW = NTW + 2 * d, RH = NTH + 2 * d;
    float* raw = smem;              // [RH][RW]
    float* rowres = smem + RH * RW;  // [RH][NTW]
    const int x0 = bx * NTW, y0 = by * NTH;
    // the tile's loads go out in batches of eight per thread before any LDS store: RW and RH are run-time values, the loop
    // is not unrolled, and one load per trip would be (RH * RW / 256) dependent round trips
    for (int base = tid; base < RH * RW; base += 8 * 256) {
        float tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 256 * u;
            const int ly = i / RW, lx = i - ly * RW;
            const int gx = cs_clampi(x0 + lx - d, 0, W - 1), gy = cs_clampi(y0 + ly - d, 0, H - 1);
            tv[u] = in[(size_t)gy * W + gx];  // (clamped: in range even when i >= RH * RW)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 256 * u;
            if (i < RH * RW) raw[i] = tv[u];
        }
    }
    __syncthreads();
    // klt_detector_nonmax.cg:12-26 with ds = (1/W, 0)
    for (int i = tid; i < RH * NTW; i += 256) {
        int ly = i / NTW, lx = i - ly * NTW;
        const float* r = raw + ly * RW + lx + d;
        float m = r[0];
        for (int k = -d; k < 0; ++k) {
            float cn = fabsf(r[k]);
            m = (cn >= fabsf(m)) ? (-cn) : m;
        }
        for (int k = 1; k <= d; ++k) {
            float cn = fabsf(r[k]);
            m = (cn >= fabsf(m)) ? (-cn) : m;
        }
        rowres[i] = m;
    }
    __syncthreads();
    // ... then ds = (0, 1/H), discriminator (value > 0) and append
    const int lx = tid & (NTW - 1), x = x0 + lx;
    for (int ly = tid / NTW; ly < NTH; ly += 256 / NTW) {
        int y = y0 + ly;
        if (x >= W || y >= H) continue;
        const float* r = rowres + (ly + d) * NTW + lx;
        float m = r[0];
        for (int k = -d; k < 0; ++k) {
            float cn = fabsf(r[k * NTW]);
            m = (cn >= fabsf(m)) ? (-cn) : m;
        }
        for (int k = 1; k <= d; ++k) {
            float cn = fabsf(r[k * NTW]);
            m = (cn >= fabsf(m)) ? (-cn) : m;
        }
        out[(size_t)y * W + x] = m;
        if (m > 0.0f) {
            int slot = atomicAdd(&ctr[0], 1);
            if (slot < maxCand) {
                CsCand cd;
                cd.key = part1by1((unsigned)x) | (part1by1((unsigned)y) << 1);
                cd.x = ((float)x + 0.5f) / (float)W;  // klt_detector_traverse_histpyr.cg:85-86
                cd.y = ((float)y + 0.5f) / (float)H;
                cd.c = m;
                cand[slot] = cd;
            }
        }
    }
}

struct CsNonmaxBatch {
    int W, H, d, maxCand;
    CsNonmaxCam cam[CS_MAX_CAMS];
};
__device__ __forceinline__ CsNonmaxArgs nonmax_args(const CsNonmaxBatch& B, int c) {
    CsNonmaxArgs Z = {B.cam[c].in, B.W, B.H, B.d, B.cam[c].out, B.cam[c].cand, B.maxCand, B.cam[c].ctr};
    return Z;
}

__global__ __launch_bounds__(256) void k_nonmax_compact(CsNonmaxBatch B) {  // blockIdx.z = camera
    extern __shared__ __attribute__((aligned(16))) float smem[];
    nonmax_body(nonmax_args(B, blockIdx.z), blockIdx.x, blockIdx.y, threadIdx.x, smem);
}

// ------------------------------------------------------------------ ordering, selection and slot fill
// One workgroup does what the reference does on the CPU after its read-backs (v3d_gpuklt.cpp:650-805): order the
// candidates (HistoPyramid order == Morton order of the pixel), keep the most distinctive ones when there are more
// than free slots (std::sort branch, :704-708,763-767: cornerness descending, ties in HistoPyramid order), and write
// them into the dead slots in ascending slot index together with the next frame's feature list.  Rank sorts are
// O(n^2 / 1024) per thread over the (normally <= a few hundred) candidates; one launch instead of three.
struct CsSelectArgs {
    const CsCand* cand;
    int maxCand, cap, maxKeepFixed;  // maxKeepFixed >= 0: use it; else maxKeep = N - ctr[1] (free slots after tracking)
    int* rankM;
    CsCand* sel;
};

// one workgroup of 1024 threads
__device__ __forceinline__ void select_fill_body(const CsSelectArgs& Q, const CsFillArgs& A) {
    __shared__ float cs[1024];
    __shared__ unsigned ks[1024];
    __shared__ int rs[1024];
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int N = A.N;
    // tracked features (redetect): status >= 0 in dest[], counted here in a fixed order
    __shared__ int nTrackedSh;
    {
        int c = 0;
        if (A.mode == 2)
            for (int i = tid; i < N; i += 1024) c += (A.dest[i].status >= 0) ? 1 : 0;
        part[tid] = c;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if (tid < s) part[tid] += part[tid + s];
            __syncthreads();
        }
        if (tid == 0) nTrackedSh = part[0];
        __syncthreads();
    }
    const int nTracked = nTrackedSh;
    const int n = min(A.ctr[0], Q.maxCand);
    const int nK = min(n, Q.cap);  // v3d_gpuklt.cpp:659,701,756: point list holds at most plw*plh
    int maxKeep = (Q.maxKeepFixed >= 0) ? Q.maxKeepFixed : (N - nTracked);
    if (maxKeep < 0) maxKeep = 0;
    const int nSel = min(nK, maxKeep);
    // ---- Morton rank ------------------------------------------------------------------------------------
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const unsigned ki = (i < n) ? Q.cand[i].key : 0u;
        int r = 0;
        for (int base = 0; base < n; base += 1024) {
            const int j = base + tid;
            ks[tid] = (j < n) ? Q.cand[j].key : 0xffffffffu;
            __syncthreads();
            const int m = min(1024, n - base);
            for (int q = 0; q < m; ++q) r += (ks[q] < ki) ? 1 : 0;
            __syncthreads();
        }
        if (i < n) Q.rankM[i] = r;
    }
    __syncthreads();
    // ---- selection ----------------------------------------------------------------------------------------
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        CsCand me;
        me.key = 0;
        me.x = me.y = me.c = 0;
        int myRank = 0x7fffffff;
        if (i < n) {
            me = Q.cand[i];
            myRank = Q.rankM[i];
        }
        if (nK <= maxKeep) {  // everything that fits the point list is used, in HistoPyramid order
            if (i < n && myRank < nK) Q.sel[myRank] = me;
            continue;
        }
        int r = 0;
        for (int base = 0; base < n; base += 1024) {
            const int j = base + tid;
            const bool ok = (j < n);
            cs[tid] = ok ? Q.cand[j].c : 0.0f;
            ks[tid] = ok ? Q.cand[j].key : 0u;
            rs[tid] = ok ? Q.rankM[j] : 0x7fffffff;
            __syncthreads();
            const int m = min(1024, n - base);
            for (int q = 0; q < m; ++q) {
                const bool in = rs[q] < nK;
                const bool before = (cs[q] > me.c) || (cs[q] == me.c && ks[q] < me.key);
                r += (in && before) ? 1 : 0;
            }
            __syncthreads();
        }
        if (i < n && myRank < nK && r < maxKeep) Q.sel[r] = me;
    }
    __syncthreads();
    // ---- slot fill (provideFeatures / provideFeaturesAndGain, :86-92,188-197) ---------------------------------
    const int chunk = (N + 1023) / 1024;
    const int lo = min(tid * chunk, N), hi = min(lo + chunk, N);
    int nDead = 0;
    for (int i = lo; i < hi; ++i) nDead += (A.mode == 2) ? (A.dest[i].status < 0 ? 1 : 0) : 1;
    part[tid] = nDead;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {  // inclusive Hillis-Steele scan
        int v = (tid >= s) ? part[tid - s] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int r = part[tid] - nDead;  // exclusive prefix = rank of my first free slot
    for (int i = lo; i < hi; ++i) {
        bool dead = (A.mode == 2) ? (A.dest[i].status < 0) : true;
        float lx, ly, lz;
        if (dead) {
            if (r < nSel) {
                CsCand c = Q.sel[r];
                cs_klt_feature f;
                f.status = 1;
                f.pos[0] = c.x;
                f.pos[1] = c.y;
                // redetect reports the cornerness (:782); detect reports data[2] after the gain reset (:716-727)
                f.gain = (A.mode == 2) ? c.c : (A.withGain ? 1.0f : c.c);
                f.fed = -1;
                A.dest[i] = f;
                lx = c.x;
                ly = c.y;
                lz = (A.mode == 2 || A.withGain) ? 1.0f : c.c;
            } else if (A.mode == 1 && r < nSel + A.nPresentGiven) {
                int q = r - nSel;
                cs_klt_feature f;
                f.status = 1;
                f.pos[0] = A.present3[3 * q];
                f.pos[1] = A.present3[3 * q + 1];
                f.gain = A.withGain ? 1.0f : 0.0f;
                f.fed = q;
                A.dest[i] = f;
                lx = f.pos[0];
                ly = f.pos[1];
                lz = f.gain;
            } else {
                A.dest[i].status = -1;
                if (A.mode != 1) A.dest[i].fed = -1;
                lx = ly = -1.0f;
                lz = (A.mode == 2 || A.withGain) ? 1.0f : -1.0f;
            }
            ++r;
        } else {
            lx = A.dest[i].pos[0];
            ly = A.dest[i].pos[1];
            lz = 1.0f;
        }
        A.list_a[3 * i] = lx;
        A.list_a[3 * i + 1] = ly;
        A.list_a[3 * i + 2] = lz;
        if (A.list_b) {
            A.list_b[3 * i] = lx;
            A.list_b[3 * i + 1] = ly;
            A.list_b[3 * i + 2] = lz;
        }
    }
    if (tid == 0) {
        int nPres = (A.mode == 2) ? nTracked : (A.mode == 1 ? min(A.nPresentGiven, max(N - nSel, 0)) : 0);
        A.ctr[1] = nTracked;
        A.ctr[2] = nSel;
        A.counts[0] = nSel + nPres;
        A.counts[1] = (A.mode == 2) ? nTracked : 0;
        A.counts[2] = A.ctr[0];
        A.counts[3] = nSel;
        A.ctr[0] = 0;  // every read of the candidate count is behind the barriers above
    }
    if (A.tagWord && tid == 0) *A.tagWord += 1u;
}

struct CsSelectBatch {
    int maxCand, cap;
    CsSelectCam cam[CS_MAX_CAMS];
};
__device__ __forceinline__ CsSelectArgs select_args(const CsSelectBatch& B, int c) {
    CsSelectArgs Q = {B.cam[c].cand, B.maxCand, B.cap, B.cam[c].maxKeepFixed, B.cam[c].rankM, B.cam[c].sel};
    return Q;
}

__global__ __launch_bounds__(1024) void k_select_fill(CsSelectBatch B) {  // one workgroup per camera
    select_fill_body(select_args(B, blockIdx.x), B.cam[blockIdx.x].fill);
}

// ---- detector tail fused with the NEXT frame's front (cs_klt_prefetch_dev) ---------------------------------------
// The tail is two small dependent launches that leave most of the chip idle (non-max: 300 workgroups; selection + slot
// fill: ONE workgroup), and the next frame's pyramid depends only on its image.  Horizontal fusion puts the two
// independent jobs into the same launches -- no second stream, no events:
//   k_tail_nonmax_level0   workgroups [0, nA): non-max + compaction of THIS frame; [nA, nA + nB): level 0 + cornerness
//                          map of the NEXT image (into the spare pyramid / cornerness buffers);
//   k_tail_select_down     workgroup 0: selection + slot fill of THIS frame; every other workgroup: one tile of the
//                          fused levels 1..3 of the NEXT pyramid (first four waves; the rest retire immediately).
struct CsLevel0Batch {
    int W, H;
    float minCornerness, lox, loy, hix, hiy;
    int nbx, nby;
    CsFrontCam cam[CS_MAX_CAMS];
};

__global__ __launch_bounds__(256) void k_tail_nonmax_level0(CsNonmaxBatch Z, int nax, int nA, CsLevel0Batch Y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, c = blockIdx.y;
    if (b < nA) {
        nonmax_body(nonmax_args(Z, c), b % nax, b / nax, threadIdx.x, smem);
    } else {
        const int q = b - nA;
        const CsFrontCam& F = Y.cam[c];
        cs_level0_body<true>(F.img, Y.W, Y.H, F.pyr, F.corner, Y.minCornerness, Y.lox, Y.loy, Y.hix, Y.hiy, nullptr, q % Y.nbx,
                             q / Y.nbx, Y.nbx, Y.nby, threadIdx.x, *(CsLevel0Lds<true>*)smem);
    }
}

struct CsTailPyr {
    cs_texel* pyr[CS_MAX_CAMS];
};
__global__ __launch_bounds__(1024) void k_tail_select_down(CsSelectBatch B, CsTailPyr P, CsDownFused F, int ntx, int nty) {
    extern __shared__ __attribute__((aligned(16))) unsigned char down_smem[];
    const int c = blockIdx.y;
    if (blockIdx.x == 0) {
        select_fill_body(select_args(B, c), B.cam[c].fill);
        return;
    }
    // one tile per workgroup, worked by its first four waves (the other twelve retire at once and leave the barriers):
    // the tiles spread over the CUs exactly as in the stand-alone launch
    if (threadIdx.x >= 256) return;
    const int t = blockIdx.x - 1;
    cs_down_body(P.pyr[c], F, t % ntx, t / ntx, threadIdx.x, (cs_texel*)down_smem, true);
}

// track() only: the tracked count (status >= 0 in dest[]), one workgroup
__global__ __launch_bounds__(1024) void k_counts_track(const cs_klt_feature* __restrict__ dest, int N, int* counts, int* ctr,
                                                       unsigned* tagWord) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    int c = 0;
    for (int i = tid; i < N; i += 1024) c += (dest[i].status >= 0) ? 1 : 0;
    part[tid] = c;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) part[tid] += part[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        counts[0] = part[0];
        counts[1] = part[0];
        counts[2] = 0;
        counts[3] = 0;
        if (ctr) ctr[0] = 0;
    }
    if (tagWord && tid == 0) *tagWord += 1u;
}

}  // namespace

int cs_launch_suppress_list(float* corner, int W, int H, int n, const float* d_list3, hipStream_t stream) {
    if (n <= 0) return CS_OK;
    hipLaunchKernelGGL(k_suppress_list, dim3((n + 255) / 256), dim3(256), 0, stream, corner, W, H, n, d_list3);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_launch_post_track(const float* feat, int N, cs_klt_feature* dest, int* ctr, float* corner, int W, int H,
                         int doSuppress, hipStream_t stream) {
    hipLaunchKernelGGL(k_post_track, dim3((N + 255) / 256), dim3(256), 0, stream, feat, N, dest, ctr, corner, W, H,
                       doSuppress);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_launch_clear_dest(cs_klt_feature* dest, int N, hipStream_t stream) {
    hipLaunchKernelGGL(k_clear_dest, dim3((N + 255) / 256), dim3(256), 0, stream, dest, N);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

size_t cs_nonmax_lds_bytes(int d) { return sizeof(float) * ((size_t)(NTH + 2 * d) * (NTW + 2 * d) + (size_t)(NTH + 2 * d) * NTW); }

static int fill_nonmax_batch(CsNonmaxBatch& B, const CsNonmaxCam* cams, int n, int W, int H, int d, int maxCand) {
    if (n < 1 || n > CS_MAX_CAMS) {
        cs_set_error("detector: %d cameras (1..%d)", n, CS_MAX_CAMS);
        return CS_ERR_INVALID;
    }
    memset(&B, 0, sizeof(B));
    B.W = W;
    B.H = H;
    B.d = d;
    B.maxCand = maxCand;
    for (int c = 0; c < n; ++c) B.cam[c] = cams[c];
    return CS_OK;
}
static void fill_select_batch(CsSelectBatch& B, const CsSelectCam* cams, int n, int maxCand, int cap) {
    memset(&B, 0, sizeof(B));
    B.maxCand = maxCand;
    B.cap = cap;
    for (int c = 0; c < n; ++c) B.cam[c] = cams[c];
}

int cs_launch_nonmax_compact(const CsNonmaxCam* cams, int n, int W, int H, int d, int maxCand, hipStream_t stream) {
    size_t lds = cs_nonmax_lds_bytes(d);
    if (lds > 160 * 1024) {
        cs_set_error("minDistance %d needs %zu B of LDS (> 160 KiB)", d, lds);
        return CS_ERR_INVALID;
    }
    CsNonmaxBatch B;
    int rc = fill_nonmax_batch(B, cams, n, W, H, d, maxCand);
    if (rc) return rc;
    dim3 grid((W + NTW - 1) / NTW, (H + NTH - 1) / NTH, n);
    hipLaunchKernelGGL(k_nonmax_compact, grid, dim3(256), lds, stream, B);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_nonmax_prepare(int d) {
    size_t lds = cs_nonmax_lds_bytes(d);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_nonmax_compact, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) {
            cs_set_error("hipFuncSetAttribute(maxDynamicShared=%zu) failed: %s", lds, hipGetErrorString(e));
            return CS_ERR_HIP;
        }
    }
    return CS_OK;
}

int cs_launch_select_fill(const CsSelectCam* cams, int n, int maxCand, int cap, hipStream_t stream) {
    if (n < 1 || n > CS_MAX_CAMS) {
        cs_set_error("selection: %d cameras (1..%d)", n, CS_MAX_CAMS);
        return CS_ERR_INVALID;
    }
    CsSelectBatch B;
    fill_select_batch(B, cams, n, maxCand, cap);
    hipLaunchKernelGGL(k_select_fill, dim3(n), dim3(1024), 0, stream, B);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_launch_counts_track(const cs_klt_feature* dest, int N, int* counts, int* ctr, unsigned* tagWord, hipStream_t stream) {
    hipLaunchKernelGGL(k_counts_track, dim3(1), dim3(1024), 0, stream, dest, N, counts, ctr, tagWord);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// detector tail of THIS frame + frame front of the NEXT image in two launches (see k_tail_* above), n cameras
int cs_launch_tail_with_next_front(const CsNonmaxCam* nm, const CsSelectCam* sel, const CsFrontCam* next, int n, int W, int H,
                                   int d, int maxCand, int cap, const CsPyrLayout& lay, int tap_mode, float minCornerness,
                                   float margin, hipStream_t stream) {
    size_t ldsA = cs_nonmax_lds_bytes(d);
    if (ldsA < sizeof(CsLevel0Lds<true>)) ldsA = sizeof(CsLevel0Lds<true>);
    if (ldsA > 64 * 1024) {
        cs_set_error("fused detector tail: minDistance %d needs %zu B of LDS", d, ldsA);
        return CS_ERR_INVALID;
    }
    CsNonmaxBatch Z;
    int rc = fill_nonmax_batch(Z, nm, n, W, H, d, maxCand);
    if (rc) return rc;
    const int nax = (W + NTW - 1) / NTW, nay = (H + NTH - 1) / NTH;
    CsLevel0Batch Y;
    memset(&Y, 0, sizeof(Y));
    Y.W = W;
    Y.H = H;
    Y.minCornerness = minCornerness;
    Y.lox = margin / (float)W;
    Y.loy = margin / (float)H;
    Y.hix = 1.0f - margin / (float)W;
    Y.hiy = 1.0f - margin / (float)H;
    Y.nbx = (W + FTW - 1) / FTW;
    Y.nby = (H + FTH - 1) / FTH;
    CsTailPyr P;
    memset(&P, 0, sizeof(P));
    for (int c = 0; c < n; ++c) {
        Y.cam[c] = next[c];
        P.pyr[c] = next[c].pyr;
    }
    hipLaunchKernelGGL(k_tail_nonmax_level0, dim3(nax * nay + Y.nbx * Y.nby, n), dim3(256), ldsA, stream, Z, nax, nax * nay, Y);
    CsSelectBatch B;
    fill_select_batch(B, sel, n, maxCand, cap);
    if (lay.L >= 2) {
        CsDownFused F;
        int ntx, nty;
        size_t lds;
        rc = cs_down_fused_plan(lay, tap_mode, &F, &ntx, &nty, &lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tail_select_down, dim3(1 + ntx * nty, n), dim3(1024), lds, stream, B, P, F, ntx, nty);
        rc = cs_launch_pyr_down_tail(P.pyr, n, lay, tap_mode, F.NL + 1, stream);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k_select_fill, dim3(n), dim3(1024), 0, stream, B);
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}
