// ncc.hip -- NCC blocks and the epipolar / NCC matrices of CoSLAM's inter-camera matching, gfx950.
//
// Replaces
//   NCCBlock::compute / computeScaled   src/slam/SL_NCCBlock.cpp:15-54    one 11 x 11 block per feature + its A, B, C
//   matchNCCBlock                       src/slam/SL_NCCBlock.cpp:258-264  121 byte products per PAIR of features
//   getEpiNccMat                        src/slam/SL_FeatureMatching.cpp:3-46   M x N epipolar errors and NCC scores --
// what NewMapPtsNCC::matchBetween (src/app/SL_NewMapPointsInterCam.cpp:273-317, every <= 4 frames per camera pair,
// SL_CoSLAM.cpp:1368-1371) computes on one host core before its greedy matcher: 2000 x 2000 pairs x 121 products.
//
// The pair scores are a matrix product of byte rows -- this one IS GEMM-shaped, so it runs on the matrix cores:
// v_mfma_i32_16x16x32_i8.  The blocks are unsigned bytes and the instruction multiplies signed ones, so the rows are fed
// as (I - 128) (one XOR with 0x80 per byte on the way in; rows are padded to 128 bytes with 0x80 = 0 after the shift) and
// the exact integer sum  sum I1 I2 = sum (I1 - 128)(I2 - 128) + 128 (A1 - 121 * 128) + 128 (A2 - 121 * 128) + 121 * 128^2
// is rebuilt from the blocks' own sums A.  Everything up to here is integer arithmetic: exact.  The score itself,
// ((121 d - A1 A2) C1) C2, and the epipolar error are binary64 in the reference's operation order: the matrices are
// bit-identical to the reference's (tests: the reference's own SL_NCCBlock.cpp / SL_FeatureMatching.cpp compiled in place).
// A workgroup = a 64 x 64 tile of pairs (four waves, 16 rows each, 16 MFMAs per wave); the output -- two M x N binary64
// matrices, what the reference's matcher reads -- is the traffic that bounds it (16 bytes per pair).
//
// epipolarError is un-vendored LibVisualSLAM: definition in DESIGN.md (distance of the first point from the line F (second
// point, 1)).  How matchBetween cuts its blocks (cv::getRectSubPix on a cv::resize'd image) is OpenCV; the in-tree
// NCCBlock::compute (truncated position on the small image) is what cs_ncc_blocks* implements.
#include "cs_common.h"
#include "small_ops.h"

#pragma clang fp contract(off)

namespace {

constexpr int NC_HW = 5, NC_BW = 11, NC_LEN = 121, NC_PITCH = 128;

// ---- NCCBlock::computeScaled for n features: one wave per feature ---------------------------------------------------
__global__ __launch_bounds__(256) void k_ncc_blocks(const unsigned char* __restrict__ img, int W, int H, int n,
                                                    const double* __restrict__ xs, const double* __restrict__ ys, double scale,
                                                    unsigned char* __restrict__ blocks, double* __restrict__ abc,
                                                    int* __restrict__ valid) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const double x = xs[i] * scale, y = ys[i] * scale;  // SL_NCCBlock.cpp:51-54
    const int x0 = (int)x, y0 = (int)y;                  // :20-21
    const bool ok = !(x0 - NC_HW < 0 || x0 + NC_HW >= W || y0 - NC_HW < 0 || y0 + NC_HW >= H);  // :24-26
    unsigned v0 = 0x80, v1 = 0x80;  // bytes lane and lane + 64 of the padded row
    if (ok) {
        const int j0 = lane, j1 = lane + 64;
        const int yy0 = j0 / NC_BW, xx0 = j0 - yy0 * NC_BW;
        v0 = img[(size_t)(y0 + yy0 - NC_HW) * W + (x0 + xx0 - NC_HW)];
        if (j1 < NC_LEN) {
            const int yy1 = j1 / NC_BW, xx1 = j1 - yy1 * NC_BW;
            v1 = img[(size_t)(y0 + yy1 - NC_HW) * W + (x0 + xx1 - NC_HW)];
        }
    }
    blocks[(size_t)i * NC_PITCH + lane] = (unsigned char)v0;
    blocks[(size_t)i * NC_PITCH + 64 + lane] = (unsigned char)v1;
    // A = sum I, B = sum I^2: integers (< 2^23), so any summation order gives the reference's binary64 values exactly
    // (select the BYTES, then square: `v0 * v0 + (in ? v1 * v1 : 0)` is turned into v_dot4_u32_u8 with the predicate as
    // the accumulator by this compiler -- v0^2 + v1^2 + in -- which the golden test caught)
    const unsigned w0 = ok ? v0 : 0u, w1 = (ok && lane + 64 < NC_LEN) ? v1 : 0u;
    unsigned a = w0 + w1;
    unsigned b = w0 * w0 + w1 * w1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if (lane == 0) {
        valid[i] = ok ? 1 : 0;
        const double A = (double)a, B = (double)b;
        abc[4 * (size_t)i] = ok ? A : 0.0;
        abc[4 * (size_t)i + 1] = ok ? B : 0.0;
        abc[4 * (size_t)i + 2] = ok ? 1 / sqrt((double)NC_LEN * B - A * A) : 0.0;  // :49
        abc[4 * (size_t)i + 3] = ok ? A / (double)NC_LEN : 0.0;                     // :40 (the byte sum is exact)
    }
}

// ---- getNCCBlocks: how matchBetween really cuts its blocks (SL_NCCBlock.cpp:79-155) -----------------------------------
// cv::resize(img, small, Size(), scale, scale) [INTER_LINEAR, 8-bit] once per image, then per point
// cv::getRectSubPix(small, 11 x 11, (x scale, y scale)) [8u -> 8u] and A, B, C.  OpenCV is not in this image: both functions
// are the published generic C++ paths restated (imgproc/src/resize.cpp: 11-bit coefficients, the (b (S >> 4)) >> 16 vertical
// pass; imgproc/src/samplers.cpp getRectSubPix_Cn_ + adjustRect: 16-bit fixed-point bilinear weights, replicated border) --
// integer arithmetic behind float weights, so the kernels are bit-exact against the CPU restatement the tests hold (which is itself unpinned: no OpenCV here).
__device__ __forceinline__ int nc_floorf(float v) {
    const int i = (int)v;
    return i - (i > v);
}
__device__ __forceinline__ int nc_clip(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }
__device__ __forceinline__ int nc_sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// the cameras of a group launch (blockIdx.z / blockIdx.y): cs_ncc_get_blocks_group_dev
constexpr int NC_MAX_CAMS = 16;
struct NcCamSet {
    const unsigned char* img[NC_MAX_CAMS];
    unsigned char* scaled[NC_MAX_CAMS];
    const double* x[NC_MAX_CAMS];
    const double* y[NC_MAX_CAMS];
    unsigned char* blocks[NC_MAX_CAMS];
    double* abc[NC_MAX_CAMS];
    int* valid[NC_MAX_CAMS];
};

// one thread per destination pixel; (dx, dy) -> the two source columns / rows and their 11-bit weights exactly as
// cv::resize's tables hold them; blockIdx.z = camera
__global__ __launch_bounds__(256) void k_resize_linear_u8(NcCamSet S, int W, int H, double scale_x, double scale_y, int Wd, int Hd) {
    const unsigned char* __restrict__ src = S.img[blockIdx.z];
    unsigned char* __restrict__ dst = S.scaled[blockIdx.z];
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= Wd || dy >= Hd) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = nc_floorf(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    bool edge = false;  // dx >= xmax: xmax is the first dx with sx + 1 >= W, and sx is monotone in dx
    if (sx + 1 >= W) {
        edge = true;
        if (sx >= W - 1) fx = 0, sx = W - 1;
    }
    const int a0 = nc_sat_short(__float2int_rn((1.f - fx) * 2048.f)), a1 = nc_sat_short(__float2int_rn(fx * 2048.f));
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = nc_floorf(fy);
    fy -= sy;
    const int b0 = nc_sat_short(__float2int_rn((1.f - fy) * 2048.f)), b1 = nc_sat_short(__float2int_rn(fy * 2048.f));
    const unsigned char* S0 = src + (size_t)nc_clip(sy, 0, H) * W;
    const unsigned char* S1 = src + (size_t)nc_clip(sy + 1, 0, H) * W;
    int r0, r1;
    if (!edge) {
        r0 = S0[sx] * a0 + S0[sx + 1] * a1;
        r1 = S1[sx] * a0 + S1[sx + 1] * a1;
    } else {
        r0 = S0[sx] * 2048;
        r1 = S1[sx] * 2048;
    }
    dst[(size_t)dy * Wd + dx] = (unsigned char)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// one wave per point: lanes 0..120 = the pixels of the 11 x 11 patch (two per lane); cv::getRectSubPix 8u -> 8u.
// The border path of the original walks the rows with a pointer that stops advancing outside the image; in closed form the
// top source row of window row i is row0 + max(0, min(i, rh) - ry), the bottom one the next row unless i < ry or i >= rh.
__global__ __launch_bounds__(256) void k_ncc_blocks_subpix(NcCamSet S, int W, int H, int n, double scale, int scaled) {
    const int cam = blockIdx.y;  // (of a group launch)
    const unsigned char* __restrict__ img = scaled ? S.scaled[cam] : S.img[cam];
    const double* __restrict__ xs = S.x[cam];
    const double* __restrict__ ys = S.y[cam];
    unsigned char* __restrict__ blocks = S.blocks[cam];
    double* __restrict__ abc = S.abc[cam];
    int* __restrict__ valid = S.valid[cam];
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n) return;
    const int win = NC_BW;
    const double px = scaled ? xs[p] * scale : xs[p], py = scaled ? ys[p] * scale : ys[p];  // SL_NCCBlock.cpp:131-132 / :99-100
    float fxc = (float)px, fyc = (float)py;
    fxc -= (win - 1) * 0.5f;
    fyc -= (win - 1) * 0.5f;
    const int ipx = nc_floorf(fxc), ipy = nc_floorf(fyc);
    const float a = fxc - ipx, b = fyc - ipy;
    const int a11 = __float2int_rn((1.f - a) * (1.f - b) * 65536.f), a12 = __float2int_rn(a * (1.f - b) * 65536.f),
              a21 = __float2int_rn((1.f - a) * b * 65536.f), a22 = __float2int_rn(a * b * 65536.f);
    const int b1 = __float2int_rn((1.f - b) * 65536.f), b2 = __float2int_rn(b * 65536.f);
    const bool inside = 0 <= ipx && ipx < W - win && 0 <= ipy && ipy < H - win;
    int rx = 0, rw = win, ry = 0, rh = win, col0 = ipx, row0 = ipy;
    if (!inside) {  // adjustRect
        if (ipx >= 0) {
            col0 = ipx, rx = 0;
        } else {
            col0 = 0, rx = -ipx;
            if (rx > win) rx = win;
        }
        if (ipx < W - win) {
            rw = win;
        } else {
            rw = W - ipx - 1;
            if (rw < 0) col0 += rw, rw = 0;
        }
        if (ipy >= 0) {
            row0 = ipy, ry = 0;
        } else {
            row0 = 0, ry = -ipy;
        }
        if (ipy < H - win) {
            rh = win;
        } else {
            rh = H - ipy - 1;
            if (rh < 0) row0 += rh, rh = 0;
        }
    }
    unsigned v[2] = {0x80u, 0x80u};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h;
        if (q >= NC_LEN) continue;
        const int i = q / win, j = q - i * win;
        int t;
        if (inside) {
            const unsigned char* s = img + (size_t)(ipy + i) * W + ipx;
            t = s[j] * a11 + s[j + 1] * a12 + s[j + W] * a21 + s[j + W + 1] * a22;
        } else {
            const int adv = (i < rh ? i : rh) - ry;
            const int rt = row0 + (adv > 0 ? adv : 0);
            const int rb = rt + ((i < ry || i >= rh) ? 0 : 1);
            const unsigned char* s = img + (size_t)rt * W + (col0 - rx);
            const unsigned char* s2 = img + (size_t)rb * W + (col0 - rx);
            if (j < rx)
                t = s[rx] * b1 + s2[rx] * b2;
            else if (j >= rw)
                t = s[rw] * b1 + s2[rw] * b2;
            else
                t = s[j] * a11 + s[j + 1] * a12 + s2[j] * a21 + s2[j + 1] * a22;
        }
        v[h] = (unsigned)((t + (1 << 15)) >> 16) & 0xffu;
    }
    blocks[(size_t)p * NC_PITCH + lane] = (unsigned char)v[0];
    blocks[(size_t)p * NC_PITCH + 64 + lane] = (unsigned char)v[1];
    const unsigned w0 = v[0], w1 = (lane + 64 < NC_LEN) ? v[1] : 0u;   // (select the bytes, then square: see k_ncc_blocks)
    unsigned sa = w0 + w1;
    unsigned sb = w0 * w0 + w1 * w1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sa += __shfl_xor(sa, off, 64);
        sb += __shfl_xor(sb, off, 64);
    }
    if (lane == 0) {
        const double A = (double)sa, B = (double)sb;
        abc[4 * (size_t)p] = A;
        abc[4 * (size_t)p + 1] = B;
        abc[4 * (size_t)p + 2] = 1 / sqrt((double)NC_LEN * B - A * A);
        abc[4 * (size_t)p + 3] = A / (double)NC_LEN;
        if (valid) valid[p] = 1;
    }
}

// which slots are features matchBetween would be given: in this frame's list (hand-back state 0 / 1) and without a map point
// (NewMapPtsNCC::addSlam collects the unmapped feature points of the current frame)
__global__ __launch_bounds__(256) void k_ncc_unmapped_mask(int n, const int* __restrict__ state, const int* __restrict__ slot2map,
                                                           int* __restrict__ valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) valid[i] = (state[i] >= 0 && slot2map[i] < 0) ? 1 : 0;
}

// ---- getEpiNccMat: 64 x 64 pairs per workgroup ------------------------------------------------------------------------
struct NcSide {
    const double* x;
    const double* y;
    const unsigned char* blocks;
    const double* abc;
    const int* valid;
    int n;
};
struct NcArgs {
    double F[9];
    const double* dF;   // null, or the fundamental matrix in device memory (cs_ncc_fmats_dev: formed from the poses the frame has just solved)
    NcSide s1, s2;
    double epiMax, nccMin, wNone;
    double* epiMat;
    double* nccMat;
    cs_ncc_pair* pairs;  // SPARSE: the entries that pass both tests, as records; capacity pairCap; *pairCount counts every one
    int pairCap;
    int* pairCount;
};

typedef int nc_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long nc_row_chunk(const unsigned char* blocks, int row, int n, int byteOff) {
    // eight bytes of a padded row, shifted to signed (x ^ 0x80 == x - 128 in two's complement); rows beyond n are zeros
    if (row >= n) return 0;
    const unsigned long long u = *(const unsigned long long*)(blocks + (size_t)row * NC_PITCH + byteOff);
    return (long)(u ^ 0x8080808080808080ull);
}

// SPARSE = false: the two dense M x N matrices getEpiNccMat fills (-1 / wNone where a pair fails a test) -- 16 bytes per pair,
// 64 MB for 2000 x 2000: what bounds the kernel.  SPARSE = true: only the pairs that pass, as {i, j, epiErr, ncc} records
// appended through one atomic counter (the order of the list is not defined; the matrices are its scatter into wNone-filled
// arrays): a matching run of 7 camera pairs then writes kilobytes instead of 448 MB.
constexpr int NC_MAX_JOBS = 8;
struct NcJobs {
    NcArgs job[NC_MAX_JOBS];  // blockIdx.z = camera pair of a group launch (cs_ncc_epi_pairs_group_dev)
};
constexpr int NC_CT = 4;   // 64-column tiles a workgroup walks: the rows' operands and constants are loaded once for 256 columns
// The column side of a tile goes through LDS: the 64 columns' blocks are 8 KB back to back in memory -- 256 threads x two 16-byte loads,
// shifted to signed bytes on the way in -- and their constants (x, y, A, C, valid) by the first 64 threads; the tile after it is fetched into
// registers while this one is multiplied and tested, then stored into the other buffer.  Until r04 every wave read its MFMA operands
// and the epilogue's constants straight from global memory: 36 short loads per lane and tile, 0.73 of the wave cycles waiting at one
// workgroup per compute unit beside the tracker (profiles/r04_ab_runs.txt).
constexpr int NC_TP = 136;                         // LDS pitch of a staged block row (128 + 8: consecutive rows on different banks)
constexpr int NC_TILE_BYTES = 64 * NC_TP + 64 * 40;   // blocks | x[64] | y[64] | A[64] | C[64] (doubles) | valid[64] (ints)
struct NcColRegs {   // one thread's share of a tile on its way from global memory to LDS
    nc_i32x4 q0, q1;
    double x, y, A, C;
    int valid;
};
__device__ __forceinline__ void nc_tile_fetch(const NcSide& S2, int j0, int N, int tid, NcColRegs& r) {
    // thread t: bytes 32 t .. 32 t + 31 of the tile's 8 KB = row t / 4, quarter t % 4; rows beyond N are zeros
    const int row = j0 + (tid >> 2);
    r.q0 = r.q1 = (nc_i32x4){0, 0, 0, 0};
    if (row < N) {
        const nc_i32x4* src = (const nc_i32x4*)(S2.blocks + (size_t)row * NC_PITCH + 32 * (tid & 3));
        r.q0 = src[0], r.q1 = src[1];
        const int sh = (int)0x80808080;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.q0[k] ^= sh, r.q1[k] ^= sh;   // x ^ 0x80 == x - 128 in two's complement
    }
    r.x = r.y = r.A = r.C = 0.0, r.valid = 0;
    if (tid < 64 && j0 + tid < N) {
        const int j = j0 + tid;
        r.x = S2.x[j], r.y = S2.y[j], r.A = S2.abc[4 * (size_t)j], r.C = S2.abc[4 * (size_t)j + 2], r.valid = S2.valid[j];
    }
}
__device__ __forceinline__ void nc_tile_store(unsigned char* buf, int tid, const NcColRegs& r) {
    nc_i32x4* dst = (nc_i32x4*)(buf + (tid >> 2) * NC_TP + 32 * (tid & 3));   // (136 t / 4 + 32 (t % 4): 8-byte aligned -> two 8-byte halves each)
    long* d8 = (long*)dst;
    const long* s8 = (const long*)&r.q0;
    d8[0] = s8[0], d8[1] = s8[1];
    s8 = (const long*)&r.q1;
    d8[2] = s8[0], d8[3] = s8[1];
    if (tid < 64) {
        double* c = (double*)(buf + 64 * NC_TP);
        c[tid] = r.x, c[64 + tid] = r.y, c[128 + tid] = r.A, c[192 + tid] = r.C;
        ((int*)(c + 256))[tid] = r.valid;
    }
}
template <bool SPARSE>
__global__ __launch_bounds__(256) void k_ncc_epi_mat(NcJobs J) {
    __shared__ __attribute__((aligned(16))) unsigned char sTile[2][NC_TILE_BYTES];
    const NcArgs& A = J.job[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int M = A.s1.n, N = A.s2.n;
    if ((int)blockIdx.y * 64 >= M || (int)blockIdx.x * NC_CT * 64 >= N) return;   // (uniform over the workgroup)
    const int i0 = blockIdx.y * 64 + 16 * wv;  // this wave's 16 rows (features of camera 1)
    const bool rowsIn = i0 < M;                // (a wave past the last row still helps staging the columns)
    double Fm[9];                              // the pair's fundamental matrix: by value, or from device memory (uniform loads)
#pragma unroll
    for (int k = 0; k < 9; ++k) Fm[k] = A.dF ? A.dF[k] : A.F[k];
    const int lr = lane & 15, lk = lane >> 4;  // MFMA operand layout: row / column lr, k-chunk lk (8 bytes)
    // the rows' side, once: the four k-chunks of the A operand, and what the epilogue needs of rows 4 lk .. 4 lk + 3
    long a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a[ks] = nc_row_chunk(A.s1.blocks, i0 + lr, M, 32 * ks + 8 * lk);
    double x1[4], y1[4], A1[4], C1[4];
    int v1[4], s1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * lk + r;
        const bool in = i < M;
        x1[r] = in ? A.s1.x[i] : 0.0;
        y1[r] = in ? A.s1.y[i] : 0.0;
        A1[r] = in ? A.s1.abc[4 * (size_t)i] : 0.0;
        C1[r] = in ? A.s1.abc[4 * (size_t)i + 2] : 0.0;
        v1[r] = in ? A.s1.valid[i] : 0;
        s1[r] = (int)A1[r] - NC_LEN * 128;  // sum (I1 - 128)
    }
    const int jBase = blockIdx.x * NC_CT * 64;
    NcColRegs cr;
    nc_tile_fetch(A.s2, jBase, N, tid, cr);
    nc_tile_store(sTile[0], tid, cr);
    __syncthreads();
    for (int ct = 0; ct < NC_CT; ++ct) {
        const int j0 = jBase + ct * 64;  // the tile's 64 columns (features of camera 2)
        if (j0 >= N) break;              // (uniform)
        const bool more = ct + 1 < NC_CT && j0 + 64 < N;
        if (more) nc_tile_fetch(A.s2, j0 + 64, N, tid, cr);   // in flight while this tile is worked on
        const unsigned char* tile = sTile[ct & 1];
        const double* cc = (const double*)(tile + 64 * NC_TP);
        const int* cv = (const int*)(cc + 256);
        if (rowsIn) {
            nc_i32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = (nc_i32x4){0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = 32 * ks + 8 * lk;
                long b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = *(const long*)(tile + (16 * t + lr) * NC_TP + off);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[ks], b[t], acc[t], 0, 0, 0);
            }
            // ---- epilogue: lane holds rows 4 lk .. 4 lk + 3 of column lr of every 16 x 16 tile ----
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int jl = 16 * t + lr, j = j0 + jl;
                if (j >= N) continue;
                const double bx = cc[jl], by = cc[64 + jl];
                const double A2 = cc[128 + jl], C2 = cc[192 + jl];
                const int v2 = cv[jl];
                const int s2 = (int)A2 - NC_LEN * 128;
                // epipolarError(F, p1, p2): the line of p2
                const double l0 = (Fm[0] * bx + Fm[1] * by) + Fm[2];
                const double l1 = (Fm[3] * bx + Fm[4] * by) + Fm[5];
                const double l2 = (Fm[6] * bx + Fm[7] * by) + Fm[8];
                const double nn = sqrt(l0 * l0 + l1 * l1);
                const double den = nn > 0 ? nn : 1.0;
                // |l . p1| / den <= epiMax is decided without the division wherever it is not close: a numerator beyond epiMax den (1 + 1e-12)
                // fails for certain (the margin is 10^4 roundings wide), and a pair that fails writes wNone whatever its quotient is.  Only
                // the few pairs near or inside the band pay the IEEE division (~20 instructions of the ~26 this test used to cost per pair).
                const double numMax = (A.epiMax * den) * (1.0 + 1e-12);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + 4 * lk + r;
                    if (i >= M) continue;
                    const double num = fabs((l0 * x1[r] + l1 * y1[r]) + l2);
                    double e = A.wNone, c = A.wNone;
                    bool pass = false;
                    double epiErr = 0;
                    bool near = v1[r] && v2 && !(num > numMax);   // (NaN stays in: the exact test below decides as it always did)
                    if (near) {
                        epiErr = num / den;                        // SL_FeatureMatching.cpp:24-25
                        near = epiErr <= A.epiMax;                 // :26
                    }
                    if (near) {
                        const int d = acc[t][r] + 128 * s1[r] + 128 * s2 + NC_LEN * 128 * 128;  // sum I1 I2, exact
                        const double ncc = (((double)NC_LEN * (double)d - A1[r] * A2) * C1[r]) * C2;  // SL_NCCBlock.cpp:263
                        if (ncc >= A.nccMin) {                                          // :29-31
                            e = epiErr;
                            c = ncc;
                            pass = true;
                        }
                    }
                    if (SPARSE) {
                        if (pass) {
                            const int at = atomicAdd(A.pairCount, 1);
                            if (at < A.pairCap) {
                                cs_ncc_pair q;
                                q.i = i, q.j = j, q.epi = e, q.ncc = c;
                                A.pairs[at] = q;
                            }
                        }
                    } else {
                        A.epiMat[(size_t)i * N + j] = e;
                        A.nccMat[(size_t)i * N + j] = c;
                    }
                }
            }
        }
        if (more) nc_tile_store(sTile[(ct + 1) & 1], tid, cr);
        __syncthreads();
    }
}

}  // namespace

extern "C" int cs_ncc_blocks_dev(int device, void* hip_stream, const unsigned char* d_img, int W, int H, int n, const double* d_x,
                                 const double* d_y, double scale, unsigned char* d_blocks, double* d_abc, int* d_valid) {
    if (W < NC_BW || H < NC_BW || n < 0 || !(scale > 0)) {
        cs_set_error("cs_ncc_blocks_dev: bad arguments (image at least 11 x 11, scale > 0)");
        return CS_ERR_INVALID;
    }
    if (n == 0) return CS_OK;
    if (!d_img || !d_x || !d_y || !d_blocks || !d_abc || !d_valid) {
        cs_set_error("cs_ncc_blocks_dev: null pointer");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_ncc_blocks, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream, d_img, W, H, n, d_x, d_y,
                       scale, d_blocks, d_abc, d_valid);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_ncc_unmapped_mask_dev(int device, void* hip_stream, int n, const int* d_state, const int* d_slot2map, int* d_valid) {
    if (n < 0 || (n && (!d_state || !d_slot2map || !d_valid))) {
        cs_set_error("cs_ncc_unmapped_mask_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (n == 0) return CS_OK;
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_ncc_unmapped_mask, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, n, d_state, d_slot2map, d_valid);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// cv::Size of the resized image: saturate_cast<int>(W * scale) (round half to even)
extern "C" int cs_ncc_scaled_dims(int W, int H, double scale, int* Ws, int* Hs) {
    if (!Ws || !Hs || W <= 0 || H <= 0 || !(scale > 0)) {
        cs_set_error("cs_ncc_scaled_dims: bad arguments");
        return CS_ERR_INVALID;
    }
    *Ws = (int)lrint(W * scale);
    *Hs = (int)lrint(H * scale);
    return CS_OK;
}

static int ncc_get_blocks_group_impl(int device, void* hip_stream, int nCams, const cs_ncc_cam* cams, int W, int H, int n, double scale,
                                     bool writeValid);
// getNCCBlocks(img, pts, blocks, scale) for n points, asynchronous on hip_stream; d_scaled: the caller's scratch for the resized
// image (cs_ncc_scaled_dims bytes; ignored when scale == 1.0).  Every point gets a block (the border is replicated), so there is
// no `valid` output -- d_valid, when given, is set to 1 for cs_ncc_epi_mat_dev.
extern "C" int cs_ncc_get_blocks_dev(int device, void* hip_stream, const unsigned char* d_img, int W, int H, int n, const double* d_x,
                                     const double* d_y, double scale, unsigned char* d_scaled, unsigned char* d_blocks, double* d_abc,
                                     int* d_valid) {
    if (!d_img || W <= 0 || H <= 0 || n < 0 || (n && (!d_x || !d_y || !d_blocks || !d_abc)) || !(scale > 0) ||
        (scale != 1.0 && !d_scaled)) {
        cs_set_error("cs_ncc_get_blocks_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    cs_ncc_cam one;
    memset(&one, 0, sizeof(one));
    one.img = d_img, one.x = d_x, one.y = d_y, one.scaled = d_scaled, one.blocks = d_blocks, one.abc = d_abc, one.valid = d_valid;
    return ncc_get_blocks_group_impl(device, hip_stream, 1, &one, W, H, n, scale, true);
}

// every camera of a group: ONE resize launch and ONE cutter launch (blockIdx.z / .y = camera).  cs_ncc_cam::valid is the caller's
// mask for the matrices (which features take part) and is left alone.
extern "C" int cs_ncc_get_blocks_group_dev(int device, void* hip_stream, int nCams, const cs_ncc_cam* cams, int W, int H, int n,
                                           double scale) {
    return ncc_get_blocks_group_impl(device, hip_stream, nCams, cams, W, H, n, scale, false);
}
static int ncc_get_blocks_group_impl(int device, void* hip_stream, int nCams, const cs_ncc_cam* cams, int W, int H, int n, double scale,
                                     bool writeValid) {
    if (nCams < 1 || nCams > NC_MAX_CAMS || !cams || W <= 0 || H <= 0 || n < 0 || !(scale > 0)) {
        cs_set_error("cs_ncc_get_blocks_group_dev: bad arguments (1..%d cameras)", NC_MAX_CAMS);
        return CS_ERR_INVALID;
    }
    NcCamSet S;
    memset(&S, 0, sizeof(S));
    for (int c = 0; c < nCams; ++c) {
        const cs_ncc_cam& q = cams[c];
        if (!q.img || (n && (!q.x || !q.y || !q.blocks || !q.abc)) || (scale != 1.0 && !q.scaled)) {
            cs_set_error("cs_ncc_get_blocks_group_dev: null pointer in camera %d", c);
            return CS_ERR_INVALID;
        }
        S.img[c] = q.img, S.scaled[c] = q.scaled, S.x[c] = q.x, S.y[c] = q.y, S.blocks[c] = q.blocks, S.abc[c] = q.abc, S.valid[c] = writeValid ? q.valid : nullptr;
    }
    CS_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    int Ws = W, Hs = H;
    if (scale != 1.0) {
        Ws = (int)lrint(W * scale), Hs = (int)lrint(H * scale);
        if (Ws < 1 || Hs < 1) {
            cs_set_error("cs_ncc_get_blocks_group_dev: the scaled image is empty");
            return CS_ERR_INVALID;
        }
        hipLaunchKernelGGL(k_resize_linear_u8, dim3((Ws + 63) / 64, (Hs + 3) / 4, nCams), dim3(256), 0, s, S, W, H, 1. / scale, 1. / scale, Ws, Hs);
    }
    if (n > 0) hipLaunchKernelGGL(k_ncc_blocks_subpix, dim3((n + 3) / 4, nCams), dim3(256), 0, s, S, Ws, Hs, n, scale, scale != 1.0 ? 1 : 0);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_ncc_epi_mat_dev(int device, void* hip_stream, const double F[9], int M, const double* d_x1, const double* d_y1,
                                  const unsigned char* d_blocks1, const double* d_abc1, const int* d_valid1, int N,
                                  const double* d_x2, const double* d_y2, const unsigned char* d_blocks2, const double* d_abc2,
                                  const int* d_valid2, double epiMax, double nccMin, double wNone, double* d_epiMat,
                                  double* d_nccMat) {
    if (!F || M < 0 || N < 0) {
        cs_set_error("cs_ncc_epi_mat_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (M == 0 || N == 0) return CS_OK;
    if (!d_x1 || !d_y1 || !d_blocks1 || !d_abc1 || !d_valid1 || !d_x2 || !d_y2 || !d_blocks2 || !d_abc2 || !d_valid2 || !d_epiMat ||
        !d_nccMat) {
        cs_set_error("cs_ncc_epi_mat_dev: null pointer");
        return CS_ERR_INVALID;
    }
    NcArgs A;
    memset(&A, 0, sizeof(A));   // (dF = null: the matrix by value)
    memcpy(A.F, F, sizeof(A.F));
    A.s1 = {d_x1, d_y1, d_blocks1, d_abc1, d_valid1, M};
    A.s2 = {d_x2, d_y2, d_blocks2, d_abc2, d_valid2, N};
    A.epiMax = epiMax;
    A.nccMin = nccMin;
    A.wNone = wNone;
    A.epiMat = d_epiMat;
    A.nccMat = d_nccMat;
    CS_HIP(hipSetDevice(device));
    A.pairs = nullptr, A.pairCap = 0, A.pairCount = nullptr;
    NcJobs J;
    J.job[0] = A;
    hipLaunchKernelGGL(k_ncc_epi_mat<false>, dim3((unsigned)((N + 64 * NC_CT - 1) / (64 * NC_CT)), (unsigned)((M + 63) / 64)), dim3(256), 0,
                       (hipStream_t)hip_stream, J);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_ncc_epi_pairs_dev(int device, void* hip_stream, const double F[9], int M, const double* d_x1, const double* d_y1,
                                    const unsigned char* d_blocks1, const double* d_abc1, const int* d_valid1, int N, const double* d_x2,
                                    const double* d_y2, const unsigned char* d_blocks2, const double* d_abc2, const int* d_valid2,
                                    double epiMax, double nccMin, cs_ncc_pair* d_pairs, int pairCap, int* d_pairCount) {
    if (!F || M < 0 || N < 0 || pairCap < 0 || !d_pairCount || (pairCap > 0 && !d_pairs)) {
        cs_set_error("cs_ncc_epi_pairs_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(device));
    CS_HIP(hipMemsetAsync(d_pairCount, 0, sizeof(int), (hipStream_t)hip_stream));
    if (M == 0 || N == 0) return CS_OK;
    if (!d_x1 || !d_y1 || !d_blocks1 || !d_abc1 || !d_valid1 || !d_x2 || !d_y2 || !d_blocks2 || !d_abc2 || !d_valid2) {
        cs_set_error("cs_ncc_epi_pairs_dev: null pointer");
        return CS_ERR_INVALID;
    }
    NcArgs A;
    memset(&A, 0, sizeof(A));   // (dF = null: the matrix by value)
    memcpy(A.F, F, sizeof(A.F));
    A.s1 = {d_x1, d_y1, d_blocks1, d_abc1, d_valid1, M};
    A.s2 = {d_x2, d_y2, d_blocks2, d_abc2, d_valid2, N};
    A.epiMax = epiMax;
    A.nccMin = nccMin;
    A.wNone = -1.0;
    A.epiMat = A.nccMat = nullptr;
    A.pairs = d_pairs, A.pairCap = pairCap, A.pairCount = d_pairCount;
    NcJobs J;
    J.job[0] = A;
    hipLaunchKernelGGL(k_ncc_epi_mat<true>, dim3((unsigned)((N + 64 * NC_CT - 1) / (64 * NC_CT)), (unsigned)((M + 63) / 64)), dim3(256), 0,
                       (hipStream_t)hip_stream, J);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// NewMapPtsNCC::matchBetween forms E and F of a camera pair from the cameras' CURRENT poses (src/app/SL_NewMapPointsInterCam.cpp:284-292:
// formEMat(R1, t1, R2, t2, E), getFMat(iK1, iK2, E, F)) -- the poses the frame has just solved, not anything known beforehand.  One lane
// per pair: x_A = R x_B + t with R = R_A R_B^T, t = t_A - R t_B; E = [t]x R; F = iK_A^T E iK_B (epipolarError(F, a, b) measures a in
// camera A against the line F (b, 1) of b in camera B).  formEMat / getFMat are un-vendored LibVisualSLAM: our definitions.
struct NcFmArgs {
    int n;
    int camA[NC_MAX_JOBS], camB[NC_MAX_JOBS];
    const double* iK[NC_MAX_JOBS][2];
    const double *R, *t;
    double* F;
};
__global__ void k_ncc_fmats(NcFmArgs A) {
    const int k = threadIdx.x;
    if (k >= A.n) return;
    const double *Ra = A.R + 9 * A.camA[k], *Rb = A.R + 9 * A.camB[k], *ta = A.t + 3 * A.camA[k], *tb = A.t + 3 * A.camB[k];
    double R[9], t[3], E[9], T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = (Ra[3 * i] * Rb[3 * j] + Ra[3 * i + 1] * Rb[3 * j + 1]) + Ra[3 * i + 2] * Rb[3 * j + 2];
    for (int i = 0; i < 3; ++i) t[i] = ta[i] - ((R[3 * i] * tb[0] + R[3 * i + 1] * tb[1]) + R[3 * i + 2] * tb[2]);
    const double X[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) E[3 * i + j] = (X[3 * i] * R[j] + X[3 * i + 1] * R[3 + j]) + X[3 * i + 2] * R[6 + j];
    const double *Ka = A.iK[k][0], *Kb = A.iK[k][1];
    for (int i = 0; i < 3; ++i)   // T = iK_A^T E
        for (int j = 0; j < 3; ++j) T[3 * i + j] = (Ka[i] * E[j] + Ka[3 + i] * E[3 + j]) + Ka[6 + i] * E[6 + j];
    for (int i = 0; i < 3; ++i)   // F = T iK_B
        for (int j = 0; j < 3; ++j) A.F[9 * k + 3 * i + j] = (T[3 * i] * Kb[j] + T[3 * i + 1] * Kb[3 + j]) + T[3 * i + 2] * Kb[6 + j];
}
// d_F [nPairs][9] <- the fundamental matrices of the pairs (camA[k], camB[k]) from the cameras' poses d_R [nCams][9], d_t [nCams][3] and
// inverse intrinsics (d_iK: nCams pointers to 9 doubles each, host array); hand d_F + 9 k to job k of cs_ncc_epi_pairs_group_dev (dF)
extern "C" int cs_ncc_fmats_dev(int device, void* hip_stream, int nCams, int nPairs, const int* camA, const int* camB, const double* const* d_iK,
                                const double* d_R, const double* d_t, double* d_F) {
    if (nCams < 1 || nPairs < 0 || nPairs > NC_MAX_JOBS || (nPairs && (!camA || !camB || !d_iK || !d_R || !d_t || !d_F))) {
        cs_set_error("cs_ncc_fmats_dev: bad arguments (<= %d pairs)", NC_MAX_JOBS);
        return CS_ERR_INVALID;
    }
    if (nPairs == 0) return CS_OK;
    NcFmArgs A;
    memset(&A, 0, sizeof(A));
    A.n = nPairs, A.R = d_R, A.t = d_t, A.F = d_F;
    for (int k = 0; k < nPairs; ++k) {
        if (camA[k] < 0 || camA[k] >= nCams || camB[k] < 0 || camB[k] >= nCams || !d_iK[camA[k]] || !d_iK[camB[k]]) {
            cs_set_error("cs_ncc_fmats_dev: bad pair %d", k);
            return CS_ERR_INVALID;
        }
        A.camA[k] = camA[k], A.camB[k] = camB[k], A.iK[k][0] = d_iK[camA[k]], A.iK[k][1] = d_iK[camB[k]];
    }
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_ncc_fmats, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, A);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// the camera pairs of a matching run in ONE launch (blockIdx.z = pair): jobs[k] = {F, camA, camB, pairs, count}; the cameras'
// blocks / abc / valid / positions as cs_ncc_get_blocks_group_dev left them (n features each)
extern "C" int cs_ncc_epi_pairs_group_dev(int device, void* hip_stream, int nCams, const cs_ncc_cam* cams, int n, int nJobs,
                                          const cs_ncc_pair_job* jobs, double epiMax, double nccMin, int pairCap) {
    if (nCams < 1 || nCams > NC_MAX_CAMS || !cams || n < 0 || nJobs < 0 || nJobs > NC_MAX_JOBS || (nJobs && !jobs) || pairCap < 0) {
        cs_set_error("cs_ncc_epi_pairs_group_dev: bad arguments (<= %d camera pairs per call)", NC_MAX_JOBS);
        return CS_ERR_INVALID;
    }
    if (nJobs == 0) return CS_OK;
    CS_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    NcJobs J;
    memset(&J, 0, sizeof(J));
    cs_small::List zero;   // the jobs' pair counts: one launch (small_ops.h), not a memset each
    for (int k = 0; k < nJobs; ++k) {
        const cs_ncc_pair_job& q = jobs[k];
        if (q.camA < 0 || q.camA >= nCams || q.camB < 0 || q.camB >= nCams || !q.count || (pairCap > 0 && !q.pairs)) {
            cs_set_error("cs_ncc_epi_pairs_group_dev: bad job %d", k);
            return CS_ERR_INVALID;
        }
        const cs_ncc_cam &a = cams[q.camA], &b = cams[q.camB];
        if (n && (!a.x || !a.y || !a.blocks || !a.abc || !a.valid || !b.x || !b.y || !b.blocks || !b.abc || !b.valid)) {
            cs_set_error("cs_ncc_epi_pairs_group_dev: null pointer in a camera of job %d", k);
            return CS_ERR_INVALID;
        }
        NcArgs& A = J.job[k];
        memcpy(A.F, q.F, sizeof(A.F));
        A.dF = q.dF;
        A.s1 = {a.x, a.y, a.blocks, a.abc, a.valid, n};
        A.s2 = {b.x, b.y, b.blocks, b.abc, b.valid, n};
        A.epiMax = epiMax, A.nccMin = nccMin, A.wNone = -1.0;
        A.epiMat = A.nccMat = nullptr;
        A.pairs = q.pairs, A.pairCap = pairCap, A.pairCount = q.count;
        zero.fill(q.count, 0, sizeof(int));
    }
    CS_HIP(zero.run(s));
    if (n == 0) return CS_OK;
    hipLaunchKernelGGL(k_ncc_epi_mat<true>, dim3((unsigned)((n + 64 * NC_CT - 1) / (64 * NC_CT)), (unsigned)((n + 63) / 64), (unsigned)nJobs), dim3(256), 0, s, J);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// Host-pointer form of the whole stage for one camera pair, as NewMapPtsNCC::matchBetween calls it: blocks of both
// cameras from their small images, then the two matrices.  One upload, three launches, one read-back.
// cutter 0: NCCBlock::computeScaled on the SMALL image passed in (SL_NCCBlock.cpp:15-54); 1: getNCCBlocks on the FULL image
// passed in (resize by `scale`, sub-pixel patch: SL_NCCBlock.cpp:79-155 -- what matchBetween itself calls)
static int ncc_match_between_impl(int cutter, int device, const unsigned char* img1, int W1, int H1, int M, const double* x1,
                                  const double* y1, const unsigned char* img2, int W2, int H2, int N, const double* x2, const double* y2,
                                  double scale, const double F[9], double epiMax, double nccMin, double wNone, double* epiMat,
                                  double* nccMat, unsigned char* blocks1, double* abc1, int* valid1, unsigned char* blocks2,
                                  double* abc2, int* valid2) {
    if (!img1 || !img2 || !F || M < 0 || N < 0 || (M && (!x1 || !y1)) || (N && (!x2 || !y2)) || (M && N && (!epiMat || !nccMat))) {
        cs_set_error("cs_ncc_match_between: bad arguments");
        return CS_ERR_INVALID;
    }
    if (M == 0 || N == 0) return CS_OK;
    CS_HIP(hipSetDevice(device));
    const size_t i1 = (size_t)W1 * H1, i2 = (size_t)W2 * H2;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    const size_t oI1 = off; off += al(i1);
    const size_t oI2 = off; off += al(i2);
    const size_t oX1 = off; off += al(8 * (size_t)M);
    const size_t oY1 = off; off += al(8 * (size_t)M);
    const size_t oX2 = off; off += al(8 * (size_t)N);
    const size_t oY2 = off; off += al(8 * (size_t)N);
    const size_t inBytes = off;
    const size_t oB1 = off; off += al(128 * (size_t)M);
    const size_t oB2 = off; off += al(128 * (size_t)N);
    const size_t oC1 = off; off += al(32 * (size_t)M);
    const size_t oC2 = off; off += al(32 * (size_t)N);
    const size_t oV1 = off; off += al(4 * (size_t)M);
    const size_t oV2 = off; off += al(4 * (size_t)N);
    const size_t oE = off; off += al(8 * (size_t)M * N);
    const size_t oN = off; off += al(8 * (size_t)M * N);
    const size_t oS1 = off; off += al(cutter ? i1 : 0);   // (scratch of the resized images: never larger than the originals
    const size_t oS2 = off; off += al(cutter ? i2 : 0);   //  for scale <= 1; larger scales are refused below)
    char* d = nullptr;
    if (hipMalloc((void**)&d, off) != hipSuccess) {
        cs_set_error("cs_ncc_match_between: out of device memory (%zu bytes)", off);
        return CS_ERR_HIP;
    }
    hipStream_t s = nullptr;
    int rc = CS_OK;
    hipError_t e = hipMemcpyAsync(d + oI1, img1, i1, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + oI2, img2, i2, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + oX1, x1, 8 * (size_t)M, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + oY1, y1, 8 * (size_t)M, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + oX2, x2, 8 * (size_t)N, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + oY2, y2, 8 * (size_t)N, hipMemcpyHostToDevice, s);
    (void)inBytes;
    if (cutter && scale > 1.0) {
        (void)hipFree(d);
        cs_set_error("cs_ncc_match_between_full: scale > 1 is not supported");
        return CS_ERR_INVALID;
    }
    if (e == hipSuccess && cutter) {
        rc = cs_ncc_get_blocks_dev(device, s, (const unsigned char*)(d + oI1), W1, H1, M, (const double*)(d + oX1), (const double*)(d + oY1),
                                   scale, (unsigned char*)(d + oS1), (unsigned char*)(d + oB1), (double*)(d + oC1), (int*)(d + oV1));
        if (rc == CS_OK)
            rc = cs_ncc_get_blocks_dev(device, s, (const unsigned char*)(d + oI2), W2, H2, N, (const double*)(d + oX2),
                                       (const double*)(d + oY2), scale, (unsigned char*)(d + oS2), (unsigned char*)(d + oB2),
                                       (double*)(d + oC2), (int*)(d + oV2));
        if (rc == CS_OK)
            rc = cs_ncc_epi_mat_dev(device, s, F, M, (const double*)(d + oX1), (const double*)(d + oY1), (const unsigned char*)(d + oB1),
                                    (const double*)(d + oC1), (const int*)(d + oV1), N, (const double*)(d + oX2),
                                    (const double*)(d + oY2), (const unsigned char*)(d + oB2), (const double*)(d + oC2),
                                    (const int*)(d + oV2), epiMax, nccMin, wNone, (double*)(d + oE), (double*)(d + oN));
    } else if (e == hipSuccess) {
        rc = cs_ncc_blocks_dev(device, s, (const unsigned char*)(d + oI1), W1, H1, M, (const double*)(d + oX1), (const double*)(d + oY1),
                               scale, (unsigned char*)(d + oB1), (double*)(d + oC1), (int*)(d + oV1));
        if (rc == CS_OK)
            rc = cs_ncc_blocks_dev(device, s, (const unsigned char*)(d + oI2), W2, H2, N, (const double*)(d + oX2),
                                   (const double*)(d + oY2), scale, (unsigned char*)(d + oB2), (double*)(d + oC2), (int*)(d + oV2));
        if (rc == CS_OK)
            rc = cs_ncc_epi_mat_dev(device, s, F, M, (const double*)(d + oX1), (const double*)(d + oY1), (const unsigned char*)(d + oB1),
                                    (const double*)(d + oC1), (const int*)(d + oV1), N, (const double*)(d + oX2),
                                    (const double*)(d + oY2), (const unsigned char*)(d + oB2), (const double*)(d + oC2),
                                    (const int*)(d + oV2), epiMax, nccMin, wNone, (double*)(d + oE), (double*)(d + oN));
    }
    if (rc == CS_OK && e == hipSuccess) e = hipMemcpyAsync(epiMat, d + oE, 8 * (size_t)M * N, hipMemcpyDeviceToHost, s);
    if (rc == CS_OK && e == hipSuccess) e = hipMemcpyAsync(nccMat, d + oN, 8 * (size_t)M * N, hipMemcpyDeviceToHost, s);
    auto back = [&](void* dst, size_t o, size_t bytes) {
        if (dst && rc == CS_OK && e == hipSuccess) e = hipMemcpyAsync(dst, d + o, bytes, hipMemcpyDeviceToHost, s);
    };
    back(blocks1, oB1, 128 * (size_t)M);
    back(blocks2, oB2, 128 * (size_t)N);
    back(abc1, oC1, 32 * (size_t)M);
    back(abc2, oC2, 32 * (size_t)N);
    back(valid1, oV1, 4 * (size_t)M);
    back(valid2, oV2, 4 * (size_t)N);
    if (rc == CS_OK && e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (rc != CS_OK) return rc;
    if (e != hipSuccess) {
        cs_set_error("cs_ncc_match_between: %s", hipGetErrorString(e));
        return CS_ERR_HIP;
    }
    return CS_OK;
}

extern "C" int cs_ncc_match_between(int device, const unsigned char* img1, int W1, int H1, int M, const double* x1, const double* y1,
                                    const unsigned char* img2, int W2, int H2, int N, const double* x2, const double* y2,
                                    double scale, const double F[9], double epiMax, double nccMin, double wNone, double* epiMat,
                                    double* nccMat, unsigned char* blocks1, double* abc1, int* valid1, unsigned char* blocks2,
                                    double* abc2, int* valid2) {
    return ncc_match_between_impl(0, device, img1, W1, H1, M, x1, y1, img2, W2, H2, N, x2, y2, scale, F, epiMax, nccMin, wNone, epiMat,
                                  nccMat, blocks1, abc1, valid1, blocks2, abc2, valid2);
}

// NewMapPtsNCC::matchBetween's own data path (src/app/SL_NewMapPointsInterCam.cpp:273-290): the FULL images, getNCCBlocks with
// blockScale, then the two matrices.
extern "C" int cs_ncc_match_between_full(int device, const unsigned char* img1, int W1, int H1, int M, const double* x1, const double* y1,
                                         const unsigned char* img2, int W2, int H2, int N, const double* x2, const double* y2,
                                         double scale, const double F[9], double epiMax, double nccMin, double wNone, double* epiMat,
                                         double* nccMat, unsigned char* blocks1, double* abc1, unsigned char* blocks2, double* abc2) {
    return ncc_match_between_impl(1, device, img1, W1, H1, M, x1, y1, img2, W2, H2, N, x2, y2, scale, F, epiMax, nccMin, wNone, epiMat,
                                  nccMat, blocks1, abc1, nullptr, blocks2, abc2, nullptr);
}
