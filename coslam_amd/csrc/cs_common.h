// cs_common.h -- shared host/device helpers of libcoslam_hip (gfx950 only).
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/coslam_hip.h"

#define CS_MAX_LEVELS 12
#define CS_WAVE 64

// ---- error plumbing -------------------------------------------------------------------------
void cs_set_error(const char* fmt, ...);
// test / diagnostic switches (cs_debug_set in the C-ABI; nothing reads the environment): -1 = not set
enum { CS_DBG_BA_SYRK = 0, CS_DBG_BA_PACKED = 1, CS_DBG_BA_GRAPHS = 2, CS_DBG_MERGE_PRINT = 3, CS_DBG_COUNT = 4 };
int cs_debug_get(int which);

#define CS_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            cs_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return CS_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define CS_CHECK_LAUNCH()                                                                        \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) {                                                                  \
            cs_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return CS_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

// ---- pyramid layout in HBM ----------------------------------------------------------------
// One allocation per pyramid.  Level l is (W>>l) x (H>>l) texels, row-major, each texel 8 bytes:
// four binary16 (I, Ix, Iy, 0) so that one 8-byte load returns the three channels the tracker and
// the detector consume together.  Level starts are aligned to 64 texels (512 B).
struct CsPyrLayout {
    int W, H, L;
    int w[CS_MAX_LEVELS], h[CS_MAX_LEVELS];
    int64_t off[CS_MAX_LEVELS];  // in texels
    size_t texels;
};

static inline CsPyrLayout cs_make_layout(int W, int H, int L) {
    CsPyrLayout p;
    memset(&p, 0, sizeof(p));
    p.W = W;
    p.H = H;
    p.L = L;
    size_t total = 0;
    for (int l = 0; l < L; ++l) {
        p.w[l] = W >> l;
        p.h[l] = H >> l;
        p.off[l] = (int64_t)total;
        size_t n = (size_t)p.w[l] * (size_t)p.h[l];
        total += (n + 63) & ~(size_t)63;
    }
    p.texels = total;
    return p;
}

// ---- device helpers -------------------------------------------------------------------------
#ifdef __HIPCC__

typedef uint2 cs_texel;  // 4 x binary16: x = I | Ix<<16, y = Iy | 0<<16

__device__ __forceinline__ int cs_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ float cs_h2f(unsigned short h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ unsigned short cs_f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }

__device__ __forceinline__ cs_texel cs_pack_texel(float I, float Ix, float Iy) {
    cs_texel t;
    t.x = (unsigned)cs_f2h(I) | ((unsigned)cs_f2h(Ix) << 16);
    t.y = (unsigned)cs_f2h(Iy);
    return t;
}
__device__ __forceinline__ void cs_unpack_texel(cs_texel t, float& I, float& Ix, float& Iy) {
    I = cs_h2f((unsigned short)(t.x & 0xffffu));
    Ix = cs_h2f((unsigned short)(t.x >> 16));
    Iy = cs_h2f((unsigned short)(t.y & 0xffffu));
}

// ---- wave64 reductions on the DPP data path (no LDS crossbar, no ds_bpermute) -----------------------
// quad_perm xor1, quad_perm xor2, row_half_mirror, row_mirror fold each 16-lane row; row_bcast15 / row_bcast31
// (gfx9 DPP controls 0x142 / 0x143) carry the row totals forward so that lane 63 holds the wave total, which
// v_readlane then broadcasts through an SGPR: the result is wave-uniform and the order of additions is fixed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int cs_dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float cs_dpp_f(float v) {
    return __int_as_float(cs_dpp_i<CTRL, ROW_MASK>(__float_as_int(v)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double cs_dpp_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = cs_dpp_i<CTRL, ROW_MASK>(lo);
    hi = cs_dpp_i<CTRL, ROW_MASK>(hi);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float cs_wave_sum(float v) {
    v += cs_dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += cs_dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += cs_dpp_f<0x141, 0xf>(v);  // row_half_mirror
    v += cs_dpp_f<0x140, 0xf>(v);  // row_mirror
    v += cs_dpp_f<0x142, 0xa>(v);  // row_bcast15 -> rows 1,3
    v += cs_dpp_f<0x143, 0xc>(v);  // row_bcast31 -> rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Four wave sums at once.  The 16-lane rows fold as above; the two cross-row steps use the gfx950 lane-swap
// instructions on PAIRS of values (v_permlane16_swap exchanges the odd rows of one register with the even rows of
// the other, v_permlane32_swap the upper half of one with the lower half of the other), so each step is one swap
// and one add for two values instead of a masked move, an add and a re-zero for each.  The addition tree is the
// same as cs_wave_sum's -- (r0 + r1) + (r2 + r3) -- hence the same bits.
__device__ __forceinline__ float cs_row_sum(float v) {
    v += cs_dpp_f<0xB1, 0xf>(v);
    v += cs_dpp_f<0x4E, 0xf>(v);
    v += cs_dpp_f<0x141, 0xf>(v);
    v += cs_dpp_f<0x140, 0xf>(v);
    return v;
}
__device__ __forceinline__ float cs_swap16_add(float x, float y) {  // rows: [x01, y01, x23, y23]
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
__device__ __forceinline__ float cs_swap32_add(float x, float y) {  // rows: [x0 + x2, x1 + x3, y0 + y2, y1 + y3]
    const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
__device__ __forceinline__ void cs_wave_sum4(float& a, float& b, float& c, float& d) {
    const float ab = cs_swap16_add(cs_row_sum(a), cs_row_sum(b));
    const float cd = cs_swap16_add(cs_row_sum(c), cs_row_sum(d));
    const int t = __float_as_int(cs_swap32_add(ab, cd));  // rows 0..3 hold the totals of a, b, c, d
    a = __int_as_float(__builtin_amdgcn_readlane(t, 0));
    b = __int_as_float(__builtin_amdgcn_readlane(t, 16));
    c = __int_as_float(__builtin_amdgcn_readlane(t, 32));
    d = __int_as_float(__builtin_amdgcn_readlane(t, 48));
}

__device__ __forceinline__ double cs_wave_sum_d(double v) {
    v += cs_dpp_d<0xB1, 0xf>(v);
    v += cs_dpp_d<0x4E, 0xf>(v);
    v += cs_dpp_d<0x141, 0xf>(v);
    v += cs_dpp_d<0x140, 0xf>(v);
    v += cs_dpp_d<0x142, 0xa>(v);
    v += cs_dpp_d<0x143, 0xc>(v);
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int cs_wave_sum_i(int v) {
    v += cs_dpp_i<0xB1, 0xf>(v);
    v += cs_dpp_i<0x4E, 0xf>(v);
    v += cs_dpp_i<0x141, 0xf>(v);
    v += cs_dpp_i<0x140, 0xf>(v);
    v += cs_dpp_i<0x142, 0xa>(v);
    v += cs_dpp_i<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}

// ---- many sums at once: transposed butterfly ----------------------------------------------------------------------
// Folding N per-lane values with N independent 64-lane butterflies costs 6 N exchange steps; here every exchange step
// halves the number of values a lane still carries (the lower lane of a pair keeps the first half of the list, the
// upper lane the second half), so N values need N/2 + N/4 + ... ~ N exchanges in total and each total ends up in ONE
// lane: value q in lane cs_reduce_owner<N>(q).  Fixed tree, deterministic.
__device__ __forceinline__ double cs_shfl_xor_d(double v, int d) {
    int lo = __shfl_xor(__double2loint(v), d, 64), hi = __shfl_xor(__double2hiint(v), d, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cs_readlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int M, int DIST>
__device__ __forceinline__ void cs_reduce_step(double* v, int lane) {
    if constexpr (DIST >= 1) {
        constexpr int H = (M + 1) / 2;
        const bool up = (lane & DIST) != 0;
#pragma unroll
        for (int t = 0; t < H; ++t) {
            const double lo = v[t];
            const double hi = (H + t < M) ? v[H + t] : 0.0;
            const double recv = cs_shfl_xor_d(up ? lo : hi, DIST);
            v[t] = (up ? hi : lo) + recv;
        }
        cs_reduce_step<H, DIST / 2>(v, lane);
    }
}
// after the call v[0] of lane l holds the total of value cs_reduce_index<N>(l) (or -1: the lane holds nothing)
template <int N>
__device__ __forceinline__ void cs_reduce_many(double* v, int lane) {
    cs_reduce_step<N, 32>(v, lane);
}
template <int N>
__host__ __device__ constexpr int cs_reduce_index(int lane) {
    int lo = 0, end = N, m = N;  // the lane's slot list covers [lo, lo + m); indices >= end are zero padding
    for (int d = 32; d >= 1; d >>= 1) {
        const int h = (m + 1) / 2;
        if (lane & d) {
            lo += h;
        } else {
            end = end < lo + h ? end : lo + h;
        }
        m = h;
    }
    return lo < end ? lo : -1;
}
template <int N>
__host__ __device__ constexpr int cs_reduce_owner(int q) {  // the lane that ends up with the total of value q
    for (int l = 0; l < 64; ++l)
        if (cs_reduce_index<N>(l) == q) return l;
    return 0;
}
// all N totals in every lane: transposed butterfly, then one v_readlane pair per value from its (constant) owner lane
template <int N>
struct CsOwnerTable {
    int lane[N];
    constexpr CsOwnerTable() : lane{} {
        for (int q = 0; q < N; ++q) lane[q] = cs_reduce_owner<N>(q);
    }
};
template <int N, int Q>
__device__ __forceinline__ void cs_bcast_totals(double tot, double (&v)[N]) {
    if constexpr (Q < N) {
        constexpr CsOwnerTable<N> T{};
        v[Q] = cs_readlane_d(tot, T.lane[Q]);  // compile-time lane: a plain v_readlane pair
        cs_bcast_totals<N, Q + 1>(tot, v);
    }
}
template <int N>
__device__ __forceinline__ void cs_wave_sum_many_d(double (&v)[N]) {
    const int lane = threadIdx.x & 63;
    cs_reduce_many<N>(v, lane);
    const double tot = v[0];
    cs_bcast_totals<N, 0>(tot, v);
}

// The pose stream's kernels are a chain of short, latency-bound launches that share every SIMD with the persistent tracker's
// two resident waves: their waves ask for the highest issue priority (s_setprio 3; the tracker has slack -- its frame is done long
// before the pose stream wants the next one -- and runs at the default 0).  A/B in profiles/r04_ab_runs.txt.
#define CS_POSE_STREAM_PRIO() __builtin_amdgcn_s_setprio(3)
// (the same on the decision's, the NCC leg's and the refinement's kernels: 2183 against 2169 frames/s, three alternating runs each -- noise; not kept)

#endif  // __HIPCC__
