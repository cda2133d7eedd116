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

#endif  // __HIPCC__
