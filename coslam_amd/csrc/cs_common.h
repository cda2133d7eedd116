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

// butterfly all-reduce over the 64 lanes of a wave
__device__ __forceinline__ float cs_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

#endif  // __HIPCC__
