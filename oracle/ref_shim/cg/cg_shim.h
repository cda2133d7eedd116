/*
 * oracle/ref_shim/cg/cg_shim.h -- the part of the Cg language and of the GL texture unit that the reference's KLT fragment
 * programs (src/tracking/CGKLT/Shaders/ *.cg) use, for clang++ on the host.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/klt_oracle.h).  oracle/Makefile pipes each .cg file where it lies under
 * /root/reference through `sed` (binding semantics `: TEXUNITn / TEXCOORDn / COLOR` stripped, `main` kept but wrapped in a
 * namespace per program, `out T x` -> `T& x`, the Cg-only scalar swizzle `(0).xxx` -> `float3(0)`, the unsuffixed literal
 * `0.00001` -> `0.00001f` because Cg's literals are binary32) into this compiler with this header force-included.  No copy of a
 * shader is written anywhere; the objects land in oracle/_ref/.  The shader BODIES -- every sum, product, comparison and
 * their order -- are therefore the reference's own text.  What is OURS and stated here:
 *
 *   - arithmetic model: IEEE binary32, no contraction (-ffp-contract=off), expressions evaluated as written;
 *     dot(a,b) = ((a.x*b.x + a.y*b.y) + a.z*b.z) + a.w*b.w, length = sqrt(dot), 1.0f/x and sqrt correctly rounded
 *     (a GeForce of the Cg era rounds MAD / RCP / RSQ differently; nobody can run one here);
 *   - texture unit (GL 2.1 section 3.8): u = s * W_level; NEAREST picks texel floor(u), LINEAR blends the four texels around
 *     u - 0.5 with binary32 weights, ((w00*p00 + w10*p10) + w01*p01) + w11*p11; CLAMP_TO_EDGE on indices.  The GL pipeline
 *     defines the interpolated coordinate as an exact rational; binary32 round-off of OUR coordinate arithmetic is absorbed
 *     by a 1/256-texel bias inside floor() for NEAREST (taps that land exactly on a texel edge -- the 2x decimation,
 *     v3d_gpupyramid.cpp:407-418, and the gain tracker's neighbour taps on a 2:1 feature grid -- resolve to the upper texel,
 *     as floor() of the exact value does);
 *   - LUMINANCE8 texel k reads as (float)k / 255.0f; RGBA8 stores round(c * 255) and reads b / 255.0f, which makes
 *     unpack_4ubyte / pack_4ubyte a bit-exact binary32 transport; pack_2half / unpack_2half through RGBA16F likewise (NaN
 *     payloads kept, subnormals kept);
 *   - RGB16F render targets round binary32 to binary16 to nearest even, subnormals kept.
 */
#ifndef COSLAM_CG_SHIM_H
#define COSLAM_CG_SHIM_H

#include <cmath>
#include <cstdint>
#include <cstring>

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int cg_int2 __attribute__((ext_vector_type(2)));
typedef int cg_int3 __attribute__((ext_vector_type(3)));
typedef int cg_int4 __attribute__((ext_vector_type(4)));
/* `half` variables in these programs only ever hold RGBA16F texels, which binary32 represents exactly */
typedef float2 half2;
typedef float4 half4;

/* ---- binary16 <-> binary32, bit-transparent (NaN payload and subnormals kept) ---- */
static inline float cg_h2f(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = s;
        else {
            int sh = 0;
            while (!(m & 0x400u)) m <<= 1, ++sh;
            x = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) x = s | 0x7f800000u | (m << 13);
    else x = s | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}
static inline uint16_t cg_f2h(float f) { /* round to nearest even */
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t s = (x >> 16) & 0x8000u, e = (x >> 23) & 0xffu, m = x & 0x7fffffu;
    if (e == 255) return (uint16_t)(s | 0x7c00u | (m ? ((m >> 13) ? (m >> 13) : 1u) : 0u));
    int E = (int)e - 127 + 15;
    if (E >= 31) return (uint16_t)(s | 0x7c00u);
    if (E <= 0) {
        if (E < -10) return (uint16_t)s;
        m |= 0x800000u;
        int sh = 14 - E;
        uint32_t r = m >> sh, rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(s | r);
    }
    uint32_t r = ((uint32_t)E << 10) | (m >> 13), rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(s | r);
}

/* ---- textures ---- */
enum { CG_L8 = 0, CG_RGBA16F = 1, CG_RGB32F = 2, CG_RGBA8 = 3, CG_RGBA32F = 4 };
enum { CG_NEAREST = 0, CG_LINEAR = 1 };
struct cg_level {
    const void* data;
    int w, h;
};
struct cg_sampler {
    int fmt, filter, base_level, n_levels;
    cg_level lv[16];
};
typedef const cg_sampler& sampler2D;

static inline int cg_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float4 cg_texel(const cg_sampler& s, int level, int i, int j) {
    const cg_level& L = s.lv[level];
    i = cg_clampi(i, 0, L.w - 1), j = cg_clampi(j, 0, L.h - 1);
    size_t k = (size_t)j * L.w + i;
    float4 r;
    switch (s.fmt) {
    case CG_L8: {
        float l = (float)((const uint8_t*)L.data)[k] / 255.0f;
        r.x = l, r.y = l, r.z = l, r.w = 1.0f;
    } break;
    case CG_RGBA16F: {
        const uint16_t* p = (const uint16_t*)L.data + 4 * k;
        r.x = cg_h2f(p[0]), r.y = cg_h2f(p[1]), r.z = cg_h2f(p[2]), r.w = cg_h2f(p[3]);
    } break;
    case CG_RGB32F: {
        const float* p = (const float*)L.data + 3 * k;
        r.x = p[0], r.y = p[1], r.z = p[2], r.w = 1.0f;
    } break;
    case CG_RGBA8: {
        const uint8_t* p = (const uint8_t*)L.data + 4 * k;
        r.x = (float)p[0] / 255.0f, r.y = (float)p[1] / 255.0f, r.z = (float)p[2] / 255.0f, r.w = (float)p[3] / 255.0f;
    } break;
    default: {
        const float* p = (const float*)L.data + 4 * k;
        r.x = p[0], r.y = p[1], r.z = p[2], r.w = p[3];
    }
    }
    return r;
}

/* what tex2D returns: a float4 that Cg narrows silently (`float g = tex2D(..)`, `float3 v = tex2D(..)`) */
struct cg_fetch {
    float x, y, z, w;
    float2 xy, yz, zw;
    float3 xyz;
    explicit cg_fetch(float4 v) : x(v.x), y(v.y), z(v.z), w(v.w) {
        xy.x = v.x, xy.y = v.y, yz.x = v.y, yz.y = v.z, zw.x = v.z, zw.y = v.w;
        xyz.x = v.x, xyz.y = v.y, xyz.z = v.z;
    }
    operator float() const { return x; }
    operator float3() const { return xyz; }
    operator float4() const {
        float4 r;
        r.x = x, r.y = y, r.z = z, r.w = w;
        return r;
    }
};

static inline float4 cg_sample_level(const cg_sampler& s, int level, float cs, float ct) {
    level = cg_clampi(level, 0, s.n_levels - 1);
    const cg_level& L = s.lv[level];
    if (s.filter == CG_NEAREST) {
        float u = cs * (float)L.w + 0.00390625f, v = ct * (float)L.h + 0.00390625f;
        u = fminf(fmaxf(u, -2.0f), (float)L.w + 1.0f), v = fminf(fmaxf(v, -2.0f), (float)L.h + 1.0f);
        return cg_texel(s, level, (int)floorf(u), (int)floorf(v));
    }
    float u = cs * (float)L.w - 0.5f, v = ct * (float)L.h - 0.5f;
    u = fminf(fmaxf(u, -2.0f), (float)L.w + 1.0f), v = fminf(fmaxf(v, -2.0f), (float)L.h + 1.0f);
    float fu = floorf(u), fv = floorf(v), a = u - fu, b = v - fv;
    int i0 = (int)fu, j0 = (int)fv;
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float4 p00 = cg_texel(s, level, i0, j0), p10 = cg_texel(s, level, i0 + 1, j0), p01 = cg_texel(s, level, i0, j0 + 1),
           p11 = cg_texel(s, level, i0 + 1, j0 + 1);
    return ((w00 * p00 + w10 * p10) + w01 * p01) + w11 * p11;
}
static inline cg_fetch tex2D(sampler2D s, float2 st) { return cg_fetch(cg_sample_level(s, s.base_level, st.x, st.y)); }
static inline cg_fetch tex2Dlod(sampler2D s, float4 st) {
    return cg_fetch(cg_sample_level(s, s.base_level + (int)floorf(st.w + 0.5f), st.x, st.y));
}

/* ---- constructors: Cg's float3(a, b.xy) forms, behind function-like macros so that declarations stay types ---- */
static inline float2 cg_mk2(float a) { float2 r; r.x = a, r.y = a; return r; }
static inline float2 cg_mk2(float a, float b) { float2 r; r.x = a, r.y = b; return r; }
static inline float3 cg_mk3(float a) { float3 r; r.x = a, r.y = a, r.z = a; return r; }
static inline float3 cg_mk3(float a, float b, float c) { float3 r; r.x = a, r.y = b, r.z = c; return r; }
static inline float3 cg_mk3(float2 a, float b) { float3 r; r.x = a.x, r.y = a.y, r.z = b; return r; }
static inline float3 cg_mk3(float a, float2 b) { float3 r; r.x = a, r.y = b.x, r.z = b.y; return r; }
static inline float4 cg_mk4(float a) { float4 r; r.x = a, r.y = a, r.z = a, r.w = a; return r; }
static inline float4 cg_mk4(float a, float b, float c, float d) { float4 r; r.x = a, r.y = b, r.z = c, r.w = d; return r; }
static inline float4 cg_mk4(float2 a, float b, float c) { float4 r; r.x = a.x, r.y = a.y, r.z = b, r.w = c; return r; }
static inline float4 cg_mk4(float2 a, float2 b) { float4 r; r.x = a.x, r.y = a.y, r.z = b.x, r.w = b.y; return r; }
static inline float4 cg_mk4(float3 a, float b) { float4 r; r.x = a.x, r.y = a.y, r.z = a.z, r.w = b; return r; }
static inline float4 cg_mk4(float a, float3 b) { float4 r; r.x = a, r.y = b.x, r.z = b.y, r.w = b.z; return r; }

/* ---- standard library ---- */
static inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float dot(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
static inline float length(float2 a) { return sqrtf(dot(a, a)); }
static inline float abs(float a) { return fabsf(a); }
static inline float sqrt(float a) { return sqrtf(a); }
static inline float max(float a, float b) { return a > b ? a : b; }
/* `float c = max(c - t, float4(0))` (klt_detector_pass2.cg:28): the scalar is smeared, the float4 result narrowed to .x */
static inline float max(float a, float4 b) { return a > b.x ? a : b.x; }
static inline bool all(cg_int2 c) { return c.x && c.y; }
static inline bool any(cg_int2 c) { return c.x || c.y; }

/* unpack_2half(a): the 32 bits of a as two binary16 values; pack_2half: back (Cg standard library) */
static inline float2 unpack_2half(float a) {
    uint32_t x;
    memcpy(&x, &a, 4);
    float2 r;
    r.x = cg_h2f((uint16_t)(x & 0xffffu)), r.y = cg_h2f((uint16_t)(x >> 16));
    return r;
}
static inline float pack_2half(float2 a) {
    uint32_t x = (uint32_t)cg_f2h(a.x) | ((uint32_t)cg_f2h(a.y) << 16);
    float f;
    memcpy(&f, &x, 4);
    return f;
}
static inline uint8_t cg_unorm8(float c) {
    float v = c * 255.0f + 0.5f;
    return (uint8_t)(v < 0.0f ? 0 : (v > 255.0f ? 255 : (int)v));
}
static inline float4 unpack_4ubyte(float a) {
    uint8_t b[4];
    memcpy(b, &a, 4);
    float4 r;
    r.x = (float)b[0] / 255.0f, r.y = (float)b[1] / 255.0f, r.z = (float)b[2] / 255.0f, r.w = (float)b[3] / 255.0f;
    return r;
}
static inline float pack_4ubyte(float4 a) {
    uint8_t b[4] = {cg_unorm8(a.x), cg_unorm8(a.y), cg_unorm8(a.z), cg_unorm8(a.w)};
    float f;
    memcpy(&f, b, 4);
    return f;
}

/* `discard`: the fragment keeps what the target held; the rasteriser below reads the flag after main() returns */
extern thread_local bool cg_discarded;
#define discard                  \
    do {                         \
        cg_discarded = true;     \
        return;                  \
    } while (0)

#define uniform
#define float2(...) cg_mk2(__VA_ARGS__)
#define float3(...) cg_mk3(__VA_ARGS__)
#define float4(...) cg_mk4(__VA_ARGS__)

#endif
