/*
 * oracle/ref_shim/cg/cgklt_driver.cpp -- a fragment rasteriser for the reference's KLT shaders, compiled from their own text.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/Makefile links this file with the objects it makes of
 * /root/reference/src/tracking/CGKLT/Shaders/ *.cg (see cg_shim.h for how) into oracle/_ref/libcgklt_ref.so.  The shader
 * arithmetic that runs below is the reference's; this file is the part of OpenGL the host code drives it with, restated:
 * one call of a shader's main() per fragment of the viewport, the texture coordinates the host's renderQuad* helpers
 * attach to the covering triangle interpolated to the fragment centre, the result converted to the render target's format.
 * Each cgref_* entry point cites the host lines whose pass schedule it follows.
 *
 * Buffers use the oracle's layouts (oracle/klt_oracle.h): the pyramid is (I, Ix, Iy, 0) binary16 per texel, level l at
 * okl_pyr_layout's offset; feature buffers are N x 3 binary32 in slot order (row-major over the fw x fh grid).
 */
#include "cg_shim.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

#undef float2
#undef float3
#undef float4

thread_local bool cg_discarded = false;

/* ---- the shader programs (defined by the piped objects) ---- */
namespace cg_pyr_pass1v { void main(sampler2D, float4, float4, float4, float4, float4&); }
namespace cg_pyr_pass1h { void main(sampler2D, float4, float4, float4, float4, float4&); }
namespace cg_pyr_pass2 { void main(sampler2D, float4, float4, float3&); }
namespace cg_det_pass1 { void main(sampler2D, float4, float4, float4, float4, float3&); }
namespace cg_det_pass2 { void main(sampler2D, float4, float4, float4, float4, float, float4, float4&); }
namespace cg_det_discriminator { void main(sampler2D, float2, float4, float4, float2, float4&); }
namespace cg_det_build_histpyr { void main(sampler2D, float2, float4, float4, float4&); }

typedef void (*cg_gain_fn)(sampler2D, sampler2D, sampler2D, sampler2D, float2, float2, float2, float2, float, float, float4, float,
                           float, float3&);
typedef void (*cg_nogain_fn)(sampler2D, sampler2D, sampler2D, float2, float2, float2, float, float, float4, float4&);
typedef void (*cg_nonmax_fn)(sampler2D, float2, float2, float4&);
typedef void (*cg_traverse_fn)(sampler2D, sampler2D, float2, float, float, float, float2, float4&);

/* variants that differ by a -D macro: the Makefile hands the same lists to the compile loop and to these tables */
#define X(hw) namespace cg_gain_hw##hw { void main(sampler2D, sampler2D, sampler2D, sampler2D, float2, float2, float2, float2, float, float, float4, float, float, float3&); }
CG_GAIN_LIST
#undef X
#define X(L, S, hw) namespace cg_nogain_L##L##_S##S##_hw##hw { void main(sampler2D, sampler2D, sampler2D, float2, float2, float2, float, float, float4, float4&); }
CG_NOGAIN_LIST
#undef X
#define X(d) namespace cg_nonmax_d##d { void main(sampler2D, float2, float2, float4&); }
CG_NONMAX_LIST
#undef X
#define X(n) namespace cg_traverse_n##n { void main(sampler2D, sampler2D, float2, float, float, float, float2, float4&); }
CG_TRAVERSE_LIST
#undef X

static cg_gain_fn find_gain(int hw) {
#define X(h) if (hw == h) return cg_gain_hw##h::main;
    CG_GAIN_LIST
#undef X
    return nullptr;
}
static cg_nogain_fn find_nogain(int L, int S, int hw) {
#define X(l, s, h) if (L == l && S == s && hw == h) return cg_nogain_L##l##_S##s##_hw##h::main;
    CG_NOGAIN_LIST
#undef X
    return nullptr;
}
static cg_nonmax_fn find_nonmax(int d) {
#define X(k) if (d == k) return cg_nonmax_d##k::main;
    CG_NONMAX_LIST
#undef X
    return nullptr;
}
static cg_traverse_fn find_traverse(int n) {
#define X(k) if (n == k) return cg_traverse_n##k::main;
    CG_TRAVERSE_LIST
#undef X
    return nullptr;
}

static inline float2 f2(float a, float b) { return cg_mk2(a, b); }
static inline float4 f4(float a, float b, float c, float d) { return cg_mk4(a, b, c, d); }

static size_t pyr_layout(int W, int H, int L, int64_t* off) { /* = okl_pyr_layout */
    size_t total = 0;
    for (int l = 0; l < L; ++l) {
        if (off) off[l] = (int64_t)total;
        size_t n = (size_t)(W >> l) * (size_t)(H >> l);
        total += (n + 63) & ~(size_t)63;
    }
    return total;
}

static cg_sampler one_level(int fmt, int filter, const void* data, int w, int h) {
    cg_sampler s;
    memset(&s, 0, sizeof(s));
    s.fmt = fmt, s.filter = filter, s.base_level = 0, s.n_levels = 1;
    s.lv[0].data = data, s.lv[0].w = w, s.lv[0].h = h;
    return s;
}
static cg_sampler pyramid_sampler(const uint16_t* pyr, int W, int H, int L, int filter, int base) {
    int64_t off[16];
    pyr_layout(W, H, L, off);
    cg_sampler s;
    memset(&s, 0, sizeof(s));
    s.fmt = CG_RGBA16F, s.filter = filter, s.base_level = base, s.n_levels = L;
    for (int l = 0; l < L; ++l) s.lv[l].data = pyr + 4 * off[l], s.lv[l].w = W >> l, s.lv[l].h = H >> l;
    return s;
}

/* the coordinates renderQuad8Tap (v3d_gpupyramid.cpp:32-52, v3d_gpuklt.cpp:20-40) attaches, at the fragment centre (s, t) */
static inline void taps8(float s, float t, float dS, float dT, float4 st[4]) {
    st[0] = f4(s - 3 * dS, t - 3 * dT, s - 2 * dS, t - 2 * dT);
    st[1] = f4(s - 1 * dS, t - 1 * dT, s - 0 * dS, t - 0 * dT);
    st[2] = f4(s + 1 * dS, t + 1 * dT, s + 2 * dS, t + 2 * dT);
    st[3] = f4(s + 3 * dS, t + 3 * dT, s + 4 * dS, t + 4 * dT);
}
/* renderQuad4Tap, v3d_gpupyramid.cpp:16-30 */
static inline void taps4(float s, float t, float dS, float dT, float4 st[2]) {
    st[0] = f4(s - 1 * dS, t - 1 * dT, s - 0 * dS, t - 0 * dT);
    st[1] = f4(s + 1 * dS, t + 1 * dT, s + 2 * dS, t + 2 * dT);
}

static inline void store_rgb16f(uint16_t* p, float3 c) { p[0] = cg_f2h(c.x), p[1] = cg_f2h(c.y), p[2] = cg_f2h(c.z), p[3] = 0; }

/* NEAREST taps exactly on a texel edge: `centered` = 1 resolves them to the lower texel (the geometrically centred
 * [1 3 3 1]); 0 is floor() of the exact coordinate.  cg_shim.h's bias is +1/256; the centred reading needs -1/256, which a
 * shift of the coordinate by -1/128 texel gives. */
static inline float edge_shift(int centered, int n_src) { return centered ? -0.0078125f / (float)n_src : 0.0f; }

extern "C" {

/* PyramidWithDerivativesCreator::buildPyramidForGrayscaleTexture, v3d_gpupyramid.cpp:376-429 (PRESMOOTHING = 1,
 * v3d_gpuklt.cpp:600-601).  Returns 0. */
int cgref_pyramid_build(const uint8_t* img, int W, int H, int L, int centered, uint16_t* pyr) {
    int64_t off[16];
    size_t total = pyr_layout(W, H, L, off);
    memset(pyr, 0, total * 4 * sizeof(uint16_t));
    /* :386-389  src (LUMINANCE8, NEAREST) -> tmp2 (RGBA16F, w x h), renderQuad8Tap(0, 1/h) */
    std::vector<uint16_t> tmp2((size_t)W * H * 4);
    {
        cg_sampler src = one_level(CG_L8, CG_NEAREST, img, W, H);
        const float dT = 1.0f / H;
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i) {
                float4 st[4], c;
                taps8((i + 0.5f) / W, (j + 0.5f) / H, 0.0f, dT, st);
                cg_pyr_pass1v::main(src, st[0], st[1], st[2], st[3], c);
                uint16_t* p = &tmp2[4 * ((size_t)j * W + i)];
                p[0] = cg_f2h(c.x), p[1] = cg_f2h(c.y), p[2] = cg_f2h(c.z), p[3] = cg_f2h(c.w);
            }
    }
    /* :392-398  tmp2 -> pyramid level 0 (RGB16F), renderQuad8Tap(1/w, 0) */
    {
        cg_sampler src = one_level(CG_RGBA16F, CG_NEAREST, tmp2.data(), W, H);
        const float dS = 1.0f / W;
        uint16_t* dst = pyr + 4 * off[0];
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i) {
                float4 st[4], c = cg_mk4(0.0f);
                taps8((i + 0.5f) / W, (j + 0.5f) / H, dS, 0.0f, st);
                cg_pyr_pass1h::main(src, st[0], st[1], st[2], st[3], c);
                store_rgb16f(dst + 4 * ((size_t)j * W + i), cg_mk3(c.x, c.y, c.z));
            }
    }
    /* :402-420  per level: pyramid level-1 -> tmp (w x h/2, RGB16F) with renderQuad4Tap(0, 1/h), then tmp -> level
     * (w/2 x h/2) with renderQuad4Tap(1/w, 0); both sample through GL_TEXTURE_BASE_LEVEL = level-1 */
    for (int level = 1; level < L; ++level) {
        const int Ws = W >> (level - 1), Hs = H >> (level - 1), Wd = Ws / 2, Hd = Hs / 2;
        std::vector<uint16_t> tmp((size_t)Ws * Hd * 4);
        {
            cg_sampler src = one_level(CG_RGBA16F, CG_NEAREST, pyr + 4 * off[level - 1], Ws, Hs);
            const float dT = 1.0f / Hs, sh = edge_shift(centered, Hs);
            for (int j = 0; j < Hd; ++j)
                for (int i = 0; i < Ws; ++i) {
                    float4 st[2];
                    float3 c;
                    taps4((i + 0.5f) / Ws, (j + 0.5f) / Hd + sh, 0.0f, dT, st);
                    cg_pyr_pass2::main(src, st[0], st[1], c);
                    store_rgb16f(&tmp[4 * ((size_t)j * Ws + i)], c);
                }
        }
        {
            cg_sampler src = one_level(CG_RGBA16F, CG_NEAREST, tmp.data(), Ws, Hd);
            const float dS = 1.0f / Ws, sh = edge_shift(centered, Ws);
            uint16_t* dst = pyr + 4 * off[level];
            for (int j = 0; j < Hd; ++j)
                for (int i = 0; i < Wd; ++i) {
                    float4 st[2];
                    float3 c;
                    taps4((i + 0.5f) / Wd + sh, (j + 0.5f) / Hd, dS, 0.0f, st);
                    cg_pyr_pass2::main(src, st[0], st[1], c);
                    store_rgb16f(dst + 4 * ((size_t)j * Wd + i), c);
                }
        }
    }
    return 0;
}

/* KLT_Tracker::trackFeatures, v3d_gpuklt.cpp:99-161: klt_tracker.cg once over the fw x fh feature buffer.  The host
 * passes -DNITERATIONS (the shader reads N_ITERATIONS, :108 vs klt_tracker.cg:16), so the program keeps its own 5.
 * levelSkip as KLT_TrackerBase stores it (v3d_gpuklt.h:14: <= 0 means nLevels-1).  Returns 0, or -1 when that macro
 * combination was not compiled into the library. */
int cgref_track_nogain(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int L, int levelSkip, int hw, int fw, int fh,
                       float margin, float convThr, float ssdThr, const float* featIn, float* featOut) {
    if (levelSkip <= 0) levelSkip = L - 1;
    cg_nogain_fn fn = find_nogain(L, levelSkip, hw);
    if (!fn) return -1;
    cg_sampler feat = one_level(CG_RGB32F, CG_NEAREST, featIn, fw, fh);
    cg_sampler im0 = pyramid_sampler(pyr0, W, H, L, CG_LINEAR, 0), im1 = pyramid_sampler(pyr1, W, H, L, CG_LINEAR, 0);
    const float ds = 1.0f / W, dt = 1.0f / H; /* :124-125 */
    const float4 vr = f4(margin / W, margin / H, 1.0f - margin / W, 1.0f - margin / H);
    for (int j = 0; j < fh; ++j)
        for (int i = 0; i < fw; ++i) {
            float4 c;
            fn(feat, im0, im1, f2((i + 0.5f) / fw, (j + 0.5f) / fh), f2(ds, dt), f2((float)W, (float)H), convThr * convThr, ssdThr,
               vr, c);
            float* o = featOut + 3 * ((size_t)j * fw + i);
            o[0] = c.x, o[1] = c.y, o[2] = c.z;
        }
    return 0;
}

/* one launch of klt_tracker_with_gain.cg on pyramid level `level` (GL_TEXTURE_BASE_LEVEL = level, v3d_gpuklt.cpp:262-266) */
int cgref_track_gain_pass(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int L, int level, int fw, int fh, int hw,
                          const float* feat0, const float* featIn, float* featOut, float sqrConvThr, float ssdThr,
                          const float validRegion[4], float lambda, float delta) {
    cg_gain_fn fn = find_gain(hw);
    if (!fn) return -1;
    cg_sampler features = one_level(CG_RGB32F, CG_NEAREST, featIn, fw, fh);
    cg_sampler features0 = one_level(CG_RGB32F, CG_NEAREST, feat0, fw, fh);
    cg_sampler im0 = pyramid_sampler(pyr0, W, H, L, CG_LINEAR, level), im1 = pyramid_sampler(pyr1, W, H, L, CG_LINEAR, level);
    const int w = W >> level, h = H >> level;
    const float2 ds = f2(1.0f / w, 1.0f / h), ds0 = f2(1.0f / fw, 1.0f / fh); /* :257-260, :247 */
    const float4 vr = f4(validRegion[0], validRegion[1], validRegion[2], validRegion[3]);
    for (int j = 0; j < fh; ++j)
        for (int i = 0; i < fw; ++i) {
            float3 c;
            fn(features, im0, im1, features0, f2((i + 0.5f) / fw, (j + 0.5f) / fh), ds, ds0, f2((float)W, (float)H), sqrConvThr,
               ssdThr, vr, lambda, delta, c);
            float* o = featOut + 3 * ((size_t)j * fw + i);
            o[0] = c.x, o[1] = c.y, o[2] = c.z;
        }
    return 0;
}

/* KLT_TrackerWithGain::trackFeaturesAndGain, v3d_gpuklt.cpp:205-305.  feat0 = _featuresBuffer2 (the list provided for the
 * previous frame), featCur = _featuresBuffer0 on entry (its gain channel is cleared to 1 first, :223-227); featOut
 * receives what _featuresBuffer2 holds on return (= what readFeaturesAndGain reads). */
int cgref_track_gain(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int L, int levelSkip, int hw, int nIterations,
                     int fw, int fh, float margin, float convThr, float ssdThr, const float* feat0, const float* featCur,
                     float* featOut) {
    if (!find_gain(hw)) return -1;
    if (levelSkip <= 0) levelSkip = L - 1;
    if (levelSkip <= 0) levelSkip = 1;
    const size_t N = (size_t)fw * fh;
    std::vector<float> A(featCur, featCur + 3 * N), B(3 * N, -1.0f);
    for (size_t k = 0; k < N; ++k) A[3 * k + 2] = 1.0f;
    float *b0 = A.data(), *b1 = B.data();
    float delta = 200.0f; /* :243 */
    const float tau = 1.0f;
    float sqrConv = 1000000.0f, ssd = 1000000.0f, vr[4] = {-1.0f, -1.0f, 2.0f, 2.0f}; /* :247-250 */
    for (int level = L - 1; level >= 0; level -= levelSkip) {                         /* :254 */
        for (int iter = 1; iter <= nIterations; ++iter) {                               /* :268 */
            const float dcur = delta;
            delta *= tau;
            if (iter == 1) { /* :271-274 */
                sqrConv = 1000000.0f, ssd = 1000000.0f;
                vr[0] = vr[1] = -1.0f, vr[2] = vr[3] = 2.0f;
            } else if (iter == nIterations) { /* :275-279 */
                sqrConv = convThr * convThr, ssd = ssdThr;
                vr[0] = margin / W, vr[1] = margin / H, vr[2] = 1.0f - margin / W, vr[3] = 1.0f - margin / H;
            }
            cgref_track_gain_pass(pyr0, pyr1, W, H, L, level, fw, fh, hw, feat0, b0, b1, sqrConv, ssd, vr, 1.0f, dcur);
            std::swap(b0, b1); /* :285 */
        }
    }
    memcpy(featOut, b0, sizeof(float) * 3 * N); /* :304 */
    return 0;
}

/* KLT_Detector::detectCorners, first half (v3d_gpuklt.cpp:457-473): klt_detector_pass1.cg into _convRowsBuffer (RGB32F),
 * klt_detector_pass2.cg into _cornernessBuffer (RGBA8 carrying one binary32).  lvl0 = pyramid level 0. */
int cgref_cornerness(const uint16_t* lvl0, int W, int H, float minCornerness, float margin, float* out) {
    cg_sampler pyr = one_level(CG_RGBA16F, CG_NEAREST, lvl0, W, H);
    std::vector<float> conv((size_t)W * H * 3);
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            float4 st[4];
            float3 c;
            taps8((i + 0.5f) / W, (j + 0.5f) / H, 0.0f, 1.0f / H, st); /* :464 */
            cg_det_pass1::main(pyr, st[0], st[1], st[2], st[3], c);
            float* o = &conv[3 * ((size_t)j * W + i)];
            o[0] = c.x, o[1] = c.y, o[2] = c.z;
        }
    cg_sampler rows = one_level(CG_RGB32F, CG_NEAREST, conv.data(), W, H);
    const float4 vr = f4(margin / W, margin / H, 1.0f - margin / W, 1.0f - margin / H); /* :470 */
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            float4 st[4], c;
            taps8((i + 0.5f) / W, (j + 0.5f) / H, 1.0f / W, 0.0f, st); /* :472 */
            cg_det_pass2::main(rows, st[0], st[1], st[2], st[3], minCornerness, vr, c);
            out[(size_t)j * W + i] = pack_4ubyte(c); /* the RGBA8 target, read back as the float it carries */
        }
    return 0;
}

/* :475-500: GL_POINTS of size 1 at the present features, colour unpack_4ubyte(-1e30).  A point at normalized (s, t)
 * under setupNormalizedProjection covers the pixel that contains it; points outside [0,1) are clipped. */
int cgref_suppress_present(float* corner, int W, int H, int nPresent, const float* present3) {
    for (int k = 0; k < nPresent; ++k) {
        const float s = present3[3 * k], t = present3[3 * k + 1];
        if (!(s >= 0.0f && t >= 0.0f)) continue;
        const float fx = floorf(s * (float)W), fy = floorf(t * (float)H);
        if (fx >= (float)W || fy >= (float)H) continue;
        corner[(size_t)(int)fy * W + (int)fx] = pack_4ubyte(unpack_4ubyte(-1e30f));
    }
    return 0;
}

static std::vector<uint8_t> as_rgba8(const float* v, size_t n) {
    std::vector<uint8_t> b(4 * n);
    memcpy(b.data(), v, 4 * n);
    return b;
}

/* :502-512: klt_detector_nonmax.cg with ds = (1/w, 0) into _nonmaxRowsBuffer, then ds = (0, 1/h) back into
 * _cornernessBuffer.  In place.  Returns -1 when MIN_DIST = minDist was not compiled in. */
int cgref_nonmax(float* corner, int W, int H, int minDist) {
    cg_nonmax_fn fn = find_nonmax(minDist);
    if (!fn) return -1;
    std::vector<uint8_t> a = as_rgba8(corner, (size_t)W * H), b(a.size());
    for (int pass = 0; pass < 2; ++pass) {
        cg_sampler src = one_level(CG_RGBA8, CG_NEAREST, pass ? b.data() : a.data(), W, H);
        uint8_t* dst = pass ? a.data() : b.data();
        const float2 ds = pass ? f2(0.0f, 1.0f / H) : f2(1.0f / W, 0.0f);
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < W; ++i) {
                float4 c;
                fn(src, f2((i + 0.5f) / W, (j + 0.5f) / H), ds, c);
                uint8_t* p = dst + 4 * ((size_t)j * W + i);
                p[0] = cg_unorm8(c.x), p[1] = cg_unorm8(c.y), p[2] = cg_unorm8(c.z), p[3] = cg_unorm8(c.w);
            }
    }
    memcpy(corner, a.data(), a.size());
    return 0;
}

/* detectCorners' second half + extractCorners (v3d_gpuklt.cpp:514-588): discriminator into level 0 of the POT histogram
 * pyramid (RGBA32F, cleared to 0, viewport w/2 x h/2), build_histpyr up to the 1 x 1 level, count = sum of its four
 * channels; then klt_detector_traverse_histpyr.cg over a plw-wide point list, `discard` leaving the clear colour -1.
 * list3 receives min(count, maxOut) entries (s, t, cornerness) in the shader's order; returns the count, or -1 when
 * PYR_LEVELS for this size was not compiled in. */
int cgref_extract(const float* corner, int W, int H, int plw, int maxOut, float* list3) {
    int pw = 1, nl = 0; /* :330-336 */
    while (pw < std::max(W, H)) pw *= 2, ++nl;
    pw = 1 << (nl - 1);
    cg_traverse_fn fn = find_traverse(nl);
    if (!fn) return -1;
    std::vector<std::vector<float>> lv(nl);
    for (int k = 0; k < nl; ++k) lv[k].assign((size_t)(pw >> k) * (pw >> k) * 4, 0.0f);
    std::vector<uint8_t> cb = as_rgba8(corner, (size_t)W * H);
    cg_sampler cor = one_level(CG_RGBA8, CG_NEAREST, cb.data(), W, H);
    { /* :515-523 render2x2Tap(-0.5/w, -0.5/h, 1/w, 1/h) on a w/2 x h/2 viewport */
        const int vw = W / 2, vh = H / 2;
        const float shS = -0.5f / W, shT = -0.5f / H, dS = 1.0f / W, dT = 1.0f / H;
        for (int j = 0; j < vh; ++j)
            for (int i = 0; i < vw; ++i) {
                const float s = (i + 0.5f) / vw, t = (j + 0.5f) / vh;
                float4 c;
                cg_det_discriminator::main(cor, f2(s, t), f4(s + shS, t + shT, s + shS + dS, t + shT),
                                           f4(s + shS, t + shT + dT, s + shS + dS, t + shT + dT), f2((float)W, (float)H), c);
                float* o = &lv[0][4 * ((size_t)j * pw + i)];
                o[0] = c.x, o[1] = c.y, o[2] = c.z, o[3] = c.w;
            }
    }
    for (int k = 1; k < nl; ++k) { /* :528-534 render2x2Tap(-0.25/W, -0.25/W, 0.5/W, 0.5/W), BASE_LEVEL = k-1 */
        const int Wk = pw >> k;
        cg_sampler src = one_level(CG_RGBA32F, CG_NEAREST, lv[k - 1].data(), pw >> (k - 1), pw >> (k - 1));
        const float sh = -0.25f / Wk, d = 0.5f / Wk;
        for (int j = 0; j < Wk; ++j)
            for (int i = 0; i < Wk; ++i) {
                const float s = (i + 0.5f) / Wk, t = (j + 0.5f) / Wk;
                float4 c;
                cg_det_build_histpyr::main(src, f2(s, t), f4(s + sh, t + sh, s + sh + d, t + sh), f4(s + sh, t + sh + d, s + sh + d, t + sh + d),
                                           c);
                float* o = &lv[k][4 * ((size_t)j * Wk + i)];
                o[0] = c.x, o[1] = c.y, o[2] = c.z, o[3] = c.w;
            }
    }
    const float* top = lv[nl - 1].data();
    const int count = (int)(top[0] + top[1] + top[2] + top[3]); /* :541-544 */
    const int nFeatures = std::min(count, maxOut);
    if (nFeatures <= 0) return count;
    cg_sampler hp;
    memset(&hp, 0, sizeof(hp));
    hp.fmt = CG_RGBA32F, hp.filter = CG_NEAREST, hp.base_level = 0, hp.n_levels = nl; /* NEAREST_MIPMAP_NEAREST, :557 */
    for (int k = 0; k < nl; ++k) hp.lv[k].data = lv[k].data(), hp.lv[k].w = hp.lv[k].h = pw >> k;
    const int rows = (nFeatures + plw - 1) / plw; /* :551 */
    /* :571-577: TEXCOORD0 runs from -0.5 to 2 plw - 0.5 over the covering triangle => (i, j) at fragment (i, j) */
    for (int j = 0; j < rows; ++j)
        for (int i = 0; i < plw; ++i) {
            float4 c = cg_mk4(-1.0f); /* glClearColor(-1,-1,-1,0), :547 */
            cg_discarded = false;
            fn(hp, cor, f2((float)i, (float)j), (float)nFeatures, (float)plw, (float)pw, f2((float)W, (float)H), c);
            const size_t k = (size_t)j * plw + i;
            if (k >= (size_t)nFeatures) continue; /* beyond what the host reads as features (:756-786 use nFeatures entries) */
            if (cg_discarded) c = cg_mk4(-1.0f);
            list3[3 * k] = c.x, list3[3 * k + 1] = c.y, list3[3 * k + 2] = c.z;
        }
    return count;
}

/* which macro variants this build holds (for skip messages) */
int cgref_has_gain(int hw) { return find_gain(hw) != nullptr; }
int cgref_has_nogain(int L, int levelSkip, int hw) { return find_nogain(L, levelSkip <= 0 ? L - 1 : levelSkip, hw) != nullptr; }
int cgref_has_nonmax(int d) { return find_nonmax(d) != nullptr; }
int cgref_has_traverse(int W, int H) {
    int pw = 1, nl = 0;
    while (pw < std::max(W, H)) pw *= 2, ++nl;
    return find_traverse(nl) != nullptr;
}

} /* extern "C" */
