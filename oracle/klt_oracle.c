/*
 * oracle/klt_oracle.c -- CPU restatement of CoSLAM's GPU-KLT hot path (see klt_oracle.h).
 * TEST INFRASTRUCTURE ONLY.  Arithmetic pinned, pass by pass and bit for bit, to the reference's own .cg shaders compiled in
 * place (oracle/build_cgref.sh, tests/test_cgklt_cpu.py); the GL texture model and the host's pass schedule are restatements.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * All citations are relative to the reference root (danping/CoSLAM).
 */
#include "klt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ half <-> float */

uint16_t okl_f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t em = x & 0x7fffffffu;
    if (em >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((em > 0x7f800000u) ? 0x200u : 0));
    }
    if (em >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (em < 0x33000001u) { /* <= 2^-25: rounds to zero (tie at 2^-25 -> even = 0) */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(em >> 23) - 127;
    uint32_t m = (em & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    int shift;                                  /* bits to drop */
    uint32_t base;
    if (e < -14) { /* subnormal half */
        shift = 13 + (-14 - e);
        base = 0;
    } else {
        shift = 13;
        base = (uint32_t)(e + 15) << 10;
        m &= 0x7fffffu;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (base + q)); /* carry into exponent is correct by construction */
}

float okl_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            int sh = 0;
            while (!(m & 0x400u)) {
                m <<= 1;
                sh++;
            }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 127 - 15) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* ------------------------------------------------------------------ config / layout */

void okl_config_default(okl_config* c) { /* v3d_gpuklt.h:181-191 */
    c->nIterations = 12;
    c->nLevels = 3;
    c->levelSkip = 2;
    c->windowWidth = 5;
    c->trackBorderMargin = 4.0f;
    c->convergenceThreshold = 0.1f;
    c->SSD_Threshold = 5000.0f;
    c->trackWithGain = 0;
    c->minDistance = 8;
    c->minCornerness = 1000.0f;
    c->detectBorderMargin = 4.0f;
}

size_t okl_pyr_layout(int W, int H, int nLevels, int64_t* off) {
    size_t total = 0;
    for (int l = 0; l < nLevels; ++l) {
        if (off) off[l] = (int64_t)total;
        size_t n = (size_t)(W >> l) * (size_t)(H >> l);
        total += (n + 63) & ~(size_t)63;
    }
    return total;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------ pyramid */

/* LUMINANCE8 fetch scaled back by 255 as the shaders do (pass1v.cg:79-80) */
static inline float lum255(uint8_t c) { return ((float)c / 255.0f) * 255.0f; }

static inline void store_texel(uint16_t* p, float I, float Ix, float Iy) {
    p[0] = okl_f32_to_f16(I);
    p[1] = okl_f32_to_f16(Ix);
    p[2] = okl_f32_to_f16(Iy);
    p[3] = 0;
}

/* level 0: pyramid_with_derivative_pass1v.cg:63-83 (PRESMOOTHING==1, v3d_gpuklt.cpp:600-601)
 * then pyramid_with_derivative_pass1h.cg:96-128; tap geometry v3d_gpupyramid.cpp:32-52,386-398.
 * The intermediate (v, dv) is carried as exact binary32 (unpack_2half into RGBA16F, pass1v.cg:124-125). */
static void pyr_level0(const uint8_t* img, int W, int H, uint16_t* out) {
    float* v = (float*)malloc(sizeof(float) * (size_t)W * H);
    float* dv = (float*)malloc(sizeof(float) * (size_t)W * H);
    for (int y = 0; y < H; ++y) {
        const uint8_t* rm2 = img + (size_t)clampi(y - 2, 0, H - 1) * W;
        const uint8_t* rm1 = img + (size_t)clampi(y - 1, 0, H - 1) * W;
        const uint8_t* r0 = img + (size_t)y * W;
        const uint8_t* rp1 = img + (size_t)clampi(y + 1, 0, H - 1) * W;
        const uint8_t* rp2 = img + (size_t)clampi(y + 2, 0, H - 1) * W;
        for (int x = 0; x < W; ++x) {
            float g0 = lum255(rm2[x]), g1 = lum255(rm1[x]), g2 = lum255(r0[x]), g3 = lum255(rp1[x]),
                  g4 = lum255(rp2[x]);
            /* dot(f1,g1), f1=(0,1,2,1)/4 ; dot(df1,g1)+df2*g2, df1=(-1,-2,0,2)/8, df2=1/8 */
            v[(size_t)y * W + x] = ((0.0f * g0 + 0.25f * g1) + 0.5f * g2) + 0.25f * g3;
            dv[(size_t)y * W + x] = (((-0.125f * g0 + -0.25f * g1) + 0.0f * g2) + 0.25f * g3) + 0.125f * g4;
        }
    }
    for (int y = 0; y < H; ++y) {
        const float* vr = v + (size_t)y * W;
        const float* dr = dv + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            int xm2 = clampi(x - 2, 0, W - 1), xm1 = clampi(x - 1, 0, W - 1), xp1 = clampi(x + 1, 0, W - 1),
                xp2 = clampi(x + 2, 0, W - 1);
            float I = ((0.0f * vr[xm2] + 0.25f * vr[xm1]) + 0.5f * vr[x]) + 0.25f * vr[xp1];
            float Ix = (((-0.125f * vr[xm2] + -0.25f * vr[xm1]) + 0.0f * vr[x]) + 0.25f * vr[xp1]) + 0.125f * vr[xp2];
            float Iy = ((0.0f * dr[xm2] + 0.25f * dr[xm1]) + 0.5f * dr[x]) + 0.25f * dr[xp1];
            store_texel(out + 4 * ((size_t)y * W + x), I, Ix, Iy);
        }
    }
    free(v);
    free(dv);
}

/* tap base for the [1 3 3 1] decimation: output index o of n_dst samples over n_src texels.
 * renderQuad4Tap (v3d_gpupyramid.cpp:16-30) issues taps at c-1, c, c+1, c+2 texels where
 * c = (o+0.5)/n_dst*n_src (= 2o+1 for even n_src): exactly on a texel boundary; GL NEAREST = floor. */
static inline int tap_base(int o, int n_dst, int n_src) {
    return (int)(((int64_t)(2 * o + 1) * n_src) / (2 * (int64_t)n_dst));
}

/* pyramid_with_derivative_pass2.cg:6-12 applied vertically then horizontally, both
 * targets RGB16F (v3d_gpupyramid.cpp:256,285-293,402-420) */
static void pyr_downsample(const uint16_t* src, int Ws, int Hs, int Wd, int Hd, int centered, uint16_t* dst) {
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * 4 * (size_t)Ws * Hd);
    int shift = centered ? -1 : 0;
    for (int y = 0; y < Hd; ++y) {
        int b = tap_base(y, Hd, Hs) + shift;
        int r1 = clampi(b - 1, 0, Hs - 1), r2 = clampi(b, 0, Hs - 1), r3 = clampi(b + 1, 0, Hs - 1),
            r4 = clampi(b + 2, 0, Hs - 1);
        for (int x = 0; x < Ws; ++x) {
            for (int c = 0; c < 3; ++c) {
                float v1 = okl_f16_to_f32(src[4 * ((size_t)r1 * Ws + x) + c]);
                float v2 = okl_f16_to_f32(src[4 * ((size_t)r2 * Ws + x) + c]);
                float v3 = okl_f16_to_f32(src[4 * ((size_t)r3 * Ws + x) + c]);
                float v4 = okl_f16_to_f32(src[4 * ((size_t)r4 * Ws + x) + c]);
                float r = (((v1 + 3.0f * v2) + 3.0f * v3) + v4) / 8.0f;
                tmp[4 * ((size_t)y * Ws + x) + c] = okl_f32_to_f16(r);
            }
            tmp[4 * ((size_t)y * Ws + x) + 3] = 0;
        }
    }
    for (int y = 0; y < Hd; ++y) {
        for (int x = 0; x < Wd; ++x) {
            int b = tap_base(x, Wd, Ws) + shift;
            int c1 = clampi(b - 1, 0, Ws - 1), c2 = clampi(b, 0, Ws - 1), c3 = clampi(b + 1, 0, Ws - 1),
                c4 = clampi(b + 2, 0, Ws - 1);
            for (int c = 0; c < 3; ++c) {
                float v1 = okl_f16_to_f32(tmp[4 * ((size_t)y * Ws + c1) + c]);
                float v2 = okl_f16_to_f32(tmp[4 * ((size_t)y * Ws + c2) + c]);
                float v3 = okl_f16_to_f32(tmp[4 * ((size_t)y * Ws + c3) + c]);
                float v4 = okl_f16_to_f32(tmp[4 * ((size_t)y * Ws + c4) + c]);
                float r = (((v1 + 3.0f * v2) + 3.0f * v3) + v4) / 8.0f;
                dst[4 * ((size_t)y * Wd + x) + c] = okl_f32_to_f16(r);
            }
            dst[4 * ((size_t)y * Wd + x) + 3] = 0;
        }
    }
    free(tmp);
}

void okl_pyramid_build(const uint8_t* img, int W, int H, int nLevels, int centered, uint16_t* pyr) {
    int64_t off[OKL_MAX_LEVELS];
    size_t total = okl_pyr_layout(W, H, nLevels, off);
    memset(pyr, 0, total * 4 * sizeof(uint16_t));
    pyr_level0(img, W, H, pyr + 4 * off[0]);
    for (int l = 1; l < nLevels; ++l) {
        pyr_downsample(pyr + 4 * off[l - 1], W >> (l - 1), H >> (l - 1), W >> l, H >> l, centered, pyr + 4 * off[l]);
    }
}

/* ------------------------------------------------------------------ bilinear fetch */

void okl_sample(const uint16_t* lvl, int Wl, int Hl, float s, float t, float out[3]) {
    float u = s * (float)Wl - 0.5f;
    float v = t * (float)Hl - 0.5f;
    /* coordinates far outside behave as the clamped edge texel; bounding them first keeps the
     * float->int conversion defined (NaN -> lower bound) without changing any in-range result */
    u = fminf(fmaxf(u, -2.0f), (float)Wl + 1.0f);
    v = fminf(fmaxf(v, -2.0f), (float)Hl + 1.0f);
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = clampi((int)fu, 0, Wl - 1), i1 = clampi((int)fu + 1, 0, Wl - 1);
    int j0 = clampi((int)fv, 0, Hl - 1), j1 = clampi((int)fv + 1, 0, Hl - 1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    const uint16_t* p00 = lvl + 4 * ((size_t)j0 * Wl + i0);
    const uint16_t* p10 = lvl + 4 * ((size_t)j0 * Wl + i1);
    const uint16_t* p01 = lvl + 4 * ((size_t)j1 * Wl + i0);
    const uint16_t* p11 = lvl + 4 * ((size_t)j1 * Wl + i1);
    for (int c = 0; c < 3; ++c) {
        out[c] = ((w00 * okl_f16_to_f32(p00[c]) + w10 * okl_f16_to_f32(p10[c])) + w01 * okl_f16_to_f32(p01[c])) +
                 w11 * okl_f16_to_f32(p11[c]);
    }
}

/* ------------------------------------------------------------------ tracker, no gain */

/* The HIP kernel's order of the window sums (k_track_nogain, coslam_amd/csrc/klt_track.hip): pixel p = 64 q + lane goes to
 * lane p % 64, a lane adds its pixels in the order of q, and the 64 lane values are folded by cs_wave_sum -- a balanced binary
 * tree inside each row of 16 lanes (commutative at every level, so every lane of the row holds the same bits), then
 * (row3 + row2) + (row1 + row0).  Same per-pixel arithmetic as the serial order below. */
static float nogain_tree_sum(const float lanes[64]) {
    float R[4];
    for (int r = 0; r < 4; ++r) {
        float q[4];
        for (int k = 0; k < 4; ++k) {
            const float* a = lanes + 16 * r + 4 * k;
            q[k] = (a[0] + a[1]) + (a[2] + a[3]);
        }
        R[r] = (q[0] + q[1]) + (q[2] + q[3]);
    }
    return (R[3] + R[2]) + (R[1] + R[0]);
}

/* threshold-margin diagnostic (defined with the gain tracker below): smallest relative distance of a tested quantity to its
 * threshold, per slot */
static void note_margin_rel(size_t k, float value, float thr);
static void note_margin_region(size_t k, float X1x, float X1y, const float vr[4]);

static int g_nogain_sum_mode = 0; /* 0 serial (the shader's loop order), 1 the HIP kernel's tree */
void okl_set_nogain_sum_mode(int mode) { g_nogain_sum_mode = mode; }

void okl_track_nogain(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int nLevels, int levelSkip,
                      int hw, int nIterShader, float margin, float convThr, float ssdThr, int N, const float* featIn,
                      float* featOut) {
    int64_t off[OKL_MAX_LEVELS];
    okl_pyr_layout(W, H, nLevels, off);
    /* uniforms, v3d_gpuklt.cpp:124-147 */
    const float ds = 1.0f / (float)W, dt = 1.0f / (float)H;
    const float whx = (float)W, why = (float)H;
    const float sqrConv = convThr * convThr;
    const float vr0 = margin / (float)W, vr1 = margin / (float)H, vr2 = 1.0f - margin / (float)W,
                vr3 = 1.0f - margin / (float)H;
    if (levelSkip <= 0) levelSkip = nLevels - 1; /* v3d_gpuklt.h:14 */
    if (levelSkip <= 0) levelSkip = 1;           /* nLevels==1: single level */

    for (int k = 0; k < N; ++k) {
        float X0x = featIn[3 * k], X0y = featIn[3 * k + 1];
        float X1x = X0x, X1y = X0y;
        int invalid = (X1x < 0); /* klt_tracker.cg:47 */
        float sqrLen = 0.0f, SSD = 0.0f;
        float mult = (float)(1 << (nLevels - 1)); /* :20,50 */
        if (invalid) {
            featOut[3 * k] = featOut[3 * k + 1] = featOut[3 * k + 2] = -1.0f;
            continue;
        }
        for (int level = nLevels - 1; level >= 0; level -= levelSkip) { /* :53 */
            const uint16_t* L0 = pyr0 + 4 * off[level];
            const uint16_t* L1 = pyr1 + 4 * off[level];
            int Wl = W >> level, Hl = H >> level;
            float dsx = ds * mult, dsy = dt * mult; /* :56 */
            for (int iter = 0; iter < nIterShader; ++iter) {
                float a = 0, b = 0, c = 0, rx = 0, ry = 0;
                SSD = 0;
                float la[6][64];
                if (g_nogain_sum_mode == 1) memset(la, 0, sizeof(la));
                int p = 0;
                for (int y = -hw; y <= hw; ++y) {
                    float st_y = X0y + (float)y * dsy, st_w = X1y + (float)y * dsy; /* :81-82 */
                    for (int x = -hw; x <= hw; ++x, ++p) {
                        float st_x = X0x + (float)x * dsx, st_z = X1x + (float)x * dsx;
                        float I0[3], I1[3];
                        okl_sample(L0, Wl, Hl, st_x, st_y, I0);
                        okl_sample(L1, Wl, Hl, st_z, st_w, I1);
                        float e = I0[0] - I1[0];                   /* :93 */
                        float gx = (I0[1] + I1[1]) * whx / 2.0f; /* :94 */
                        float gy = (I0[2] + I1[2]) * why / 2.0f;
                        if (g_nogain_sum_mode == 1) {
                            const int ln = p & 63;
                            la[0][ln] += gx * gx, la[1][ln] += gx * gy, la[2][ln] += gy * gy;
                            la[3][ln] += e * gx, la[4][ln] += e * gy, la[5][ln] += e * e;
                            continue;
                        }
                        a += gx * gx; /* :105 abc += IJ.yyz*IJ.yzz */
                        b += gx * gy;
                        c += gy * gy;
                        rx += e * gx; /* :106 */
                        ry += e * gy;
                        SSD += e * e; /* :107 */
                    }
                }
                if (g_nogain_sum_mode == 1) {
                    a = nogain_tree_sum(la[0]), b = nogain_tree_sum(la[1]), c = nogain_tree_sum(la[2]);
                    rx = nogain_tree_sum(la[3]), ry = nogain_tree_sum(la[4]), SSD = nogain_tree_sum(la[5]);
                }
                float det = a * c - b * b;               /* :111 */
                note_margin_rel((size_t)k, det, 0.00001f);
                invalid = invalid || (det < 0.00001f);   /* :113 */
                float rdet = 1.0f / det;                 /* :115 */
                float dXx = rdet * (c * rx - b * ry);
                float dXy = rdet * (-b * rx + a * ry);
                X1x += dXx;
                X1y += dXy;
                dXx *= whx;
                dXy *= why;
                sqrLen = dXx * dXx + dXy * dXy; /* :120 */
            }
            note_margin_rel((size_t)k, sqrLen, sqrConv);
            note_margin_rel((size_t)k, SSD, ssdThr);
            invalid = invalid || (sqrLen > sqrConv); /* :123 */
            invalid = invalid || (SSD > ssdThr);     /* :124 */
            mult /= (float)(1 << levelSkip);         /* :126 */
        }
        {
            const float vr[4] = {vr0, vr1, vr2, vr3};
            note_margin_region((size_t)k, X1x, X1y, vr);
        }
        invalid = invalid || (X1x < vr0 || X1y < vr1) || (X1x > vr2 || X1y > vr3); /* :129 */
        if (invalid || !(X1x == X1x) || !(X1y == X1y)) {
            featOut[3 * k] = featOut[3 * k + 1] = featOut[3 * k + 2] = -1.0f;
        } else {
            featOut[3 * k] = X1x;
            featOut[3 * k + 1] = X1y;
            featOut[3 * k + 2] = X0x; /* :131 */
        }
    }
}

/* ------------------------------------------------------------------ tracker with gain */

/* Test diagnostic: when set, every gain pass records per slot the smallest RELATIVE distance of a tested quantity to the
 * threshold it is tested against -- |det - 1e-5| / 1e-5, |SSD - thr| / thr, |dX|^2 vs thr^2 likewise, and the distance
 * of the new position to the valid-region border relative to the border margin.  A slot whose status differs between
 * two summation orders must sit within ~1 % of one of them. */
static float* g_thr_margin = NULL;
void okl_set_threshold_margin_buffer(float* perSlot) { g_thr_margin = perSlot; }

static void note_margin_rel(size_t k, float value, float thr) {
    if (!g_thr_margin) return;
    const float m = fabsf(value - thr) / thr;
    if (!(m >= g_thr_margin[k])) g_thr_margin[k] = m;
}
static void note_margin_region(size_t k, float X1x, float X1y, const float vr[4]) {
    if (!g_thr_margin || !(vr[0] > 0.0f)) return;
    float m = fminf(fabsf(X1x - vr[0]), fabsf(X1x - vr[2])) / vr[0];
    const float t = fminf(fabsf(X1y - vr[1]), fabsf(X1y - vr[3])) / vr[1];
    if (t < m) m = t;
    if (!(m >= g_thr_margin[k])) g_thr_margin[k] = m;
}

static void note_margin(size_t k, float det, float SSD, float ssdThr, float sqrLen, float sqrConvThr, float X1x, float X1y,
                        const float vr[4]) {
    if (!g_thr_margin) return;
    float m = fabsf(det - 0.00001f) / 0.00001f;
    float t = fabsf(SSD - ssdThr) / ssdThr;
    if (t < m) m = t;
    t = fabsf(sqrLen - sqrConvThr) / sqrConvThr;
    if (t < m) m = t;
    if (vr[0] > 0.0f) { /* real valid region (the relaxed one is (-1,-1,2,2)) */
        const float bx = vr[0], by = vr[1];
        t = fminf(fabsf(X1x - vr[0]), fabsf(X1x - vr[2])) / bx;
        if (t < m) m = t;
        t = fminf(fabsf(X1y - vr[1]), fabsf(X1y - vr[3])) / by;
        if (t < m) m = t;
    }
    if (!(m >= g_thr_margin[k])) g_thr_margin[k] = m; /* NaN counts as "at a threshold" */
}

static inline float slot_beta(const float* feat, int fw, int fh, int i, int j) {
    i = clampi(i, 0, fw - 1); /* NEAREST + clamp, features textures */
    j = clampi(j, 0, fh - 1);
    return feat[3 * ((size_t)j * fw + i) + 2];
}

void okl_track_gain_pass(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int nLevels, int level, int fw,
                         int fh, int hw, const float* feat0, const float* featIn, float* featOut, float sqrConvThr,
                         float ssdThr, const float vr[4], float lambda, float delta) {
    int64_t off[OKL_MAX_LEVELS];
    okl_pyr_layout(W, H, nLevels, off);
    const uint16_t* L0 = pyr0 + 4 * off[level];
    const uint16_t* L1 = pyr1 + 4 * off[level];
    const int Wl = W >> level, Hl = H >> level;
    const float dsx = 1.0f / (float)Wl, dsy = 1.0f / (float)Hl; /* v3d_gpuklt.cpp:255-260 */
    const float whx = (float)W, why = (float)H;                  /* :246 */
    /* neighbour offsets in slot texels. st0 +- ds0.x is a scalar broadcast (klt_tracker_with_gain.cg:64-67):
     * both coordinates move by 1/fw (resp. 1/fh); NEAREST => floor((i+0.5) +- r) = i + floor(0.5 +- r) */
    const double rxy = (double)fh / (double)fw, ryx = (double)fw / (double)fh;
    const int n1x[4] = {1, -1, (int)floor(0.5 + ryx), (int)floor(0.5 - ryx)};
    const int n1y[4] = {(int)floor(0.5 + rxy), (int)floor(0.5 - rxy), 1, -1};
    const int n2x[4] = {1, -1, 0, 0}; /* :69-72 */
    const int n2y[4] = {0, 0, 1, -1};

    for (int j = 0; j < fh; ++j) {
        for (int i = 0; i < fw; ++i) {
            size_t k = (size_t)j * fw + i;
            float X0x = feat0[3 * k], X0y = feat0[3 * k + 1]; /* :57 */
            float X1x = featIn[3 * k], X1y = featIn[3 * k + 1], beta = featIn[3 * k + 2];
            int invalid = (X1x < 0) || (X0x < 0); /* :77 */
            if (invalid) {                        /* result is (-1,-1,-1) whatever the sums are (:147) */
                featOut[3 * k] = featOut[3 * k + 1] = featOut[3 * k + 2] = -1.0f;
                continue;
            }
            float bsum = 0.0f; /* dot(float4(1), betaN1+betaN2-2*beta), :111 */
            {
                float t4[4];
                for (int q = 0; q < 4; ++q) {
                    float b1 = slot_beta(featIn, fw, fh, i + n1x[q], j + n1y[q]);
                    float b2 = slot_beta(featIn, fw, fh, i + n2x[q], j + n2y[q]);
                    if (b1 < 0) b1 = beta; /* :74-75 */
                    if (b2 < 0) b2 = beta;
                    t4[q] = (b1 + b2) - 2.0f * beta;
                }
                bsum = ((t4[0] + t4[1]) + t4[2]) + t4[3];
            }
            float a = 0, b = 0, c = 0, d = 0, e_ = 0, f = 0, r0 = 0, r1 = 0, r2 = 0, SSD = 0;
            for (int y = -hw; y <= hw; ++y) {
                float st_y = X0y + (float)y * dsy, st_w = X1y + (float)y * dsy; /* :88-89 */
                for (int x = -hw; x <= hw; ++x) {
                    float st_x = X0x + (float)x * dsx, st_z = X1x + (float)x * dsx;
                    float I0[3], I1[3];
                    okl_sample(L0, Wl, Hl, st_x, st_y, I0);
                    okl_sample(L1, Wl, Hl, st_z, st_w, I1);
                    float ex = beta * I0[0] - I1[0];                        /* :99 */
                    float gx = (beta * I0[1] + I1[1]) * whx / 2.0f;       /* :100 */
                    float gy = (beta * I0[2] + I1[2]) * why / 2.0f;
                    float m0 = sqrtf(I0[1] * I0[1] + I0[2] * I0[2]); /* :102 */
                    float m1 = sqrtf(I1[1] * I1[1] + I1[2] * I1[2]); /* :103 */
                    a += gx * gx;                                      /* :106 abc += IJ.y*(IJ.y,IJ.z,-I0.x) */
                    b += gx * gy;
                    c += gx * (-I0[0]);
                    d += gy * gy; /* :107 def.xy += IJ.z*(IJ.z,-I0.x) */
                    e_ += gy * (-I0[0]);
                    f += (I0[0] * I0[0] + lambda * m0 * m0) + delta * 8.0f; /* :108 */
                    r0 += ex * gx;                                          /* :110 */
                    r1 += ex * gy;
                    r2 += (-ex * I0[0] + lambda * m0 * (m1 - beta * m0)) + delta * bsum; /* :111 */
                    SSD += ex * ex;                                                      /* :121 */
                }
            }
            /* det3x3symm :12-24 */
            float det = a * d * f + 2.0f * b * c * e_;
            det -= (a * e_ * e_ + b * b * f) + c * c * d;
            float rcp = 1.0f / det; /* :126 */
            /* adjoint3x3symm :26-40 */
            float A_ = d * f - e_ * e_, B_ = c * e_ - b * f, C_ = b * e_ - c * d;
            float D_ = a * f - c * c, E_ = b * c - a * e_, F_ = a * d - b * b;
            float dX = (A_ * r0 + B_ * r1) + C_ * r2; /* :132-134 */
            float dY = (B_ * r0 + D_ * r1) + E_ * r2;
            float dZ = (C_ * r0 + E_ * r1) + F_ * r2;
            dX *= rcp;
            dY *= rcp;
            dZ *= rcp;
            X1x += dX; /* :137 */
            X1y += dY;
            float ux = dX * whx, uy = dY * why; /* :139-140 */
            float sqrLen = ux * ux + uy * uy;
            invalid = invalid || (det < 0.00001f);                                                 /* :142 */
            invalid = invalid || (SSD > ssdThr);                                                   /* :143 */
            invalid = invalid || (sqrLen > sqrConvThr);                                            /* :144 */
            invalid = invalid || (X1x < vr[0] || X1y < vr[1]) || (X1x > vr[2] || X1y > vr[3]);     /* :145 */
            note_margin(k, det, SSD, ssdThr, sqrLen, sqrConvThr, X1x, X1y, vr);
            float nb = beta + dZ;
            if (invalid || !(X1x == X1x) || !(X1y == X1y) || !(nb == nb)) {
                featOut[3 * k] = featOut[3 * k + 1] = featOut[3 * k + 2] = -1.0f;
            } else {
                featOut[3 * k] = X1x;
                featOut[3 * k + 1] = X1y;
                featOut[3 * k + 2] = nb; /* :147 */
            }
        }
    }
}

/* ---- the same pass with the window sums taken in the GPU's fixed order ("tree" mode) -----------------------
 * The shader accumulates its ten window sums serially over all (2hw+1)^2 pixels; the HIP tracker
 * (coslam_amd/csrc/klt_track_rows.hip) gives every window ROW to one lane -- which sums its pixels serially, left to
 * right, starting from 0.0f, exactly like the shader's inner loop -- and folds the rows of a feature in a fixed tree:
 * ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), rows beyond the window contributing 0.0f (windows wider than 7:
 * that tree over rows 0..7 plus the same tree over rows 8..15), and adds the neighbour term of the third right-hand
 * side once, as nPix * (delta * bsum), instead of delta * bsum per pixel (klt_tracker_with_gain.cg:111).  Same
 * mathematics, different binary32 rounding: in this mode the oracle and the HIP path must agree BIT FOR BIT, which turns
 * every statistical status/position comparison of the tracking tests into an exact one.  hw must be 1..7. */
static float fold8(const float* v) { return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])); }
static float fold_rows(const float* v, int lpf) { return lpf == 8 ? fold8(v) : fold8(v) + fold8(v + 8); }

void okl_track_gain_pass_tree(const uint16_t* pyr0, const uint16_t* pyr1, int W, int H, int nLevels, int level, int fw,
                              int fh, int hw, const float* feat0, const float* featIn, float* featOut, float sqrConvThr,
                              float ssdThr, const float vr[4], float lambda, float delta) {
    int64_t off[OKL_MAX_LEVELS];
    okl_pyr_layout(W, H, nLevels, off);
    const uint16_t* L0 = pyr0 + 4 * off[level];
    const uint16_t* L1 = pyr1 + 4 * off[level];
    const int Wl = W >> level, Hl = H >> level;
    const float dsx = 1.0f / (float)Wl, dsy = 1.0f / (float)Hl;
    const float whx = (float)W, why = (float)H;
    const double rxy = (double)fh / (double)fw, ryx = (double)fw / (double)fh;
    const int n1x[4] = {1, -1, (int)floor(0.5 + ryx), (int)floor(0.5 - ryx)};
    const int n1y[4] = {(int)floor(0.5 + rxy), (int)floor(0.5 - rxy), 1, -1};
    const int n2x[4] = {1, -1, 0, 0};
    const int n2y[4] = {0, 0, 1, -1};
    const int lpf = (hw <= 3) ? 8 : 16;
    const int nPix = (2 * hw + 1) * (2 * hw + 1);

    for (int j = 0; j < fh; ++j) {
        for (int i = 0; i < fw; ++i) {
            size_t k = (size_t)j * fw + i;
            float X0x = feat0[3 * k], X0y = feat0[3 * k + 1];
            float X1x = featIn[3 * k], X1y = featIn[3 * k + 1], beta = featIn[3 * k + 2];
            if ((X1x < 0) || (X0x < 0)) {
                featOut[3 * k] = featOut[3 * k + 1] = featOut[3 * k + 2] = -1.0f;
                continue;
            }
            float bsum;
            {
                float t4[4];
                for (int q = 0; q < 4; ++q) {
                    float b1 = slot_beta(featIn, fw, fh, i + n1x[q], j + n1y[q]);
                    float b2 = slot_beta(featIn, fw, fh, i + n2x[q], j + n2y[q]);
                    if (b1 < 0) b1 = beta;
                    if (b2 < 0) b2 = beta;
                    t4[q] = (b1 + b2) - 2.0f * beta;
                }
                bsum = ((t4[0] + t4[1]) + t4[2]) + t4[3];
            }
            float ra[16] = {0}, rb[16] = {0}, rc[16] = {0}, rd[16] = {0}, re[16] = {0}, rf[16] = {0};
            float rr0[16] = {0}, rr1[16] = {0}, rr2[16] = {0}, rss[16] = {0};
            for (int y = -hw; y <= hw; ++y) {
                const int row = y + hw;
                float st_y = X0y + (float)y * dsy, st_w = X1y + (float)y * dsy;
                float a = 0, b = 0, c = 0, d = 0, e_ = 0, f = 0, r0 = 0, r1 = 0, r2s = 0, ssd = 0;
                for (int x = -hw; x <= hw; ++x) {
                    float st_x = X0x + (float)x * dsx, st_z = X1x + (float)x * dsx;
                    float I0[3], I1[3];
                    okl_sample(L0, Wl, Hl, st_x, st_y, I0);
                    okl_sample(L1, Wl, Hl, st_z, st_w, I1);
                    float ex = beta * I0[0] - I1[0];
                    float gx = (beta * I0[1] + I1[1]) * whx / 2.0f;
                    float gy = (beta * I0[2] + I1[2]) * why / 2.0f;
                    float m0 = sqrtf(I0[1] * I0[1] + I0[2] * I0[2]);
                    float m1 = sqrtf(I1[1] * I1[1] + I1[2] * I1[2]);
                    a += gx * gx;
                    b += gx * gy;
                    c += gx * (-I0[0]);
                    d += gy * gy;
                    e_ += gy * (-I0[0]);
                    f += (I0[0] * I0[0] + lambda * m0 * m0) + delta * 8.0f;
                    r0 += ex * gx;
                    r1 += ex * gy;
                    r2s += -ex * I0[0] + lambda * m0 * (m1 - beta * m0);
                    ssd += ex * ex;
                }
                ra[row] = a, rb[row] = b, rc[row] = c, rd[row] = d, re[row] = e_, rf[row] = f;
                rr0[row] = r0, rr1[row] = r1, rr2[row] = r2s, rss[row] = ssd;
            }
            const float a = fold_rows(ra, lpf), b = fold_rows(rb, lpf), c = fold_rows(rc, lpf), d = fold_rows(rd, lpf);
            const float e_ = fold_rows(re, lpf), f = fold_rows(rf, lpf), r0 = fold_rows(rr0, lpf), r1 = fold_rows(rr1, lpf);
            const float r2s = fold_rows(rr2, lpf), SSD = fold_rows(rss, lpf);
            const float r2 = r2s + (float)nPix * (delta * bsum);
            float det = a * d * f + 2.0f * b * c * e_;
            det -= (a * e_ * e_ + b * b * f) + c * c * d;
            float rcp = 1.0f / det;
            float A_ = d * f - e_ * e_, B_ = c * e_ - b * f, C_ = b * e_ - c * d;
            float D_ = a * f - c * c, E_ = b * c - a * e_, F_ = a * d - b * b;
            float dX = ((A_ * r0 + B_ * r1) + C_ * r2) * rcp;
            float dY = ((B_ * r0 + D_ * r1) + E_ * r2) * rcp;
            float dZ = ((C_ * r0 + E_ * r1) + F_ * r2) * rcp;
            X1x += dX;
            X1y += dY;
            float ux = dX * whx, uy = dY * why;
            float sqrLen = ux * ux + uy * uy;
            int invalid = (det < 0.00001f);
            invalid = invalid || (SSD > ssdThr);
            invalid = invalid || (sqrLen > sqrConvThr);
            invalid = invalid || (X1x < vr[0] || X1y < vr[1]) || (X1x > vr[2] || X1y > vr[3]);
            note_margin(k, det, SSD, ssdThr, sqrLen, sqrConvThr, X1x, X1y, vr);
            float nb = beta + dZ;
            if (invalid || !(X1x == X1x) || !(X1y == X1y) || !(nb == nb)) {
                featOut[3 * k] = featOut[3 * k + 1] = featOut[3 * k + 2] = -1.0f;
            } else {
                featOut[3 * k] = X1x;
                featOut[3 * k + 1] = X1y;
                featOut[3 * k + 2] = nb;
            }
        }
    }
}

/* ------------------------------------------------------------------ detector */

void okl_cornerness(const uint16_t* lvl0, int W, int H, float minCornerness, float margin, float* out) {
    float* conv = (float*)malloc(sizeof(float) * 3 * (size_t)W * H);
    /* klt_detector_pass1.cg: 7 vertical taps -3..+3 (renderQuad8Tap(0,1/H), v3d_gpuklt.cpp:464) */
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            float r0 = 0, r1 = 0, r2 = 0;
            for (int k = -3; k <= 3; ++k) {
                const uint16_t* p = lvl0 + 4 * ((size_t)clampi(y + k, 0, H - 1) * W + x);
                float Ix = okl_f16_to_f32(p[1]), Iy = okl_f16_to_f32(p[2]);
                r0 += Ix * Ix;
                r1 += Ix * Iy;
                r2 += Iy * Iy;
            }
            float* c = conv + 3 * ((size_t)y * W + x);
            c[0] = r0;
            c[1] = r1;
            c[2] = r2;
        }
    }
    /* klt_detector_pass2.cg:12-33, validRegion v3d_gpuklt.cpp:470 */
    const float lox = margin / (float)W, loy = margin / (float)H;
    const float hix = 1.0f - margin / (float)W, hiy = 1.0f - margin / (float)H;
    for (int y = 0; y < H; ++y) {
        float sty = ((float)y + 0.5f) / (float)H;
        for (int x = 0; x < W; ++x) {
            float a = 0, b = 0, c = 0;
            for (int k = -3; k <= 3; ++k) {
                const float* p = conv + 3 * ((size_t)y * W + clampi(x + k, 0, W - 1));
                a += p[0];
                b += p[1];
                c += p[2];
            }
            float amc = a - c;
            float cn = 0.5f * ((a + c) - sqrtf(amc * amc + 4.0f * (b * b)));
            cn = fmaxf(cn - minCornerness, 0.0f);
            float stx = ((float)x + 0.5f) / (float)W;
            int inside = (stx >= lox && sty >= loy) && (stx <= hix && sty <= hiy);
            out[(size_t)y * W + x] = inside ? cn : 0.0f;
        }
    }
    free(conv);
}

void okl_suppress_present(float* corner, int W, int H, int nPresent, const float* p3) {
    /* GL_POINTS of size 1 at (x,y) in the normalized projection (v3d_gpuklt.cpp:475-500):
     * the fragment is the pixel containing the point; points outside [0,1) are clipped */
    for (int k = 0; k < nPresent; ++k) {
        float s = p3[3 * k], t = p3[3 * k + 1];
        if (!(s >= 0.0f && t >= 0.0f)) continue;
        float fx = floorf(s * (float)W), fy = floorf(t * (float)H);
        if (fx >= (float)W || fy >= (float)H) continue;
        corner[(size_t)(int)fy * W + (int)fx] = -1e30f;
    }
}

void okl_nonmax(float* corner, int W, int H, int d) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)W * H);
    /* klt_detector_nonmax.cg:12-26 with ds=(1/W,0) then (0,1/H) (v3d_gpuklt.cpp:503-512) */
    for (int y = 0; y < H; ++y) {
        const float* row = corner + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            float m = row[x];
            for (int i = -d; i < 0; ++i) {
                float cn = fabsf(row[clampi(x + i, 0, W - 1)]);
                m = (cn >= fabsf(m)) ? (-cn) : m;
            }
            for (int i = 1; i <= d; ++i) {
                float cn = fabsf(row[clampi(x + i, 0, W - 1)]);
                m = (cn >= fabsf(m)) ? (-cn) : m;
            }
            tmp[(size_t)y * W + x] = m;
        }
    }
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            float m = tmp[(size_t)y * W + x];
            for (int i = -d; i < 0; ++i) {
                float cn = fabsf(tmp[(size_t)clampi(y + i, 0, H - 1) * W + x]);
                m = (cn >= fabsf(m)) ? (-cn) : m;
            }
            for (int i = 1; i <= d; ++i) {
                float cn = fabsf(tmp[(size_t)clampi(y + i, 0, H - 1) * W + x]);
                m = (cn >= fabsf(m)) ? (-cn) : m;
            }
            corner[(size_t)y * W + x] = m;
        }
    }
    free(tmp);
}

static inline uint32_t part1by1(uint32_t v) {
    v &= 0xffffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
static inline uint32_t morton2(uint32_t x, uint32_t y) { return part1by1(x) | (part1by1(y) << 1); }

typedef struct {
    uint32_t key;
    float x, y, c;
} cand_t;

static int cmp_key(const void* a, const void* b) {
    uint32_t ka = ((const cand_t*)a)->key, kb = ((const cand_t*)b)->key;
    return (ka > kb) - (ka < kb);
}
/* cornerness descending, ties by HistoPyramid order (the std::sort branch of v3d_gpuklt.cpp:704-708,763-767
 * made total) */
static int cmp_corner(const void* a, const void* b) {
    const cand_t *A = (const cand_t*)a, *B = (const cand_t*)b;
    if (A->c > B->c) return -1;
    if (A->c < B->c) return 1;
    return (A->key > B->key) - (A->key < B->key);
}

/* Equivalent of discriminator (klt_detector_discriminator.cg) + build_histpyr + traverse_histpyr.cg:
 * the k-th emitted element is the k-th survivor in the quadrant order (--,+-,-+,++) at every
 * HistoPyramid level, i.e. Morton order of the pixel with x as the minor bit. */
static int extract_cands(const float* corner, int W, int H, cand_t** out) {
    int n = 0, cap = 1024;
    cand_t* v = (cand_t*)malloc(sizeof(cand_t) * cap);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float c = corner[(size_t)y * W + x];
            if (c > 0.0f) {
                if (n == cap) {
                    cap *= 2;
                    v = (cand_t*)realloc(v, sizeof(cand_t) * cap);
                }
                v[n].key = morton2((uint32_t)x, (uint32_t)y);
                v[n].x = ((float)x + 0.5f) / (float)W; /* traverse_histpyr.cg:85-86 */
                v[n].y = ((float)y + 0.5f) / (float)H;
                v[n].c = c;
                n++;
            }
        }
    qsort(v, n, sizeof(cand_t), cmp_key);
    *out = v;
    return n;
}

int okl_extract(const float* corner, int W, int H, int maxOut, float* list3) {
    cand_t* v;
    int n = extract_cands(corner, W, H, &v);
    int m = n < maxOut ? n : maxOut;
    for (int i = 0; i < m; ++i) {
        list3[3 * i] = v[i].x;
        list3[3 * i + 1] = v[i].y;
        list3[3 * i + 2] = v[i].c;
    }
    free(v);
    return n;
}

/* ------------------------------------------------------------------ sequence tracker */

struct okl_seq {
    okl_config cfg;
    int centered;
    int W, H, L, fw, fh, plw, plh, N;
    float margin, convThr, ssdThr; /* tracker thresholds (KLT_TrackerBase) */
    float detMargin;               /* KLT_Detector::_margin, 10 unless setBorderMargin (v3d_gpuklt.h:114) */
    uint16_t* pyr[2];
    int p0, p1; /* _pyrCreator0/_pyrCreator1 */
    float* fb[3];
    int b0, b1, b2; /* _featuresBuffer0/1/2 */
    float* corner;
    cand_t* corners; /* host _corners, plw*plh */
    int sum_mode;    /* 0: the shader's serial window sums; 1: the HIP tracker's fixed tree (okl_track_gain_pass_tree) */
};

okl_seq* okl_seq_create(const okl_config* cfg, int centered) {
    okl_seq* s = (okl_seq*)calloc(1, sizeof(okl_seq));
    s->cfg = *cfg;
    s->centered = centered;
    return s;
}

/* deep copy (test support: two summation orders from IDENTICAL state, frame after frame) */
okl_seq* okl_seq_clone(const okl_seq* src) {
    okl_seq* s = (okl_seq*)malloc(sizeof(okl_seq));
    *s = *src;
    size_t tex = okl_pyr_layout(src->W, src->H, src->L, NULL);
    for (int i = 0; i < 2; ++i) {
        s->pyr[i] = (uint16_t*)malloc(tex * 4 * sizeof(uint16_t));
        memcpy(s->pyr[i], src->pyr[i], tex * 4 * sizeof(uint16_t));
    }
    for (int i = 0; i < 3; ++i) {
        s->fb[i] = (float*)malloc(sizeof(float) * 3 * src->N);
        memcpy(s->fb[i], src->fb[i], sizeof(float) * 3 * src->N);
    }
    s->corner = (float*)malloc((size_t)src->W * src->H * sizeof(float));
    memcpy(s->corner, src->corner, (size_t)src->W * src->H * sizeof(float));
    size_t nc = (size_t)src->plw * src->plh + src->N;
    s->corners = (cand_t*)malloc(nc * sizeof(cand_t));
    memcpy(s->corners, src->corners, nc * sizeof(cand_t));
    return s;
}

void okl_seq_destroy(okl_seq* s) {
    if (!s) return;
    free(s->pyr[0]);
    free(s->pyr[1]);
    for (int i = 0; i < 3; ++i) free(s->fb[i]);
    free(s->corner);
    free(s->corners);
    free(s);
}

void okl_seq_allocate(okl_seq* s, int W, int H, int nLevels, int fw, int fh, int plw, int plh) {
    s->W = W;
    s->H = H;
    s->L = nLevels;
    s->fw = fw;
    s->fh = fh;
    s->plw = plw;
    s->plh = plh;
    s->N = fw * fh;
    /* v3d_gpuklt.cpp:603-619 */
    s->margin = s->cfg.trackBorderMargin;
    s->convThr = s->cfg.convergenceThreshold;
    s->ssdThr = s->cfg.SSD_Threshold;
    s->detMargin = 10.0f;
    size_t tex = okl_pyr_layout(W, H, nLevels, NULL);
    for (int i = 0; i < 2; ++i) s->pyr[i] = (uint16_t*)calloc(tex * 4, sizeof(uint16_t));
    s->p0 = 0;
    s->p1 = 1;
    for (int i = 0; i < 3; ++i) {
        s->fb[i] = (float*)malloc(sizeof(float) * 3 * s->N);
        for (int k = 0; k < 3 * s->N; ++k) s->fb[i][k] = -1.0f; /* RTT buffers start undefined; we define dead */
    }
    s->b0 = 0;
    s->b1 = 1;
    s->b2 = 2;
    s->corner = (float*)calloc((size_t)W * H, sizeof(float));
    s->corners = (cand_t*)calloc((size_t)plw * plh + s->N, sizeof(cand_t));
}

void okl_seq_set_border_margin(okl_seq* s, float m) { /* v3d_gpuklt.h:219-226 */
    s->margin = m;
    s->detMargin = m;
}
void okl_seq_set_sum_mode(okl_seq* s, int mode) { s->sum_mode = mode; }
void okl_seq_set_convergence_threshold(okl_seq* s, float t) { s->convThr = t; }
void okl_seq_set_ssd_threshold(okl_seq* s, float t) { s->ssdThr = t; }
const uint16_t* okl_seq_cur_pyramid(const okl_seq* s) { return s->pyr[s->p1]; }
const float* okl_seq_cornerness(const okl_seq* s) { return s->corner; }

static void provide(okl_seq* s, const float* list3) {
    if (s->cfg.trackWithGain) { /* v3d_gpuklt.cpp:188-197 */
        memcpy(s->fb[s->b2], list3, sizeof(float) * 3 * s->N);
        memcpy(s->fb[s->b1], list3, sizeof(float) * 3 * s->N);
    } else { /* :86-92 */
        memcpy(s->fb[s->b1], list3, sizeof(float) * 3 * s->N);
    }
}
void okl_seq_read_features(const okl_seq* s, float* out3) {
    const float* src = s->cfg.trackWithGain ? s->fb[s->b2] : s->fb[s->b1]; /* :94-97,199-203 */
    memcpy(out3, src, sizeof(float) * 3 * s->N);
}

void okl_seq_advance(okl_seq* s) { /* v3d_gpuklt.h:252-259 */
    int t = s->b0;
    s->b0 = s->b1;
    s->b1 = t;
    t = s->p0;
    s->p0 = s->p1;
    s->p1 = t;
}

static void run_tracker(okl_seq* s) {
    const okl_config* c = &s->cfg;
    const int hw = c->windowWidth / 2;
    const uint16_t *P0 = s->pyr[s->p0], *P1 = s->pyr[s->p1];
    if (!c->trackWithGain) {
        /* host passes -DNITERATIONS, the shader reads N_ITERATIONS => always 5 (v3d_gpuklt.cpp:108, klt_tracker.cg:16-18) */
        okl_set_nogain_sum_mode(s->sum_mode == 1);
        okl_track_nogain(P0, P1, s->W, s->H, s->L, c->levelSkip, hw, 5, s->margin, s->convThr, s->ssdThr, s->N,
                         s->fb[s->b0], s->fb[s->b1]);
        return;
    }
    /* v3d_gpuklt.cpp:223-227: blue channel of buffer0 cleared to 1 */
    for (int k = 0; k < s->N; ++k) s->fb[s->b0][3 * k + 2] = 1.0f;
    float delta = 200.0f;
    const float tau = 1.0f;
    int levelSkip = c->levelSkip > 0 ? c->levelSkip : (c->nLevels - 1); /* v3d_gpuklt.h:14 */
    if (levelSkip <= 0) levelSkip = 1;
    float sqrConv = 1000000.0f, ssd = 1000000.0f;
    float vr[4] = {-1.0f, -1.0f, 2.0f, 2.0f};
    for (int level = s->L - 1; level >= 0; level -= levelSkip) { /* :254 */
        for (int iter = 1; iter <= c->nIterations; ++iter) {      /* :268 */
            float dcur = delta;
            delta *= tau;
            if (iter == 1) { /* :271-279 */
                sqrConv = 1000000.0f;
                ssd = 1000000.0f;
                vr[0] = vr[1] = -1.0f;
                vr[2] = vr[3] = 2.0f;
            } else if (iter == c->nIterations) {
                sqrConv = s->convThr * s->convThr;
                ssd = s->ssdThr;
                vr[0] = s->margin / (float)s->W;
                vr[1] = s->margin / (float)s->H;
                vr[2] = 1.0f - s->margin / (float)s->W;
                vr[3] = 1.0f - s->margin / (float)s->H;
            }
            if (s->sum_mode == 1 && hw >= 1 && hw <= 7)
                okl_track_gain_pass_tree(P0, P1, s->W, s->H, s->L, level, s->fw, s->fh, hw, s->fb[s->b2], s->fb[s->b0],
                                         s->fb[s->b1], sqrConv, ssd, vr, 1.0f, dcur);
            else
                okl_track_gain_pass(P0, P1, s->W, s->H, s->L, level, s->fw, s->fh, hw, s->fb[s->b2], s->fb[s->b0],
                                    s->fb[s->b1], sqrConv, ssd, vr, 1.0f, dcur);
            int t = s->b0;
            s->b0 = s->b1;
            s->b1 = t; /* :285 */
        }
    }
    int t = s->b0;
    s->b0 = s->b2;
    s->b2 = t; /* :304 */
}

void okl_seq_track(okl_seq* s, const uint8_t* img, int* nPresent, okl_tracked_feature* dest) {
    okl_pyramid_build(img, s->W, s->H, s->L, s->centered, s->pyr[s->p1]); /* :858 */
    run_tracker(s);
    float* r = (float*)malloc(sizeof(float) * 3 * s->N);
    okl_seq_read_features(s, r);
    int n = 0;
    for (int i = 0; i < s->N; ++i) { /* :872-888 */
        float X = r[3 * i], Y = r[3 * i + 1], g = r[3 * i + 2];
        if (X >= 0) {
            dest[i].status = 0;
            dest[i].pos[0] = X;
            dest[i].pos[1] = Y;
            dest[i].gain = g;
            dest[i].fed = -1;
            ++n;
        } else {
            dest[i].status = -1;
            dest[i].fed = -1;
        }
    }
    free(r);
    *nPresent = n;
}

/* detectCorners + extractCorners + top-K (v3d_gpuklt.cpp:423-588 and callers). Returns the number kept,
 * candidates in s->corners[0..kept). */
static int detect_and_select(okl_seq* s, int nPresent, const float* present3, int maxKeep) {
    int64_t off[OKL_MAX_LEVELS];
    okl_pyr_layout(s->W, s->H, s->L, off);
    okl_cornerness(s->pyr[s->p1] + 4 * off[0], s->W, s->H, s->cfg.minCornerness, s->detMargin, s->corner);
    if (nPresent > 0) okl_suppress_present(s->corner, s->W, s->H, nPresent, present3);
    okl_nonmax(s->corner, s->W, s->H, s->cfg.minDistance);
    cand_t* v;
    int n = extract_cands(s->corner, s->W, s->H, &v);
    int cap = s->plw * s->plh;
    if (n > cap) n = cap; /* :659,701,756 */
    if (n > maxKeep) {
        qsort(v, n, sizeof(cand_t), cmp_corner);
        n = maxKeep < 0 ? 0 : maxKeep;
    }
    memcpy(s->corners, v, sizeof(cand_t) * n);
    free(v);
    return n;
}

void okl_seq_detect(okl_seq* s, const uint8_t* img, int* nDetected, okl_tracked_feature* dest) {
    okl_pyramid_build(img, s->W, s->H, s->L, s->centered, s->pyr[s->p1]); /* :697 */
    int n = detect_and_select(s, 0, NULL, s->N);
    float* list = (float*)malloc(sizeof(float) * 3 * s->N);
    for (int i = 0; i < s->N; ++i) {
        if (i < n) {
            list[3 * i] = s->corners[i].x;
            list[3 * i + 1] = s->corners[i].y;
            list[3 * i + 2] = s->cfg.trackWithGain ? 1.0f : s->corners[i].c; /* :716-721 */
        } else { /* the reference uploads stale host memory here; we define dead slots */
            list[3 * i] = list[3 * i + 1] = -1.0f;
            list[3 * i + 2] = s->cfg.trackWithGain ? 1.0f : -1.0f;
        }
    }
    provide(s, list);
    for (int i = 0; i < n; ++i) { /* :723-729 */
        dest[i].status = 1;
        dest[i].pos[0] = list[3 * i];
        dest[i].pos[1] = list[3 * i + 1];
        dest[i].gain = list[3 * i + 2];
        dest[i].fed = -1;
    }
    for (int i = n; i < s->N; ++i) {
        dest[i].status = -1;
        dest[i].fed = -1;
    }
    free(list);
    *nDetected = n;
}

void okl_seq_detect_present(okl_seq* s, const uint8_t* img, int* nDetected, okl_tracked_feature* dest, int nPresent,
                            const float* present3) { /* :650-691 */
    okl_pyramid_build(img, s->W, s->H, s->L, s->centered, s->pyr[s->p1]);
    int n = detect_and_select(s, nPresent, present3, s->N - nPresent);
    float* list = (float*)malloc(sizeof(float) * 3 * s->N);
    for (int i = 0; i < s->N; ++i) {
        if (i < n) {
            list[3 * i] = s->corners[i].x;
            list[3 * i + 1] = s->corners[i].y;
            list[3 * i + 2] = s->cfg.trackWithGain ? 1.0f : s->corners[i].c;
        } else if (i < n + nPresent) { /* :667-670 */
            list[3 * i] = present3[3 * (i - n)];
            list[3 * i + 1] = present3[3 * (i - n) + 1];
            list[3 * i + 2] = s->cfg.trackWithGain ? 1.0f : 0.0f;
        } else {
            list[3 * i] = list[3 * i + 1] = -1.0f;
            list[3 * i + 2] = s->cfg.trackWithGain ? 1.0f : -1.0f;
        }
    }
    provide(s, list);
    for (int i = 0; i < n + nPresent && i < s->N; ++i) { /* :679-685 */
        dest[i].status = 1;
        dest[i].pos[0] = list[3 * i];
        dest[i].pos[1] = list[3 * i + 1];
        dest[i].gain = list[3 * i + 2];
        dest[i].fed = i >= n ? i - n : -1;
    }
    for (int i = n + nPresent; i < s->N; ++i) dest[i].status = -1;
    free(list);
    *nDetected = n + nPresent;
}

void okl_seq_redetect(okl_seq* s, const uint8_t* img, int* nNew, okl_tracked_feature* dest) { /* :737-805 */
    int nPresent = 0;
    okl_seq_track(s, img, &nPresent, dest);
    float* list = (float*)malloc(sizeof(float) * 3 * s->N);
    for (int i = 0; i < s->N; ++i) { /* :744-752 */
        if (dest[i].status >= 0) {
            list[3 * i] = dest[i].pos[0];
            list[3 * i + 1] = dest[i].pos[1];
        } else {
            list[3 * i] = list[3 * i + 1] = -1.0f;
        }
        list[3 * i + 2] = 0.0f;
    }
    int n = detect_and_select(s, s->N, list, s->N - nPresent);
    int k = 0;
    for (int i = 0; i < s->N && k < n; ++i) { /* :775-786 */
        if (dest[i].status < 0) {
            dest[i].status = 1;
            dest[i].pos[0] = s->corners[k].x;
            dest[i].pos[1] = s->corners[k].y;
            dest[i].gain = s->corners[k].c;
            dest[i].fed = -1;
            ++k;
        }
    }
    for (int i = 0; i < s->N; ++i) { /* :787-797 */
        if (dest[i].status >= 0) {
            list[3 * i] = dest[i].pos[0];
            list[3 * i + 1] = dest[i].pos[1];
        } else {
            list[3 * i] = list[3 * i + 1] = -1.0f;
        }
        list[3 * i + 2] = 1.0f;
    }
    provide(s, list);
    free(list);
    *nNew = n + nPresent;
}

void okl_seq_feed(okl_seq* s, int npts, const float* featPts, int* trackIds, int* nFed) { /* :808-855 */
    float* c = (float*)malloc(sizeof(float) * 3 * s->N);
    okl_seq_read_features(s, c);
    const double radius2 = 1e-4;
    for (int k = 0; k < npts; ++k) {
        for (int i = 0; i < s->N; ++i) {
            if (c[3 * i] < 0) continue;
            double dx = (double)(featPts[2 * k] - c[3 * i]); /* stride-2 read of a stride-3 array: reference quirk */
            double dy = (double)(featPts[2 * k + 1] - c[3 * i + 1]);
            if (dx * dx + dy * dy < radius2) c[3 * i] = -1.0f;
        }
    }
    int k = 0;
    for (int i = 0; i < s->N && k < npts; ++i) {
        if (c[3 * i] < 0) {
            c[3 * i] = featPts[3 * k];
            c[3 * i + 1] = featPts[3 * k + 1];
            c[3 * i + 2] = 1.0f;
            trackIds[k] = i;
            ++k;
        }
    }
    *nFed = k;
    provide(s, c);
    free(c);
}
