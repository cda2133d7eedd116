// klt_track_rows.hip -- Lucas-Kanade with gain, several features per wave and several cameras per launch, gfx950.
//
// Replaces KLT_TrackerWithGain::trackFeaturesAndGain (src/tracking/CGKLT/v3d_gpuklt.cpp:205-305,
// Shaders/klt_tracker_with_gain.cg:42-148) for windows up to 15 x 15, for every camera of a group in ONE launch
// (the reference tracks its cameras one after the other: CoSLAM::featureTracking, src/app/SL_CoSLAM.cpp:299-305).
//
// Layout.  A feature owns LPF = 8 lanes of a wave (16 for windows wider than 7): lane r of the group works window ROW
// r -- its FW = 2 hw + 1 pixels serially, left to right, exactly as the shader's inner loop does -- so a wave carries
// 8 (4) features, a camera with 2000 slots needs 250 waves, and eight cameras are 2000 waves: two per SIMD of the
// chip, all co-resident, where one wave per feature (klt_track.hip) needed 16000.  The per-feature work that does not
// scale with the window (the folds of the nine window sums, the 3 x 3 adjugate solve, the validity tests, the hand-off)
// is now shared by 8 features per instruction instead of 1, and the folds are three DPP steps inside a 16-lane row
// (quad_perm, quad_perm, row_half_mirror; + row_mirror for LPF 16): no cross-row traffic at all.
//
// Summation order (the test oracle's "tree" mode, okl_track_gain_pass_tree, mirrors it bit for bit):
//   row sums      serial over the row's pixels, starting from 0.0f (== the shader's inner `for x`);
//   window sums   ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), rows >= FW contribute 0.0f
//                 (LPF 16: that tree over rows 0..7 plus the same tree over rows 8..15);
//   third rhs     r2 = r2s + nPix * (delta * bsum): the shader adds delta * bsum once per pixel
//                 (klt_tracker_with_gain.cg:111); the neighbour term enters once, after the hand-off.
//
// Schedules.  k_track_rows_fused: ONE persistent launch for all levels x iterations of all cameras; neighbouring
// slots hand their gain over through 8-byte {tag, beta} granules exactly as in klt_track.hip (one row per pass,
// frame-unique tags, `sc1` stores, L1-bypassing polls, bounded spins).  A wave's eight slots are contiguous, so a
// pass publishes ONE 64-byte line per wave.  k_track_rows_pass: one launch per Gauss-Newton pass (the reference's
// schedule, v3d_gpuklt.cpp:254-287), same arithmetic, bit-identical results; used when the grid cannot be co-resident.
#include "klt_internal.h"

#pragma clang fp contract(off)

namespace {

#ifndef CS_ROWS_UNROLL
#define CS_ROWS_UNROLL 7
#endif
#ifndef CS_ROWS_WPB
// Waves per workgroup.  4: a workgroup's waves go to the CU's four SIMDs (53 KB of LDS per workgroup for a 7 x 7 window: two per
// CU), so eight cameras (2000 waves) land as two waves on every SIMD.  With single-wave workgroups the
// dispatcher piles up to ten of them on a CU while others idle: 218 vs 164 us for the eight-camera tracker stage.
#define CS_ROWS_WPB 4
#endif
#ifndef CS_ROWS_MINBLOCKS
#define CS_ROWS_MINBLOCKS 3  // resident workgroups per CU the 8-lanes-per-feature instantiations are compiled for
#endif
#ifndef CS_ROWS_PRIO_BASE
#define CS_ROWS_PRIO_BASE 1  // s_setprio while a wave samples its patch
#define CS_ROWS_PRIO_HOT 3   // ... from the first poll to the publish
#endif
#ifndef CS_ROWS_LDS_PAD
#define CS_ROWS_LDS_PAD 0  // extra dynamic LDS per wave (bytes): caps the workgroups a CU admits
#endif
#ifndef CS_ROWS_MARGIN
// Texels of slack around the window footprint in the LDS patch.  1 (round 5; 2 before): a 7 x 7 window's wave then holds 13.25 KB
// of LDS instead of 16.  A CU still holds TWO of these workgroups (64 per XCD: measured, tools/r05_diag.py -- the occupancy query's
// "three" does not materialise), but beside them 52 KB of its LDS stay free instead of 32: the pose stream's kernels find room on
// a CU that carries two tracker workgroups, which is what the camera-per-XCD placement (xcdsPerCam: 63 of an XCD's 64 slots) needs
// beside them (with 16 KB per wave the placed grid met a co-residency timeout in the frame loop; profiles/r05_ab_runs.txt).
// Re-centring got no more frequent in the counters (FETCH_SIZE 52.7 vs 54.7 MB per launch) and the launch 1.6 % shorter.
#define CS_ROWS_MARGIN 1
#endif
constexpr int RW_MARGIN = CS_ROWS_MARGIN;
typedef unsigned long long cs_granule;
typedef __attribute__((address_space(1))) cs_granule gu64;

__device__ __forceinline__ cs_granule gran_load(const cs_granule* p) {
    return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_store(cs_granule* p, unsigned tag, float beta) {
    __hip_atomic_store((gu64*)p, ((cs_granule)tag << 32) | (cs_granule)__float_as_uint(beta), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// all-reduce over the LPF lanes of a feature group; every lane ends with the same bits (fp add commutes)
template <int LPF>
__device__ __forceinline__ float rows_fold(float v) {
    v += cs_dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += cs_dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += cs_dpp_f<0x141, 0xf>(v);  // row_half_mirror
    if (LPF == 16) v += cs_dpp_f<0x140, 0xf>(v);  // row_mirror
    return v;
}

// floor of the (clamped) texel coordinate the bilinear fetch computes for normalised coordinate s
__device__ __forceinline__ int fp_floor(float s, int Wl) {
    float u = s * (float)Wl - 0.5f;
    u = fminf(fmaxf(u, -2.0f), (float)Wl + 1.0f);
    return (int)floorf(u);
}

// GL_LINEAR + CLAMP_TO_EDGE fetch straight from the level (same arithmetic as oracle okl_sample())
__device__ __forceinline__ void sample_g(const cs_texel* __restrict__ lvl, int Wl, int Hl, float s, float t, float& I,
                                         float& Ix, float& Iy) {
    float u = s * (float)Wl - 0.5f;
    float v = t * (float)Hl - 0.5f;
    u = fminf(fmaxf(u, -2.0f), (float)Wl + 1.0f);
    v = fminf(fmaxf(v, -2.0f), (float)Hl + 1.0f);
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = cs_clampi((int)fu, 0, Wl - 1), i1 = cs_clampi((int)fu + 1, 0, Wl - 1);
    int j0 = cs_clampi((int)fv, 0, Hl - 1), j1 = cs_clampi((int)fv + 1, 0, Hl - 1);
    cs_texel t00 = lvl[(size_t)j0 * Wl + i0], t10 = lvl[(size_t)j0 * Wl + i1];
    cs_texel t01 = lvl[(size_t)j1 * Wl + i0], t11 = lvl[(size_t)j1 * Wl + i1];
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float I00, X00, Y00, I10, X10, Y10, I01, X01, Y01, I11, X11, Y11;
    cs_unpack_texel(t00, I00, X00, Y00);
    cs_unpack_texel(t10, I10, X10, Y10);
    cs_unpack_texel(t01, I01, X01, Y01);
    cs_unpack_texel(t11, I11, X11, Y11);
    I = ((w00 * I00 + w10 * I10) + w01 * I01) + w11 * I11;
    Ix = ((w00 * X00 + w10 * X10) + w01 * X01) + w11 * X11;
    Iy = ((w00 * Y00 + w10 * Y10) + w01 * Y01) + w11 * Y11;
}

// The same fetch out of the group's LDS patch: cell (lx, ly) holds the texel at (clamp(rx0 + lx), clamp(ry0 + ly)), so
// unclamped footprint indices minus the patch origin address it and CLAMP_TO_EDGE is already folded in.
template <int R>
__device__ __forceinline__ void sample_p(const cs_texel* patch, int rx0, int ry0, int Wl, int Hl, float s, float t, float& I,
                                         float& Ix, float& Iy) {
    float u = s * (float)Wl - 0.5f;
    float v = t * (float)Hl - 0.5f;
    u = fminf(fmaxf(u, -2.0f), (float)Wl + 1.0f);
    v = fminf(fmaxf(v, -2.0f), (float)Hl + 1.0f);
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    const cs_texel* c = patch + ((int)fv - ry0) * R + ((int)fu - rx0);
    cs_texel t00 = c[0], t10 = c[1], t01 = c[R], t11 = c[R + 1];
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float I00, X00, Y00, I10, X10, Y10, I01, X01, Y01, I11, X11, Y11;
    cs_unpack_texel(t00, I00, X00, Y00);
    cs_unpack_texel(t10, I10, X10, Y10);
    cs_unpack_texel(t01, I01, X01, Y01);
    cs_unpack_texel(t11, I11, X11, Y11);
    I = ((w00 * I00 + w10 * I10) + w01 * I01) + w11 * I11;
    Ix = ((w00 * X00 + w10 * X10) + w01 * X01) + w11 * X11;
    Iy = ((w00 * Y00 + w10 * Y10) + w01 * Y01) + w11 * Y11;
}

// Packed FP32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE binary32 operations per instruction, same rounding as the scalar
// forms): the two gradient channels travel as one register pair through the bilinear blend and the window sums, the four
// bilinear weights as two pairs.  Operation for operation the arithmetic of sample_p / the shader's inner loop --
// bit-identical results, ~20 % fewer vector instructions per window pixel.
typedef float cs_f2 __attribute__((ext_vector_type(2)));

// w * (binary16 half of a packed texel word) in ONE instruction: v_fma_mix_f32 reads the half directly (op_sel_hi marks the
// source as f16, op_sel picks the upper half) and the addend neg(0) = -0.0 leaves the correctly rounded product, sign of zero
// included -- the same bits as v_cvt_f32_f16 + v_mul_f32, one instruction instead of two.
__device__ __forceinline__ float mixmul_lo(float w, unsigned packed) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, neg(0) op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed));
    return r;
}
__device__ __forceinline__ float mixmul_hi(float w, unsigned packed) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, neg(0) op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(packed));
    return r;
}

template <int R>
__device__ __forceinline__ void sample_p2(const cs_texel* patch, int rx0, int ry0, int Wl, int Hl, float s, float t, float& I,
                                          cs_f2& G) {
    float u = s * (float)Wl - 0.5f;
    float v = t * (float)Hl - 0.5f;
    u = fminf(fmaxf(u, -2.0f), (float)Wl + 1.0f);
    v = fminf(fmaxf(v, -2.0f), (float)Hl + 1.0f);
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    const cs_texel* c = patch + ((int)fv - ry0) * R + ((int)fu - rx0);
    cs_texel t00 = c[0], t10 = c[1], t01 = c[R], t11 = c[R + 1];
    const cs_f2 wa = {1.0f - a, a};
    const cs_f2 w0 = wa * (1.0f - b), w1 = wa * b;  // (w00, w10), (w01, w11)
    // texel word x = I | Ix << 16, word y = Iy: the twelve weight x channel products straight off the packed halves
    I = ((mixmul_lo(w0.x, t00.x) + mixmul_lo(w0.y, t10.x)) + mixmul_lo(w1.x, t01.x)) + mixmul_lo(w1.y, t11.x);
    const cs_f2 g00 = {mixmul_hi(w0.x, t00.x), mixmul_lo(w0.x, t00.y)}, g10 = {mixmul_hi(w0.y, t10.x), mixmul_lo(w0.y, t10.y)};
    const cs_f2 g01 = {mixmul_hi(w1.x, t01.x), mixmul_lo(w1.x, t01.y)}, g11 = {mixmul_hi(w1.y, t11.x), mixmul_lo(w1.y, t11.y)};
    G = ((g00 + g10) + g01) + g11;
}

// Correctly rounded square root of a gradient magnitude squared (a sum of two squares of blends of binary16 texels: 0 or
// >= 2^-48, finite).  sqrtf() compiles to v_sqrt_f32 (<= 1 ulp) followed by the two-sided correction below AND a 2^32 pre-scale /
// 2^-16 post-scale for inputs under 2^-96 and a class test for 0 / inf -- 16 instructions, a fifth of the window pixel's work
// (profiles/r03_*): the input's range makes the scale and the class test dead weight.  What is kept is the compiler's own
// correction, instruction for instruction: y -= 1 ulp if x - (y - 1 ulp) y <= 0, y += 1 ulp if x - (y + 1 ulp) y > 0 (both
// residuals in one FMA each).  x = 0: y = 0, both tests fail (NaN / -0), the result is 0.  Same bits as sqrtf on the range.
__device__ __forceinline__ float rows_sqrt(float x) {
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
    float r = (rm <= 0.0f) ? ym : y;
    r = (rp > 0.0f) ? yp : r;
    return r;
}

// one Gauss-Newton pass worth of window sums for this lane's row (all zero for an inactive lane)
struct RowSums {
    cs_f2 ab, ce, r01;  // (a, b), (c, e'), (r0, r1) of klt_tracker_with_gain.cg:106-110
    float d, r2s, ssd;
};

// klt_tracker_with_gain.cg:99-121 for one window pixel: frame-0 sample (I0, G0, m0), frame-1 sample (I1, G1)
__device__ __forceinline__ void rows_accumulate(RowSums& s, float beta, float I0, cs_f2 G0, float m0, float I1, cs_f2 G1, cs_f2 wh,
                                                float lambda) {
    const float ex = beta * I0 - I1;             // :99
    // :100, (x * wh) / 2 as x * (wh / 2): halving is exact, so the two roundings coincide bit for bit (no subnormals here)
    const cs_f2 g = (beta * G0 + G1) * wh;
    const cs_f2 q = G1 * G1;
    const float m1 = rows_sqrt(q.x + q.y);       // :103
    s.ab += g.x * g;                             // :106  a += gx gx, b += gx gy
    s.ce += g * (-I0);                           //       c += gx (-I0) ... :107 e' += gy (-I0)
    s.d += g.y * g.y;                            // :107
    s.r01 += ex * g;                             // :110
    s.r2s += -ex * I0 + lambda * m0 * (m1 - beta * m0);  // :111 without the neighbour term
    s.ssd += ex * ex;                            // :121
}

// group-cooperative patch fill, split so that every load is issued before the first LDS store
template <int LPF, int R, int NT>
__device__ __forceinline__ void patch_load(const cs_texel* __restrict__ L, int Wl, int Hl, int rx0, int ry0, int r,
                                           cs_texel (&tv)[NT]) {
    // (opaque copy of r: otherwise the (lx, ly) of all NT cells are hoisted out of the level loop and live -- spilled --
    // across the whole kernel)
    asm volatile("" : "+v"(r));
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int idx = r + LPF * u;
        const int ly = idx / R, lx = idx - ly * R;
        const int gx = cs_clampi(rx0 + lx, 0, Wl - 1), gy = cs_clampi(ry0 + ly, 0, Hl - 1);
        tv[u] = L[(size_t)gy * Wl + gx];  // (clamped: in range even when idx >= R * R)
    }
}
template <int LPF, int R, int NT>
__device__ __forceinline__ void patch_store(cs_texel* patch, int r, const cs_texel (&tv)[NT]) {
    asm volatile("" : "+v"(r));
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int idx = r + LPF * u;
        if (idx < R * R) patch[idx] = tv[u];
    }
}

// the 3 x 3 Gauss-Newton solve, split where the neighbours' gains enter (klt_tracker_with_gain.cg:12-40,125-134)
struct RowsSolve {
    float det, rcp, pX, pY, pZ, C_, E_, F_;
};
__device__ __forceinline__ RowsSolve solve_prepare(float a, float b, float c, float d, float e_, float f, float r0, float r1) {
    RowsSolve S;
    float det = a * d * f + 2.0f * b * c * e_;
    det -= (a * e_ * e_ + b * b * f) + c * c * d;
    S.det = det;
    S.rcp = 1.0f / det;
    const float A_ = d * f - e_ * e_, B_ = c * e_ - b * f, D_ = a * f - c * c;
    S.C_ = b * e_ - c * d;
    S.E_ = b * c - a * e_;
    S.F_ = a * d - b * b;
    S.pX = A_ * r0 + B_ * r1;
    S.pY = B_ * r0 + D_ * r1;
    S.pZ = S.C_ * r0 + S.E_ * r1;
    return S;
}
__device__ __forceinline__ void solve_finish(const RowsSolve& S, float r2s, float nPixF, float delta, float bsum, float& dX,
                                             float& dY, float& dZ) {
    const float r2 = r2s + nPixF * (delta * bsum);
    dX = (S.pX + S.C_ * r2) * S.rcp;
    dY = (S.pY + S.E_ * r2) * S.rcp;
    dZ = (S.pZ + S.F_ * r2) * S.rcp;
}

// nb: lane r of the group holds neighbour r's gain (r 0..3 betaN1, 4..7 betaN2; negative = dead -> own gain).
// dot(1, N1 + N2 - 2 beta), klt_tracker_with_gain.cg:74-75,111; the result is valid in every lane of the group.
template <int LPF>
__device__ __forceinline__ float rows_bsum(float nb, float beta, int r) {
    const float v = (nb < 0) ? beta : nb;
    const float pair = v + cs_dpp_f<0x104, 0xf>(v);  // row_shl:4 -- lane r reads lane r + 4: N1[r] + N2[r] for r < 4
    const float t = pair - 2.0f * beta;
    float bs = ((cs_dpp_f<0x00, 0xf>(t) + cs_dpp_f<0x55, 0xf>(t)) + cs_dpp_f<0xAA, 0xf>(t)) + cs_dpp_f<0xFF, 0xf>(t);
    const float m = cs_dpp_f<0x141, 0xf>(bs);  // lanes 4..7 <- lanes 3..0
    bs = (r < 4) ? bs : m;
    if (LPF == 16) {
        const float m2 = cs_dpp_f<0x140, 0xf>(bs);  // lanes 8..15 <- lanes 7..0
        bs = (r < 8) ? bs : m2;
    }
    return bs;
}

// slot of neighbour q (0..3 betaN1, 4..7 betaN2) of slot (si, sj): NEAREST + CLAMP_TO_EDGE on the feature texture
__device__ __forceinline__ int rows_nb_slot(const CsRowsArgs& A, int si, int sj, int q) {
    int dx, dy;
    if (q < 4) {
        dx = A.n1x[q];
        dy = A.n1y[q];
    } else {
        const int p = q - 4;
        dx = (p == 0) ? 1 : (p == 1 ? -1 : 0);
        dy = (p == 2) ? 1 : (p == 3 ? -1 : 0);
    }
    return cs_clampi(sj + dy, 0, A.fh - 1) * A.fw + cs_clampi(si + dx, 0, A.fw - 1);
}

// frame-1 patches + region B (frame-0 patches, later the per-lane frame-0 records)
__host__ __device__ constexpr size_t rows_lds_bytes(int fpw, int R, int R0, int FW) {
    const size_t b0 = (size_t)fpw * R0 * R0 * sizeof(cs_texel), b1 = (size_t)64 * FW * 16;
    return (size_t)fpw * R * R * sizeof(cs_texel) + (b0 > b1 ? b0 : b1);
}


// ---- ALL passes of ALL cameras in one persistent launch ----------------------------------------------------------
template <int LPF, int FW, bool PROBE>
__global__ __launch_bounds__(64 * CS_ROWS_WPB, (LPF == 8 ? CS_ROWS_MINBLOCKS : 2)) void k_track_rows_fused(CsRowsArgs A) {
    constexpr int HW = FW / 2, R = FW + 1 + 2 * RW_MARGIN, FPW = 64 / LPF, NT = (R * R + LPF - 1) / LPF;
    constexpr int R0 = FW + 2, NT0 = (R0 * R0 + LPF - 1) / LPF;  // frame-0 footprint: fixed position, no slack needed
    constexpr int NPIX = FW * FW;
    unsigned long long tTex = 0, tMath = 0, tPoll = 0, tPost = 0, nPoll = 0, nReload = 0, tStart = 0, tm0 = 0, tm1 = 0;
    if (PROBE) tStart = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_setprio(CS_ROWS_PRIO_BASE);
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    // Placement.  Workgroup b of a launch runs on XCD b % 8 (observed; used for speed only -- nothing below depends on it):
    // with xcdsPerCam = q the 1-D grid hands XCDs c q .. c q + q - 1 to camera c, so a camera's two pyramids are fetched into
    // ONE L2 instead of eight and its granule rows are swept inside that L2's XCD.
    int camIdx = blockIdx.y, bx = blockIdx.x;
    if (A.xcdsPerCam > 0) {
        const int xcd = blockIdx.x & 7, round = blockIdx.x >> 3;
        camIdx = xcd / A.xcdsPerCam;
        bx = round * A.xcdsPerCam + (xcd - camIdx * A.xcdsPerCam);
        if (bx * (CS_ROWS_WPB * FPW) >= A.N) return;  // (padding of the 1-D grid: no slot, nobody waits for it)
    }
    const int waveId = bx * CS_ROWS_WPB + wib;
    const int g = lane / LPF, r = lane - g * LPF;
    const int k = waveId * FPW + g;
    const bool valid = k < A.N;
    const CsRowsCam& Cm = A.cam[camIdx];
    const cs_texel* __restrict__ pyr0 = Cm.pyr0;
    const cs_texel* __restrict__ pyr1 = Cm.pyr1;
    cs_granule* gran = Cm.gran;
    const int N = A.N;

    float X0x = -1.0f, X0y = -1.0f, X1x = -1.0f, X1y = -1.0f;
    if (valid) {
        X0x = Cm.feat0[3 * k];
        X0y = Cm.feat0[3 * k + 1];
        X1x = Cm.featStart[3 * k];
        X1y = Cm.featStart[3 * k + 1];
    }
    float beta = 1.0f;  // v3d_gpuklt.cpp:223-227
    bool dead = !valid || (X1x < 0) || (X0x < 0);
    float pX = X1x, pY = X1y, pB = 1.0f;

    const unsigned tagBase = *Cm.tagWord;  // frame-unique: what the rows hold from the previous frame can never match
    if (valid && r == 0) gran_store(gran + k, tagBase + 1u, 1.0f);  // row 0: beta_0 = 1 for every slot, dead or alive

    // lane r < 8 of a group sweeps neighbour r's granule; every other lane (and a neighbour that clamps onto the slot
    // itself) re-reads the group's own granule
    int nbSlot = valid ? k : 0;
    if (valid && r < 8) nbSlot = rows_nb_slot(A, k % A.fw, k / A.fw, r);
    const bool polls = valid && (nbSlot != k);

    const float whx = (float)A.W, why = (float)A.H;
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_rows_smem[];
    // LDS of the wave: [frame-1 patches: FPW x R x R texels][region B], region B = the frame-0 patches (FPW x R0 x R0
    // texels) while a level's frame-0 footprint is sampled, then every lane's FW frame-0 records {I0, I0x, I0y, |grad I0|}
    // ([px][lane] float4: conflict-free 16-byte reads) for the level's passes -- the row loop then carries no
    // per-pixel register state and stays rolled.
    unsigned char* waveSmem = cs_rows_smem + (size_t)wib * (rows_lds_bytes(FPW, R, R0, FW) + CS_ROWS_LDS_PAD);
    cs_texel* patch = (cs_texel*)waveSmem + (size_t)g * (R * R);
    cs_texel* patch0 = (cs_texel*)waveSmem + (size_t)FPW * (R * R) + (size_t)g * (R0 * R0);
    float4* rec = (float4*)((cs_texel*)waveSmem + (size_t)FPW * (R * R)) + lane;
    const bool rowOn = r < FW;
    unsigned pass = 0;
    for (int level = A.lv.L - 1; level >= 0; level -= A.levelSkip) {
        const cs_texel* L0 = pyr0 + A.lv.off[level];
        const cs_texel* L1 = pyr1 + A.lv.off[level];
        const int Wl = A.lv.w[level], Hl = A.lv.h[level];
        const float dsx = 1.0f / (float)Wl, dsy = 1.0f / (float)Hl;  // v3d_gpuklt.cpp:255-260
        const float oxLo = (float)(-HW) * dsx, oxHi = (float)HW * dsx, oyLo = (float)(-HW) * dsy, oyHi = (float)HW * dsy;
        const float oy = (float)(r - HW) * dsy;
        // Both footprints of the level are requested in ONE batch of loads (the iterate does not move between levels):
        // frame 0 around X0 -- sampled once per level, kept in registers -- and the first frame-1 patch around X1.
        int qx0 = 0, qy0 = 0, rx0 = 0, ry0 = 0;
        if (!dead) {
            cs_texel tv0[NT0], tv1[NT];
            qx0 = fp_floor(X0x + oxLo, Wl);
            qy0 = fp_floor(X0y + oyLo, Hl);
            rx0 = fp_floor(X1x + oxLo, Wl) - RW_MARGIN;
            ry0 = fp_floor(X1y + oyLo, Hl) - RW_MARGIN;
            patch_load<LPF, R0, NT0>(L0, Wl, Hl, qx0, qy0, r, tv0);
            patch_load<LPF, R, NT>(L1, Wl, Hl, rx0, ry0, r, tv1);
            patch_store<LPF, R0, NT0>(patch0, r, tv0);
            patch_store<LPF, R, NT>(patch, r, tv1);
            if (PROBE) ++nReload;
        }
        bool patchValid = !dead;
        wave_lds_sync();
        float fRow = 0.0f;
        {
            float I0[FW], I0x[FW], I0y[FW], m0[FW];
#pragma unroll
            for (int px = 0; px < FW; ++px) I0[px] = I0x[px] = I0y[px] = m0[px] = 0.0f;
            if (!dead && rowOn) {
                const float t0 = X0y + oy;
#pragma unroll
                for (int px = 0; px < FW; ++px) {
                    sample_p<R0>(patch0, qx0, qy0, Wl, Hl, X0x + (float)(px - HW) * dsx, t0, I0[px], I0x[px], I0y[px]);
                    m0[px] = sqrtf(I0x[px] * I0x[px] + I0y[px] * I0y[px]);  // klt_tracker_with_gain.cg:102
                    fRow += (I0[px] * I0[px] + A.lambda * m0[px] * m0[px]) + A.delta * 8.0f;  // :108
                }
            }
            wave_lds_sync();  // every lane is done with the frame-0 patches: region B becomes the record array
#pragma unroll
            for (int px = 0; px < FW; ++px) rec[px * 64] = make_float4(I0[px], I0x[px], I0y[px], m0[px]);
        }
        const float fLevel = rows_fold<LPF>(fRow);  // does not change within a level

        for (int iter = 1; iter <= A.nIter; ++iter) {
            ++pass;
            const cs_granule* src = gran + (size_t)(pass - 1) * N + nbSlot;
            const unsigned want = tagBase + pass;
            if (PROBE) tm0 = __builtin_amdgcn_s_memtime();
            // re-centre the patch when the window's footprint has left it (rare: RW_MARGIN texels of slack per level)
            bool need = false;
            int iLo = 0, jLo = 0;
            if (!dead) {
                iLo = fp_floor(X1x + oxLo, Wl);
                jLo = fp_floor(X1y + oyLo, Hl);
                const int iHi = fp_floor(X1x + oxHi, Wl) + 1, jHi = fp_floor(X1y + oyHi, Hl) + 1;
                need = !patchValid || iLo < rx0 || jLo < ry0 || iHi >= rx0 + R || jHi >= ry0 + R;
            }
            if (__any(need)) {
                cs_texel tv[NT];
                if (need) {
                    rx0 = iLo - RW_MARGIN;
                    ry0 = jLo - RW_MARGIN;
                    patch_load<LPF, R, NT>(L1, Wl, Hl, rx0, ry0, r, tv);
                }
                wave_lds_sync();
                if (need) {
                    patch_store<LPF, R, NT>(patch, r, tv);
                    patchValid = true;
                    if (PROBE) ++nReload;
                }
                wave_lds_sync();
            }
            // ---- this lane's window row: klt_tracker_with_gain.cg:86-122 ------------------------------------
            RowSums s = {{0, 0}, {0, 0}, {0, 0}, 0, 0, 0};
            if (!dead && rowOn) {
                const float t1 = X1y + oy;
                const cs_f2 wh = {whx * 0.5f, why * 0.5f};
#pragma unroll CS_ROWS_UNROLL
                for (int px = 0; px < FW; ++px) {
                    const float4 q0 = rec[px * 64];
                    float I1;
                    cs_f2 G1;
                    sample_p2<R>(patch, rx0, ry0, Wl, Hl, X1x + (float)(px - HW) * dsx, t1, I1, G1);
                    rows_accumulate(s, beta, q0.x, (cs_f2){q0.y, q0.z}, q0.w, I1, G1, wh, A.lambda);
                }
            }
            if (PROBE) {
                asm volatile("" : "+v"(s.d));
                tm1 = __builtin_amdgcn_s_memtime();
                tTex += tm1 - tm0;
                tm0 = tm1;
            }
            // The first poll goes out here and flies under the folds and the adjugate.  From here to the publish the
            // wave runs at raised priority: the closer a wave is to publishing the granule its neighbours wait for, the
            // earlier it gets the SIMD's issue slots over a co-resident wave that is still sampling.
            __builtin_amdgcn_s_setprio(CS_ROWS_PRIO_HOT);
            cs_granule got = gran_load(src);
            if (PROBE) ++nPoll;
            const float a = rows_fold<LPF>(s.ab.x), b = rows_fold<LPF>(s.ab.y), c = rows_fold<LPF>(s.ce.x), d = rows_fold<LPF>(s.d);
            const float e_ = rows_fold<LPF>(s.ce.y), r0 = rows_fold<LPF>(s.r01.x), r1 = rows_fold<LPF>(s.r01.y);
            float r2s = rows_fold<LPF>(s.r2s);
            const float SSD = rows_fold<LPF>(s.ssd);
            const RowsSolve S = solve_prepare(a, b, c, d, e_, fLevel, r0, r1);
            // thresholds: v3d_gpuklt.cpp:271-279
            const bool real = (iter == A.nIter) && (iter != 1);
            const float sqrConvThr = real ? A.sqrConvThr : 1000000.0f;
            const float ssdThr = real ? A.ssdThr : 1000000.0f;
            const float vr0 = real ? A.vr[0] : -1.0f, vr1 = real ? A.vr[1] : -1.0f;
            const float vr2 = real ? A.vr[2] : 2.0f, vr3 = real ? A.vr[3] : 2.0f;
            float invalidEarly = ((S.det < 0.00001f) || (SSD > ssdThr)) ? 1.0f : 0.0f;
            RowsSolve Sp = S;
            // pin the prepared solve here: otherwise the compiler sinks the adjugate and the IEEE division below the
            // sweep, back onto the hand-off's critical path
            asm volatile("; solve prepared" : "+v"(Sp.rcp), "+v"(Sp.pX), "+v"(Sp.pY), "+v"(Sp.pZ), "+v"(Sp.C_), "+v"(Sp.E_),
                         "+v"(Sp.F_), "+v"(invalidEarly), "+v"(r2s));
            if (PROBE) {
                tm1 = __builtin_amdgcn_s_memtime();
                tMath += tm1 - tm0;
                tm0 = tm1;
            }
            // ---- sweep the neighbours' granules of the previous pass ---------------------------------------
            float nbBeta = beta;
            {
                unsigned spins = 0;
#pragma nounroll
                while (!__all(!polls || ((unsigned)(got >> 32) == want))) {
                    got = gran_load(src);
                    if (PROBE) ++nPoll;
                    if (++spins > (1u << 20)) {
                        if (lane == 0) atomicExch(Cm.err, 1);
                        break;
                    }
                }
                if (polls) nbBeta = __uint_as_float((unsigned)got);
                if (PROBE) {
                    asm volatile("" : "+v"(nbBeta));
                    tm1 = __builtin_amdgcn_s_memtime();
                    tPoll += tm1 - tm0;
                    tm0 = tm1;
                }
            }
            const float bsum = rows_bsum<LPF>(nbBeta, beta, r);
            float newX = -1.0f, newY = -1.0f, newB = -1.0f;
            if (!dead) {
                float dX, dY, dZ;
                solve_finish(Sp, r2s, (float)NPIX, A.delta, bsum, dX, dY, dZ);
                const float nX = X1x + dX, nY = X1y + dY;  // :137
                const float ux = dX * whx, uy = dY * why;  // :139-140
                const float sqrLen = ux * ux + uy * uy;
                bool invalid = (invalidEarly != 0.0f);                                       // :142-143
                invalid = invalid || (sqrLen > sqrConvThr);                                  // :144
                invalid = invalid || (nX < vr0 || nY < vr1) || (nX > vr2 || nY > vr3);       // :145
                const float nB = beta + dZ;
                if (!(invalid || !(nX == nX) || !(nY == nY) || !(nB == nB))) {
                    newX = nX;
                    newY = nY;
                    newB = nB;
                }
            }
            pX = dead ? -1.0f : X1x;
            pY = dead ? -1.0f : X1y;
            pB = dead ? -1.0f : beta;
            if (pass == 1) {  // the buffer the first pass read from holds (x, y, 1) for every slot
                pX = X1x;
                pY = X1y;
                pB = 1.0f;
            }
            X1x = newX;
            X1y = newY;
            beta = newB;
            dead = dead || (newX < 0);
            if (valid && r == 0) gran_store(gran + (size_t)pass * N + k, want + 1u, beta);
            __builtin_amdgcn_s_setprio(CS_ROWS_PRIO_BASE);
            if (PROBE) {
                tm1 = __builtin_amdgcn_s_memtime();
                tPost += tm1 - tm0;
            }
        }
    }
    if (valid && r == 0) {
        Cm.outLast[3 * k] = X1x;
        Cm.outLast[3 * k + 1] = X1y;
        Cm.outLast[3 * k + 2] = beta;
        Cm.outPrev[3 * k] = pX;
        Cm.outPrev[3 * k + 1] = pY;
        Cm.outPrev[3 * k + 2] = pB;
        if (Cm.dest) {  // v3d_gpuklt.cpp:872-888 status loop + :744-752 present scatter
            cs_klt_feature* dst = Cm.dest + k;
            if (X1x >= 0) {
                dst->status = 0;
                dst->pos[0] = X1x;
                dst->pos[1] = X1y;
                dst->gain = beta;
                dst->fed = -1;
                if (A.doSuppress && X1y >= 0.0f) {  // v3d_gpuklt.cpp:444-447
                    const float fx = floorf(X1x * (float)A.W), fy = floorf(X1y * (float)A.H);
                    if (fx < (float)A.W && fy < (float)A.H) Cm.corner[(size_t)(int)fy * A.W + (int)fx] = -1e30f;
                }
            } else {
                dst->status = -1;
                dst->fed = -1;
            }
        }
    }
    if (PROBE && Cm.probe && lane == 0) {
        unsigned long long* o = Cm.probe + 8 * (size_t)waveId;
        o[0] = tTex;
        o[1] = tMath;
        o[2] = tPoll;
        o[3] = tPost;
        o[4] = nPoll;
        o[5] = __builtin_amdgcn_s_memtime() - tStart;
        o[6] = tStart;
        o[7] = nReload;
    }
}

// ---- one launch of klt_tracker_with_gain.cg:42-148 on level A.level, every camera of the group --------------------
// Same lane layout and the same arithmetic as the persistent kernel (bit-identical results); the neighbours' gains come
// from the previous launch's output buffer, the footprints straight from the pyramid.
template <int LPF, int FW>
__global__ __launch_bounds__(64) void k_track_rows_pass(CsRowsArgs A) {
    constexpr int HW = FW / 2, FPW = 64 / LPF, NPIX = FW * FW;
    const int lane = threadIdx.x;
    const int g = lane / LPF, r = lane - g * LPF;
    const int k = blockIdx.x * FPW + g;
    const bool valid = k < A.N;
    const CsRowsCam& Cm = A.cam[blockIdx.y];
    const float* __restrict__ featIn = Cm.featStart;
    float X0x = -1.0f, X0y = -1.0f, X1x = -1.0f, X1y = -1.0f, beta = -1.0f;
    if (valid) {
        X0x = Cm.feat0[3 * k];
        X0y = Cm.feat0[3 * k + 1];
        X1x = featIn[3 * k];
        X1y = featIn[3 * k + 1];
        beta = featIn[3 * k + 2];
    }
    const bool dead = !valid || (X1x < 0) || (X0x < 0);  // :77 -> :147
    float nb = beta;
    if (valid && r < 8) nb = featIn[3 * (size_t)rows_nb_slot(A, k % A.fw, k / A.fw, r) + 2];
    const float bsum = rows_bsum<LPF>(nb, beta, r);

    const int level = A.level;
    const cs_texel* L0 = Cm.pyr0 + A.lv.off[level];
    const cs_texel* L1 = Cm.pyr1 + A.lv.off[level];
    const int Wl = A.lv.w[level], Hl = A.lv.h[level];
    const float dsx = 1.0f / (float)Wl, dsy = 1.0f / (float)Hl;
    const float whx = (float)A.W, why = (float)A.H;
    const float oy = (float)(r - HW) * dsy;
    RowSums s = {{0, 0}, {0, 0}, {0, 0}, 0, 0, 0};
    float fRow = 0.0f;
    if (!dead && r < FW) {
        const float t0 = X0y + oy, t1 = X1y + oy;
        const cs_f2 wh = {whx * 0.5f, why * 0.5f};
#pragma unroll
        for (int px = 0; px < FW; ++px) {
            const float ox = (float)(px - HW) * dsx;
            float I0, I0x, I0y, I1, I1x, I1y;
            sample_g(L0, Wl, Hl, X0x + ox, t0, I0, I0x, I0y);
            sample_g(L1, Wl, Hl, X1x + ox, t1, I1, I1x, I1y);
            const float m0 = sqrtf(I0x * I0x + I0y * I0y);
            fRow += (I0 * I0 + A.lambda * m0 * m0) + A.delta * 8.0f;
            rows_accumulate(s, beta, I0, (cs_f2){I0x, I0y}, m0, I1, (cs_f2){I1x, I1y}, wh, A.lambda);
        }
    }
    const float f = rows_fold<LPF>(fRow);
    const float a = rows_fold<LPF>(s.ab.x), b = rows_fold<LPF>(s.ab.y), c = rows_fold<LPF>(s.ce.x), d = rows_fold<LPF>(s.d);
    const float e_ = rows_fold<LPF>(s.ce.y), r0 = rows_fold<LPF>(s.r01.x), r1 = rows_fold<LPF>(s.r01.y);
    const float r2s = rows_fold<LPF>(s.r2s);
    const float SSD = rows_fold<LPF>(s.ssd);
    const RowsSolve S = solve_prepare(a, b, c, d, e_, f, r0, r1);
    float dX, dY, dZ;
    solve_finish(S, r2s, (float)NPIX, A.delta, bsum, dX, dY, dZ);
    const float nX = X1x + dX, nY = X1y + dY;
    const float ux = dX * whx, uy = dY * why;
    const float sqrLen = ux * ux + uy * uy;
    bool invalid = dead || (S.det < 0.00001f);
    invalid = invalid || (SSD > A.ssdThr);
    invalid = invalid || (sqrLen > A.sqrConvThr);
    invalid = invalid || (nX < A.vr[0] || nY < A.vr[1]) || (nX > A.vr[2] || nY > A.vr[3]);
    const float nB = beta + dZ;
    if (valid && r == 0) {
        float* o = Cm.outLast + 3 * (size_t)k;
        if (invalid || !(nX == nX) || !(nY == nY) || !(nB == nB)) {
            o[0] = o[1] = o[2] = -1.0f;
        } else {
            o[0] = nX;
            o[1] = nY;
            o[2] = nB;
        }
    }
}

template <int LPF, int FW>
int launch_fused(const CsRowsArgs& a, hipStream_t stream) {
    constexpr int R = FW + 1 + 2 * RW_MARGIN, R0 = FW + 2, FPW = 64 / LPF;
    const size_t lds = (rows_lds_bytes(FPW, R, R0, FW) + CS_ROWS_LDS_PAD) * CS_ROWS_WPB;
    const int waves = (a.N + FPW - 1) / FPW;
    const int wgPerCam = (waves + CS_ROWS_WPB - 1) / CS_ROWS_WPB;
    dim3 grid(wgPerCam, a.nCams), block(64 * CS_ROWS_WPB);
    if (a.xcdsPerCam > 0) grid = dim3(8 * ((wgPerCam + a.xcdsPerCam - 1) / a.xcdsPerCam), 1);
    bool probe = false;
    for (int c = 0; c < a.nCams; ++c) probe = probe || (a.cam[c].probe != nullptr);
    if (probe)
        hipLaunchKernelGGL((k_track_rows_fused<LPF, FW, true>), grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL((k_track_rows_fused<LPF, FW, false>), grid, block, lds, stream, a);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
template <int LPF, int FW>
int launch_pass(const CsRowsArgs& a, hipStream_t stream) {
    constexpr int FPW = 64 / LPF;
    dim3 grid((a.N + FPW - 1) / FPW, a.nCams), block(64);
    hipLaunchKernelGGL((k_track_rows_pass<LPF, FW>), grid, block, 0, stream, a);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
template <int LPF, int FW>
int occupancy(int* blocksPerCu) {
    constexpr int R = FW + 1 + 2 * RW_MARGIN, R0 = FW + 2, FPW = 64 / LPF;
    const size_t lds = (rows_lds_bytes(FPW, R, R0, FW) + CS_ROWS_LDS_PAD) * CS_ROWS_WPB;
    int n = 0;
    CS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_track_rows_fused<LPF, FW, false>,
                                                        64 * CS_ROWS_WPB, lds));
    *blocksPerCu = n * CS_ROWS_WPB;  // in waves
    return CS_OK;
}

}  // namespace

#define CS_ROWS_DISPATCH(hw, CALL)      \
    switch (hw) {                       \
        case 1: return CALL(8, 3);      \
        case 2: return CALL(8, 5);      \
        case 3: return CALL(8, 7);      \
        case 4: return CALL(16, 9);     \
        case 5: return CALL(16, 11);    \
        case 6: return CALL(16, 13);    \
        case 7: return CALL(16, 15);    \
        default: break;                 \
    }

bool cs_rows_supported(int hw) { return hw >= 1 && hw <= 7; }

size_t cs_rows_lds_bytes(int hw) {
    const int R = 2 * hw + 2 + 2 * RW_MARGIN, R0 = 2 * hw + 3, fpw = (hw <= 3) ? 8 : 4;
    return (rows_lds_bytes(fpw, R, R0, 2 * hw + 1) + CS_ROWS_LDS_PAD) * CS_ROWS_WPB;
}

int cs_rows_waves(int hw, int N) {
    const int fpw = (hw <= 3) ? 8 : 4;
    return (N + fpw - 1) / fpw;
}

// Resident WAVES of the persistent kernel the whole device admits (occupancy query for the instantiation
// actually launched, with its dynamic LDS size); 0 on error.
int cs_rows_max_resident_blocks(int hw, int device, int* blocksPerCu) {
    int per = 0;
    int rc = CS_ERR_INVALID;
#define CS_ROWS_OCC(LPF, FW) (rc = occupancy<LPF, FW>(&per), 0)
    switch (hw) {
        case 1: CS_ROWS_OCC(8, 3); break;
        case 2: CS_ROWS_OCC(8, 5); break;
        case 3: CS_ROWS_OCC(8, 7); break;
        case 4: CS_ROWS_OCC(16, 9); break;
        case 5: CS_ROWS_OCC(16, 11); break;
        case 6: CS_ROWS_OCC(16, 13); break;
        case 7: CS_ROWS_OCC(16, 15); break;
        default: break;
    }
#undef CS_ROWS_OCC
    if (rc != CS_OK) return 0;
    if (per > 32) per = 32;  // 32 waves per CU
    {
        // What the chip really holds is less than the query says when the workgroups' LDS adds up to almost all of a CU's 160 KB: the
        // 7 x 7 instantiation (53 KB per workgroup) is reported as three per CU, and an XCD takes 64 of them, not 65 -- two per CU
        // (tools/r05_diag.py, profiles/r05_ab_runs.txt).  Budget with 4 KB of a CU's LDS set aside.
        const size_t lds = cs_rows_lds_bytes(hw);
        const int byLds = lds ? (int)((160 * 1024 - 4096) / lds) * CS_ROWS_WPB : per;
        if (per > byLds) per = byLds;
    }
    if (blocksPerCu) *blocksPerCu = per;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    return per * prop.multiProcessorCount;
}

int cs_launch_track_rows_fused(const CsRowsArgs& a, int hw, hipStream_t stream) {
#define CS_ROWS_F(LPF, FW) launch_fused<LPF, FW>(a, stream)
    CS_ROWS_DISPATCH(hw, CS_ROWS_F)
#undef CS_ROWS_F
    cs_set_error("rows tracker: windowWidth %d not supported", 2 * hw + 1);
    return CS_ERR_INVALID;
}

int cs_launch_track_rows_pass(const CsRowsArgs& a, int hw, hipStream_t stream) {
#define CS_ROWS_P(LPF, FW) launch_pass<LPF, FW>(a, stream)
    CS_ROWS_DISPATCH(hw, CS_ROWS_P)
#undef CS_ROWS_P
    cs_set_error("rows tracker: windowWidth %d not supported", 2 * hw + 1);
    return CS_ERR_INVALID;
}
