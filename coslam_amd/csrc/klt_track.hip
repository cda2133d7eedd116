// klt_track.hip -- per-feature Lucas-Kanade window iteration, gfx950.
//
// Replaces KLT_Tracker::trackFeatures (src/tracking/CGKLT/v3d_gpuklt.cpp:99-161, Shaders/klt_tracker.cg)
// and KLT_TrackerWithGain::trackFeaturesAndGain (v3d_gpuklt.cpp:205-305, Shaders/klt_tracker_with_gain.cg).
//
// Design (feature-major): the reference runs one fragment per feature and walks the (2hw+1)^2 window
// serially with 2 x 4-tap texture fetches per pixel.  Here ONE WAVE owns one feature: lane p owns
// window pixel p (49 of 64 lanes for the 7x7 default; wider windows stride by 64), fetches its own
// bilinear footprints (8-byte texels, L2-resident pyramid), and the 5 (no gain) / 10 (gain) normal-
// equation sums are folded across the wave with 64-lane butterflies.  Every lane then holds the
// same sums and redundantly solves the 2x2 / 3x3 system, so no LDS round trip and no divergence.
//   - no gain: all levels and all iterations run inside one launch; the frame-0 samples (which do not
//     depend on the iterate) are fetched once per level and kept in registers.
//   - with gain: every Gauss-Newton step reads the neighbours' gains of the previous step
//     (klt_tracker_with_gain.cg:64-75), which the reference honours with one launch per step.  The default here is
//     ONE persistent launch (k_track_gain_fused): every feature's wave stays resident, neighbours hand their gain
//     over through 8-byte tagged granules, the frame-1 footprint lives in a wave-private LDS patch; the
//     launch-per-step schedule (k_track_gain_pass) is kept as the fallback when the grid cannot be co-resident and
//     as the reference the persistent kernel must match bit for bit.
#include "klt_internal.h"

#pragma clang fp contract(off)



namespace {

// GL_LINEAR + CLAMP_TO_EDGE fetch of one texel footprint; same arithmetic as oracle okl_sample()
__device__ __forceinline__ void sample(const cs_texel* __restrict__ lvl, int Wl, int Hl, float s, float t, float& I,
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

// The same fetch out of a wave-private LDS patch: cell (lx, ly) of `patch` (row pitch R) holds the texel at
// (clamp(rx0 + lx), clamp(ry0 + ly)) of the level, so unclamped footprint indices minus the patch origin address it
// and CLAMP_TO_EDGE is already folded in.  Arithmetic identical to sample().
__device__ __forceinline__ void sample_patch(const cs_texel* patch, int R, int rx0, int ry0, int Wl, int Hl, float s,
                                             float t, float& I, float& Ix, float& Iy) {
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

// floor of the (clamped) texel coordinate sample() would compute for normalised coordinate s on a level of width Wl
__device__ __forceinline__ int footprint_floor(float s, int Wl) {
    float u = s * (float)Wl - 0.5f;
    u = fminf(fmaxf(u, -2.0f), (float)Wl + 1.0f);
    return (int)floorf(u);
}

// ---- no gain: klt_tracker.cg:24-132 -----------------------------------------------------------
template <int NPL>  // window pixels per lane = ceil((2hw+1)^2 / 64)
__global__ __launch_bounds__(256) void k_track_nogain(const cs_texel* __restrict__ pyr0,
                                                      const cs_texel* __restrict__ pyr1, CsTrackLevels lv, int W, int H,
                                                      int levelSkip, int hw, int nIter, float sqrConv, float ssdThr,
                                                      float vr0, float vr1, float vr2, float vr3, int N,
                                                      const float* __restrict__ featIn, float* __restrict__ featOut) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= N) return;
    const float X0x = featIn[3 * k], X0y = featIn[3 * k + 1];
    if (X0x < 0) {  // klt_tracker.cg:47
        if (lane == 0) featOut[3 * k] = featOut[3 * k + 1] = featOut[3 * k + 2] = -1.0f;
        return;
    }
    const int fwid = 2 * hw + 1, nPix = fwid * fwid;
    const float ds = 1.0f / (float)W, dt = 1.0f / (float)H;
    const float whx = (float)W, why = (float)H;
    float X1x = X0x, X1y = X0y;
    bool invalid = false;
    float sqrLen = 0.0f, SSD = 0.0f;
    float mult = (float)(1 << (lv.L - 1));

    for (int level = lv.L - 1; level >= 0; level -= levelSkip) {
        const cs_texel* L0 = pyr0 + lv.off[level];
        const cs_texel* L1 = pyr1 + lv.off[level];
        const int Wl = lv.w[level], Hl = lv.h[level];
        const float dsx = ds * mult, dsy = dt * mult;
        float ox[NPL], oy[NPL], I0[NPL], I0x[NPL], I0y[NPL];
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            int p = lane + 64 * q;
            int py = p / fwid, px = p - py * fwid;
            ox[q] = (float)(px - hw) * dsx;
            oy[q] = (float)(py - hw) * dsy;
            I0[q] = I0x[q] = I0y[q] = 0.0f;
            if (p < nPix) sample(L0, Wl, Hl, X0x + ox[q], X0y + oy[q], I0[q], I0x[q], I0y[q]);
        }
        for (int iter = 0; iter < nIter; ++iter) {
            float a = 0, b = 0, c = 0, rx = 0, ry = 0, ssd = 0;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                if (lane + 64 * q < nPix) {
                    float J, Jx, Jy;
                    sample(L1, Wl, Hl, X1x + ox[q], X1y + oy[q], J, Jx, Jy);
                    float e = I0[q] - J;
                    float gx = (I0x[q] + Jx) * whx / 2.0f;
                    float gy = (I0y[q] + Jy) * why / 2.0f;
                    a += gx * gx;
                    b += gx * gy;
                    c += gy * gy;
                    rx += e * gx;
                    ry += e * gy;
                    ssd += e * e;
                }
            }
            a = cs_wave_sum(a);
            b = cs_wave_sum(b);
            c = cs_wave_sum(c);
            rx = cs_wave_sum(rx);
            ry = cs_wave_sum(ry);
            SSD = cs_wave_sum(ssd);
            float det = a * c - b * b;
            invalid = invalid || (det < 0.00001f);
            float rdet = 1.0f / det;
            float dXx = rdet * (c * rx - b * ry);
            float dXy = rdet * (-b * rx + a * ry);
            X1x += dXx;
            X1y += dXy;
            dXx *= whx;
            dXy *= why;
            sqrLen = dXx * dXx + dXy * dXy;
        }
        invalid = invalid || (sqrLen > sqrConv);
        invalid = invalid || (SSD > ssdThr);
        mult /= (float)(1 << levelSkip);
    }
    invalid = invalid || (X1x < vr0 || X1y < vr1) || (X1x > vr2 || X1y > vr3);
    if (lane == 0) {
        if (invalid || !(X1x == X1x) || !(X1y == X1y)) {
            featOut[3 * k] = featOut[3 * k + 1] = featOut[3 * k + 2] = -1.0f;
        } else {
            featOut[3 * k] = X1x;
            featOut[3 * k + 1] = X1y;
            featOut[3 * k + 2] = X0x;  // klt_tracker.cg:131
        }
    }
}

// ---- with gain: the 3x3 Gauss-Newton solve, split where the neighbours' gains enter ---------------------------
// klt_tracker_with_gain.cg:111-134.  The shader adds delta * bsum to the third right-hand side once per window
// pixel; here the per-pixel part is wave-summed on its own and delta * bsum enters as nPix * (delta * bsum), so
// that (in the persistent kernel) nothing but one multiply-add per Cramer row waits for the hand-off.  Both gain
// kernels go through these three functions: their results are bit-identical to each other.
struct CsGainSolve {
    float det, rcp, pX, pY, pZ, C_, E_, F_;
};

__device__ __forceinline__ CsGainSolve cs_gain_solve_prepare(float a, float b, float c, float d, float e_, float f,
                                                             float r0, float r1) {
    CsGainSolve S;
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

__device__ __forceinline__ void cs_gain_solve_finish(const CsGainSolve& S, float r2s, float nPixF, float delta,
                                                     float bsum, float& dX, float& dY, float& dZ) {
    const float r2 = r2s + nPixF * (delta * bsum);
    dX = (S.pX + S.C_ * r2) * S.rcp;
    dY = (S.pY + S.E_ * r2) * S.rcp;
    dZ = (S.pZ + S.F_ * r2) * S.rcp;
}

// nb: lanes 0..3 hold betaN1, lanes 4..7 betaN2 (negative = dead neighbour -> own gain); dot(1, N1 + N2 - 2 beta).
// The substitution and the pair sums run in the lanes (one row_shl:4 DPP add), four v_readlane and three adds finish.
__device__ __forceinline__ float cs_gain_bsum(float nb, float beta) {
    const float v = (nb < 0) ? beta : nb;
    const float pair = v + cs_dpp_f<0x104, 0xf>(v);  // lane q: N1[q] + N2[q]   (row_shl:4: lane i reads lane i + 4)
    const float t = pair - 2.0f * beta;
    const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 0));
    const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 1));
    const float t2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 2));
    const float t3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 3));
    return ((t0 + t1) + t2) + t3;
}

// ---- with gain: one launch of klt_tracker_with_gain.cg:42-148 --------------------------------
__device__ __forceinline__ float slot_beta(const float* __restrict__ feat, int fw, int fh, int i, int j) {
    i = cs_clampi(i, 0, fw - 1);
    j = cs_clampi(j, 0, fh - 1);
    return feat[3 * ((size_t)j * fw + i) + 2];
}

__global__ __launch_bounds__(256) void k_track_gain_pass(CsGainPassArgs A) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= A.N) return;
    const float X0x = A.feat0[3 * k], X0y = A.feat0[3 * k + 1];
    float X1x = A.featIn[3 * k], X1y = A.featIn[3 * k + 1];
    const float beta = A.featIn[3 * k + 2];
    if ((X1x < 0) || (X0x < 0)) {  // :77 -> :147
        if (lane == 0) A.featOut[3 * k] = A.featOut[3 * k + 1] = A.featOut[3 * k + 2] = -1.0f;
        return;
    }
    const int si = k % A.fw, sj = k / A.fw;
    float bsum;
    {
        const int n2x[4] = {1, -1, 0, 0}, n2y[4] = {0, 0, 1, -1};
        float t4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float b1 = slot_beta(A.featIn, A.fw, A.fh, si + A.n1x[q], sj + A.n1y[q]);
            float b2 = slot_beta(A.featIn, A.fw, A.fh, si + n2x[q], sj + n2y[q]);
            if (b1 < 0) b1 = beta;
            if (b2 < 0) b2 = beta;
            t4[q] = (b1 + b2) - 2.0f * beta;
        }
        bsum = ((t4[0] + t4[1]) + t4[2]) + t4[3];
    }
    const int hw = A.hw, fwid = 2 * hw + 1, nPix = fwid * fwid;
    const float dsx = 1.0f / (float)A.Wl, dsy = 1.0f / (float)A.Hl;
    float a = 0, b = 0, c = 0, d = 0, e_ = 0, f = 0, r0 = 0, r1 = 0, r2s = 0, ssd = 0;
    for (int p = lane; p < nPix; p += 64) {
        int py = p / fwid, px = p - py * fwid;
        float ox = (float)(px - hw) * dsx, oy = (float)(py - hw) * dsy;
        float I0, I0x, I0y, I1, I1x, I1y;
        sample(A.lvl0, A.Wl, A.Hl, X0x + ox, X0y + oy, I0, I0x, I0y);
        sample(A.lvl1, A.Wl, A.Hl, X1x + ox, X1y + oy, I1, I1x, I1y);
        float ex = beta * I0 - I1;
        float gx = (beta * I0x + I1x) * A.whx / 2.0f;
        float gy = (beta * I0y + I1y) * A.why / 2.0f;
        float m0 = sqrtf(I0x * I0x + I0y * I0y);
        float m1 = sqrtf(I1x * I1x + I1y * I1y);
        a += gx * gx;
        b += gx * gy;
        c += gx * (-I0);
        d += gy * gy;
        e_ += gy * (-I0);
        f += (I0 * I0 + A.lambda * m0 * m0) + A.delta * 8.0f;
        r0 += ex * gx;
        r1 += ex * gy;
        r2s += -ex * I0 + A.lambda * m0 * (m1 - beta * m0);
        ssd += ex * ex;
    }
    cs_wave_sum4(a, b, c, d);
    cs_wave_sum4(e_, r0, r1, r2s);
    f = cs_wave_sum(f);
    const float SSD = cs_wave_sum(ssd);

    const CsGainSolve S = cs_gain_solve_prepare(a, b, c, d, e_, f, r0, r1);
    const float det = S.det;
    float dX, dY, dZ;
    cs_gain_solve_finish(S, r2s, (float)nPix, A.delta, bsum, dX, dY, dZ);
    X1x += dX;
    X1y += dY;
    const float ux = dX * A.whx, uy = dY * A.why;
    const float sqrLen = ux * ux + uy * uy;
    bool invalid = (det < 0.00001f);
    invalid = invalid || (SSD > A.ssdThr);
    invalid = invalid || (sqrLen > A.sqrConvThr);
    invalid = invalid || (X1x < A.vr[0] || X1y < A.vr[1]) || (X1x > A.vr[2] || X1y > A.vr[3]);
    const float nb = beta + dZ;
    if (lane == 0) {
        if (invalid || !(X1x == X1x) || !(X1y == X1y) || !(nb == nb)) {
            A.featOut[3 * k] = A.featOut[3 * k + 1] = A.featOut[3 * k + 2] = -1.0f;
        } else {
            A.featOut[3 * k] = X1x;
            A.featOut[3 * k + 1] = X1y;
            A.featOut[3 * k + 2] = nb;
        }
    }
}

// ---- with gain, ALL passes in one persistent launch ------------------------------------------------
// The Jacobi coupling between features is only through the gains of <= 6 neighbouring slots, so instead of a
// kernel boundary per pass (40 boundaries per frame, ~5 us each measured) every feature's wave stays resident for
// the whole schedule and neighbours hand their gain over through memory: after pass p a wave publishes one
// naturally aligned 8-byte granule {tag, beta} with a write-through (sc1) store into row p + 1 of gran[passes + 1][N];
// before pass p it sweeps its neighbours' granules in row p with L1-bypassing loads until every tag equals the one
// pass p - 1 publishes (MI355X guide, Guideline 16 recipe R2: the data is the flag, no fences).  One row per pass and
// tags offset by a per-frame base (a device word the frame's last kernel bumps): a granule is written once per frame
// and what a row still holds from the previous frame can never match, so nothing is zeroed and nothing is ever
// overwritten under a reader.  (Two rows alternating by parity were the first design; they are only safe when the
// neighbour relation is symmetric, and the reference's betaN1 offsets are not for every slot grid -- 100 x 50 has
// (1,1) without (-1,-1) -- so a wave two passes ahead of a reader it does not itself read could overwrite the gain the
// reader was about to fetch: a handful of gains differing in the 6th digit from run to run.)
// Dead features keep sweeping and publishing beta = -1 so the protocol never waits on them.  Every wave of
// the grid must be co-resident (the launcher checks the grid against the device); every spin is bounded and a
// timeout raises *err instead of hanging the GPU.  Frame-0 samples are fetched once per level and kept in
// registers.  Arithmetic per pass is identical to k_track_gain_pass (bit-identical results).
//
// Critical path of a pass (cycle counters, tools/track_sweep.py): a hand-off costs ~1.5 us from publish to the
// neighbour's successful poll, so everything that does not need the neighbours' gains is finished BEFORE the
// sweep -- all ten window sums and their wave folds, the adjugate, 1/det and the two-thirds of each Cramer row
// that multiply r0, r1 -- and only `bsum`, one multiply-add per row, the validity tests and the publish remain
// behind it.
constexpr int CS_PATCH_MARGIN = 2;
typedef unsigned long long cs_granule;
typedef __attribute__((address_space(1))) cs_granule gu64;

__device__ __forceinline__ cs_granule gran_load(const cs_granule* p) {
    return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_store(cs_granule* p, unsigned tag, float beta) {
    __hip_atomic_store((gu64*)p, ((cs_granule)tag << 32) | (cs_granule)__float_as_uint(beta), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

template <int NPL, bool PROBE = false>
__global__ __launch_bounds__(256) void k_track_gain_fused(CsGainFusedArgs A) {
    unsigned long long tTex = 0, tMath = 0, tPoll = 0, tPost = 0, nPoll = 0, nReload = 0, tStart = 0, tm0 = 0, tm1 = 0;
    if (PROBE) tStart = __builtin_amdgcn_s_memtime();
    // the mesh advances at the pace of its slowest wave: win issue arbitration against any foreign wave (pose, BA)
    // that lands on one of these SIMDs
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= A.N) return;
    const float X0x = A.feat0[3 * k], X0y = A.feat0[3 * k + 1];
    float X1x = A.featStart[3 * k], X1y = A.featStart[3 * k + 1];
    float beta = 1.0f;  // v3d_gpuklt.cpp:223-227
    bool dead = (X1x < 0) || (X0x < 0);
    float pX = X1x, pY = X1y, pB = 1.0f;

    const unsigned tagBase = *A.tagWord;  // frame-unique: what the rows hold from the previous frame can never match
    if (lane == 0) gran_store(A.gran + k, tagBase + 1u, 1.0f);  // row 0: beta_0 = 1 for every slot, dead or alive

    // neighbour slots: lanes 0..3 = betaN1, lanes 4..7 = betaN2 (klt_tracker_with_gain.cg:64-72); every other
    // lane (and a neighbour that clamps onto the slot itself) points at the wave's own granule, whose line the
    // row neighbours share, so the sweep is one unconditional load per lane
    const int si = k % A.fw, sj = k / A.fw;
    int nbSlot = k;
    if (lane < 8) {
        const int q = lane & 3;
        int dx, dy;
        if (lane < 4) {
            dx = A.n1x[q];
            dy = A.n1y[q];
        } else {
            dx = (q == 0) ? 1 : (q == 1 ? -1 : 0);
            dy = (q == 2) ? 1 : (q == 3 ? -1 : 0);
        }
        nbSlot = cs_clampi(sj + dy, 0, A.fh - 1) * A.fw + cs_clampi(si + dx, 0, A.fw - 1);
    }
    const bool polls = (nbSlot != k);

    const int hw = A.hw, fwid = 2 * hw + 1, nPix = fwid * fwid;
    const float whx = (float)A.W, why = (float)A.H;
    // wave-private LDS patch of the frame-1 level around the iterate (the window's bilinear footprint plus a margin
    // of CS_PATCH_MARGIN texels on every side): filled once, every pass samples it; it is re-centred only when the
    // iterate drifts out of the margin.  Keeps the per-pass texel traffic out of the CU's memory queue, where the
    // hand-off polls wait.
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_track_smem[];
    const int R = A.patchR;
    cs_texel* patch = (cs_texel*)cs_track_smem + (size_t)(threadIdx.x >> 6) * R * R;
    unsigned pass = 0;
    for (int level = A.lv.L - 1; level >= 0; level -= A.levelSkip) {
        const cs_texel* L0 = A.pyr0 + A.lv.off[level];
        const cs_texel* L1 = A.pyr1 + A.lv.off[level];
        const int Wl = A.lv.w[level], Hl = A.lv.h[level];
        const float dsx = 1.0f / (float)Wl, dsy = 1.0f / (float)Hl;
        const float oxLo = (float)(-hw) * dsx, oxHi = (float)hw * dsx, oyLo = (float)(-hw) * dsy, oyHi = (float)hw * dsy;
        float ox[NPL], oy[NPL], I0[NPL], I0x[NPL], I0y[NPL];
        // The level's first patch fill is known here (the iterate does not move between levels), so its loads go out
        // TOGETHER with the frame-0 samples below: one round trip to the freshly written pyramid instead of two.
        int rx0 = 0, ry0 = 0;
        bool patchValid = false;
        cs_texel tvL[4];
        const bool preFill = !dead && R * R <= 256;
        if (preFill) {
            rx0 = footprint_floor(X1x + oxLo, Wl) - CS_PATCH_MARGIN;
            ry0 = footprint_floor(X1y + oyLo, Hl) - CS_PATCH_MARGIN;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = lane + 64 * u;
                const int ly = idx / R, lx = idx - ly * R;
                const int gx = cs_clampi(rx0 + lx, 0, Wl - 1), gy = cs_clampi(ry0 + ly, 0, Hl - 1);
                tvL[u] = L1[(size_t)gy * Wl + gx];  // (clamped: in range even when idx >= R * R)
            }
        }
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            int p = lane + 64 * q;
            int py = p / fwid, px = p - py * fwid;
            ox[q] = (float)(px - hw) * dsx;
            oy[q] = (float)(py - hw) * dsy;
            I0[q] = I0x[q] = I0y[q] = 0.0f;
            if (!dead && p < nPix) sample(L0, Wl, Hl, X0x + ox[q], X0y + oy[q], I0[q], I0x[q], I0y[q]);
        }
        // the (I0^2 + lambda |grad I0|^2 + 8 delta) window sum does not change within a level
        float fLevel = 0;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            if (lane + 64 * q < nPix) {
                float m0 = sqrtf(I0x[q] * I0x[q] + I0y[q] * I0y[q]);
                fLevel += (I0[q] * I0[q] + A.lambda * m0 * m0) + A.delta * 8.0f;
            }
        }
        fLevel = cs_wave_sum(fLevel);
        if (preFill) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = lane + 64 * u;
                if (idx < R * R) patch[idx] = tvL[u];
            }
            patchValid = true;
            if (PROBE) ++nReload;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        for (int iter = 1; iter <= A.nIter; ++iter) {
            ++pass;
            const cs_granule* src = A.gran + (size_t)(pass - 1) * A.N + nbSlot;
            const unsigned want = tagBase + pass;
            if (PROBE) tm0 = __builtin_amdgcn_s_memtime();
            float J1[NPL], J1x[NPL], J1y[NPL];
            if (!dead) {
                // footprint extent of the whole window (the coordinate map is monotone in the window offset)
                const int iLo = footprint_floor(X1x + oxLo, Wl), iHi = footprint_floor(X1x + oxHi, Wl) + 1;
                const int jLo = footprint_floor(X1y + oyLo, Hl), jHi = footprint_floor(X1y + oyHi, Hl) + 1;
                if (!patchValid || iLo < rx0 || jLo < ry0 || iHi >= rx0 + R || jHi >= ry0 + R) {
                    rx0 = iLo - CS_PATCH_MARGIN;
                    ry0 = jLo - CS_PATCH_MARGIN;
                    // four texels per lane per batch, every load issued before the first LDS store: a loop with a
                    // run-time trip count is not unrolled and would make the 144-texel fill three dependent round trips
                    for (int base = lane; base < R * R; base += 256) {
                        cs_texel tv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int idx = base + 64 * u;
                            const int ly = idx / R, lx = idx - ly * R;
                            const int gx = cs_clampi(rx0 + lx, 0, Wl - 1), gy = cs_clampi(ry0 + ly, 0, Hl - 1);
                            tv[u] = L1[(size_t)gy * Wl + gx];  // (clamped: in range even when idx >= R * R)
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int idx = base + 64 * u;
                            if (idx < R * R) patch[idx] = tv[u];
                        }
                    }
                    patchValid = true;
                    if (PROBE) ++nReload;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                J1[q] = J1x[q] = J1y[q] = 0.0f;
                if (!dead && lane + 64 * q < nPix)
                    sample_patch(patch, R, rx0, ry0, Wl, Hl, X1x + ox[q], X1y + oy[q], J1[q], J1x[q], J1y[q]);
            }
            if (PROBE) {
                asm volatile("" : "+v"(J1[0]));
                tm1 = __builtin_amdgcn_s_memtime();
                tTex += tm1 - tm0;
                tm0 = tm1;
            }
            __builtin_amdgcn_s_setprio(2);
            // ---- everything that does not need the neighbours ------------------------------------------
            float a = 0, b = 0, c = 0, d = 0, e_ = 0, r0 = 0, r1 = 0, r2s = 0, ssd = 0;
            const float f = fLevel;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                if (lane + 64 * q < nPix) {
                    const float I1 = J1[q], I1x = J1x[q], I1y = J1y[q];
                    float ex = beta * I0[q] - I1;
                    float gx = (beta * I0x[q] + I1x) * whx / 2.0f;
                    float gy = (beta * I0y[q] + I1y) * why / 2.0f;
                    float m0 = sqrtf(I0x[q] * I0x[q] + I0y[q] * I0y[q]);
                    float m1 = sqrtf(I1x * I1x + I1y * I1y);
                    a += gx * gx;
                    b += gx * gy;
                    c += gx * (-I0[q]);
                    d += gy * gy;
                    e_ += gy * (-I0[q]);
                    r0 += ex * gx;
                    r1 += ex * gy;
                    r2s += -ex * I0[q] + A.lambda * m0 * (m1 - beta * m0);
                    ssd += ex * ex;
                }
            }
            // The first poll goes out here: a poll is a ~0.4 us fabric round trip and every pass needs at least one, so it
            // flies under the ten wave folds and the adjugate and has landed when they are done.
            // Priority staircase within a pass: 1 while sampling the patch, 2 for the window arithmetic, 3 from here (the
            // first poll, the folds, the sweep, the solve) to the publish.  The closer a wave is to publishing the granule
            // its neighbours wait for, the earlier it gets the SIMD's issue slots over its co-resident waves; every level
            // is above the foreign waves (pose, BA: priority 0).  192-CU partition: 99.4 us flat priority, 94.5 with the
            // sweep..publish section raised, 90.9 from the first poll, 89.5 with the staircase.
            __builtin_amdgcn_s_setprio(3);
            cs_granule got = gran_load(src);
            if (PROBE) ++nPoll;
            cs_wave_sum4(a, b, c, d);
            cs_wave_sum4(e_, r0, r1, r2s);
            const float SSD = cs_wave_sum(ssd);
            const CsGainSolve S = cs_gain_solve_prepare(a, b, c, d, e_, f, r0, r1);
            // thresholds: v3d_gpuklt.cpp:271-279
            const bool real = (iter == A.nIter) && (iter != 1);
            const float sqrConvThr = real ? A.sqrConvThr : 1000000.0f;
            const float ssdThr = real ? A.ssdThr : 1000000.0f;
            const float vr0 = real ? A.vr[0] : -1.0f, vr1 = real ? A.vr[1] : -1.0f;
            const float vr2 = real ? A.vr[2] : 2.0f, vr3 = real ? A.vr[3] : 2.0f;
            float invalidEarly = ((S.det < 0.00001f) || (SSD > ssdThr)) ? 1.0f : 0.0f;
            CsGainSolve Sp = S;
            // pin the prepared solve in registers here: without this the compiler sinks the whole adjugate and the
            // IEEE division below the sweep, back onto the hand-off's critical path
            asm volatile("; solve prepared" : "+v"(Sp.rcp), "+v"(Sp.pX), "+v"(Sp.pY), "+v"(Sp.pZ), "+v"(Sp.C_), "+v"(Sp.E_),
                         "+v"(Sp.F_), "+v"(invalidEarly), "+v"(r2s));
            if (PROBE) {
                tm1 = __builtin_amdgcn_s_memtime();
                tMath += tm1 - tm0;
                tm0 = tm1;
            }
            // ---- sweep the neighbours' granules of the previous pass ----------------------------------------
            float nbBeta = beta;
            {
                unsigned spins = 0;
#pragma nounroll
                while (!__all(!polls || ((unsigned)(got >> 32) == want))) {
                    for (int z = 0; z < A.pollGap; ++z) __builtin_amdgcn_s_sleep(1);
                    got = gran_load(src);
                    if (PROBE) ++nPoll;
                    if (++spins > (1u << 20)) {
                        if (lane == 0) atomicExch(A.err, 1);
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
            float newX = -1.0f, newY = -1.0f, newB = -1.0f;
            if (!dead) {
                const float bsum = cs_gain_bsum(nbBeta, beta);
                float dX, dY, dZ;
                cs_gain_solve_finish(Sp, r2s, (float)nPix, A.delta, bsum, dX, dY, dZ);
                const float nX = X1x + dX, nY = X1y + dY;
                const float ux = dX * whx, uy = dY * why;
                const float sqrLen = ux * ux + uy * uy;
                bool invalid = (invalidEarly != 0.0f);
                invalid = invalid || (sqrLen > sqrConvThr);
                invalid = invalid || (nX < vr0 || nY < vr1) || (nX > vr2 || nY > vr3);
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
            if (lane == 0) gran_store(A.gran + (size_t)pass * A.N + k, want + 1u, beta);
            __builtin_amdgcn_s_setprio(1);
            if (PROBE) {
                tm1 = __builtin_amdgcn_s_memtime();
                tPost += tm1 - tm0;
            }
        }
    }
    if (lane == 0) {
        A.outLast[3 * k] = X1x;
        A.outLast[3 * k + 1] = X1y;
        A.outLast[3 * k + 2] = beta;
        A.outPrev[3 * k] = pX;
        A.outPrev[3 * k + 1] = pY;
        A.outPrev[3 * k + 2] = pB;
        if (A.dest) {  // what k_post_track would do in its own launch
            cs_klt_feature* dst = A.dest + k;
            if (X1x >= 0) {
                dst->status = 0;
                dst->pos[0] = X1x;
                dst->pos[1] = X1y;
                dst->gain = beta;
                dst->fed = -1;  // the tracked count is taken by the consumer of dest[] (one hot atomic word would
                                // serialise the 2000 waves that finish together: ~88 atomics/us)
                if (A.doSuppress && X1y >= 0.0f) {  // v3d_gpuklt.cpp:444-447
                    const float fx = floorf(X1x * (float)A.W), fy = floorf(X1y * (float)A.H);
                    if (fx < (float)A.W && fy < (float)A.H) A.corner[(size_t)(int)fy * A.W + (int)fx] = -1e30f;
                }
            } else {
                dst->status = -1;
                dst->fed = -1;
            }
        }
        if (PROBE && A.probe) {
            unsigned long long* o = A.probe + 8 * (size_t)k;
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
}

// glClear of the blue channel to 1, v3d_gpuklt.cpp:223-227
__global__ void k_reset_beta(float* feat, int N) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N) feat[3 * k + 2] = 1.0f;
}

}  // namespace

int cs_launch_track_nogain(const cs_texel* pyr0, const cs_texel* pyr1, const CsPyrLayout& lay, int levelSkip, int hw,
                           int nIterShader, float margin, float convThr, float ssdThr, int N, const float* featIn,
                           float* featOut, hipStream_t stream) {
    CsTrackLevels lv;
    lv.L = lay.L;
    for (int l = 0; l < lay.L; ++l) {
        lv.w[l] = lay.w[l];
        lv.h[l] = lay.h[l];
        lv.off[l] = lay.off[l];
    }
    if (levelSkip <= 0) levelSkip = lay.L - 1;  // v3d_gpuklt.h:14
    if (levelSkip <= 0) levelSkip = 1;
    const float W = (float)lay.W, H = (float)lay.H;
    const float sqrConv = convThr * convThr;
    const float vr0 = margin / W, vr1 = margin / H, vr2 = 1.0f - margin / W, vr3 = 1.0f - margin / H;
    const int nPix = (2 * hw + 1) * (2 * hw + 1);
    const int npl = (nPix + 63) / 64;
    dim3 grid((N + 3) / 4), block(256);
#define CS_LAUNCH_NG(NPL)                                                                                          \
    hipLaunchKernelGGL(k_track_nogain<NPL>, grid, block, 0, stream, pyr0, pyr1, lv, lay.W, lay.H, levelSkip, hw,     \
                       nIterShader, sqrConv, ssdThr, vr0, vr1, vr2, vr3, N, featIn, featOut)
    if (npl <= 1) {
        CS_LAUNCH_NG(1);
    } else if (npl <= 2) {
        CS_LAUNCH_NG(2);
    } else if (npl <= 4) {
        CS_LAUNCH_NG(4);
    } else if (npl <= 8) {
        CS_LAUNCH_NG(8);
    } else {
        cs_set_error("windowWidth %d too large (max 45)", 2 * hw + 1);
        return CS_ERR_INVALID;
    }
#undef CS_LAUNCH_NG
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_launch_track_gain_pass(const CsGainPassArgs& a, hipStream_t stream) {
    dim3 grid((a.N + 3) / 4), block(256);
    hipLaunchKernelGGL(k_track_gain_pass, grid, block, 0, stream, a);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_launch_reset_beta(float* feat, int N, hipStream_t stream) {
    hipLaunchKernelGGL(k_reset_beta, dim3((N + 255) / 256), dim3(256), 0, stream, feat, N);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_track_patch_r(int hw) { return 2 * hw + 2 + 2 * CS_PATCH_MARGIN; }

int cs_launch_track_gain_fused(const CsGainFusedArgs& a0, hipStream_t stream) {
    CsGainFusedArgs a = a0;
    const int nPix = (2 * a.hw + 1) * (2 * a.hw + 1);
    const int npl = (nPix + 63) / 64;
    a.patchR = cs_track_patch_r(a.hw);
    const size_t lds = (size_t)4 * a.patchR * a.patchR * sizeof(cs_texel);
    dim3 grid((a.N + 3) / 4), block(256);
    if (npl <= 1 && a.probe) {
        hipLaunchKernelGGL((k_track_gain_fused<1, true>), grid, block, lds, stream, a);
    } else if (npl <= 1) {
        hipLaunchKernelGGL((k_track_gain_fused<1>), grid, block, lds, stream, a);
    } else if (npl <= 2) {
        hipLaunchKernelGGL((k_track_gain_fused<2>), grid, block, lds, stream, a);
    } else if (npl <= 4) {
        hipLaunchKernelGGL((k_track_gain_fused<4>), grid, block, lds, stream, a);
    } else {
        cs_set_error("fused gain tracker: windowWidth %d too large", 2 * a.hw + 1);
        return CS_ERR_INVALID;
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}
