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
//     (klt_tracker_with_gain.cg:64-75), which the reference honours with one launch per step.  Windows up to 15 x 15
//     run in klt_track_rows.hip (several features per wave, all cameras of a group and all passes in one persistent
//     launch); k_track_gain_pass here is the launch-per-step schedule for the window sizes that design does not cover.
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
