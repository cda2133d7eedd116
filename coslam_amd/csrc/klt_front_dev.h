// klt_front_dev.h -- device bodies of the frame front end (level 0 + cornerness, fused levels 1..3), shared by the
// stand-alone kernels (klt_pyramid.hip) and the horizontally fused detector-tail kernels (klt_detect.hip) that build
// the NEXT frame's pyramid next to this frame's non-max / selection stages.  Same arithmetic in both: bit-identical.
#pragma once
#include "klt_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int HALO = 2;  // level-0 filters are 5 taps wide

__device__ __forceinline__ float dec4(float v1, float v2, float v3, float v4) {
    return (((v1 + 3.0f * v2) + 3.0f * v3) + v4) / 8.0f;  // pass2.cg:12
}


// =====================================================================================================
constexpr int FTW = 64, FTH = 8;

template <bool CORNER>
struct CsLevel0Lds {
    static constexpr int CR = CORNER ? 3 : 0;
    static constexpr int RW = FTW + 2 * CR, RH = FTH + 2 * CR;  // level-0 region computed by a block
    float g[RH + 2 * HALO][RW + 2 * HALO];
    float v[RH][RW + 2 * HALO];
    float dv[RH][RW + 2 * HALO];
    float gxy[CORNER ? RH : 1][CORNER ? RW : 1][2];
    float conv[3][CORNER ? FTH : 1][CORNER ? RW : 1];
};

// one workgroup (256 threads) of the level-0 + cornerness stage: tile (bx, by) of an (nbx x nby) tiling
template <bool CORNER>
__device__ __forceinline__ void cs_level0_body(const uint8_t* __restrict__ img, int W, int H, cs_texel* __restrict__ out,
                                               float* __restrict__ corner, float minCornerness, float lox, float loy,
                                               float hix, float hiy, int* ctr, int bx, int by, int nbx, int nby, int tid,
                                               CsLevel0Lds<CORNER>& S) {
    constexpr int CR = CORNER ? 3 : 0;
    constexpr int RW = FTW + 2 * CR, RH = FTH + 2 * CR;
    auto& g = S.g;
    auto& v = S.v;
    auto& dv = S.dv;
    auto& gxy = S.gxy;
    auto& conv = S.conv;
    {
        const int gid = (by * nbx + bx) * 256 + tid;
        if (ctr && gid < 8) ctr[gid] = 0;
    }
    const int x0 = bx * FTW, y0 = by * FTH;
    const int rx0 = x0 - CR, ry0 = y0 - CR;
    {  // the tile's pixel loads in one batch per thread, ahead of the LDS stores
        constexpr int NG = (RH + 2 * HALO) * (RW + 2 * HALO), NB = (NG + 255) / 256;
        uint8_t pv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * u;
            const int ly = i / (RW + 2 * HALO), lx = i - ly * (RW + 2 * HALO);
            const int gx = cs_clampi(rx0 + lx - HALO, 0, W - 1);
            const int gy = cs_clampi(ry0 + ly - HALO, 0, H - 1);
            pv[u] = img[(size_t)gy * W + gx];  // (clamped: in range even when i >= NG)
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * u;
            if (i < NG) {
                const int ly = i / (RW + 2 * HALO), lx = i - ly * (RW + 2 * HALO);
                g[ly][lx] = ((float)pv[u] / 255.0f) * 255.0f;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < RH * (RW + 2 * HALO); i += 256) {
        int ly = i / (RW + 2 * HALO), lx = i - ly * (RW + 2 * HALO);
        float g0 = g[ly][lx], g1 = g[ly + 1][lx], g2 = g[ly + 2][lx], g3 = g[ly + 3][lx], g4 = g[ly + 4][lx];
        v[ly][lx] = ((0.0f * g0 + 0.25f * g1) + 0.5f * g2) + 0.25f * g3;
        dv[ly][lx] = (((-0.125f * g0 + -0.25f * g1) + 0.0f * g2) + 0.25f * g3) + 0.125f * g4;
    }
    __syncthreads();
    for (int i = tid; i < RH * RW; i += 256) {
        int ly = i / RW, lx = i - ly * RW;
        const int gx = rx0 + lx, gy = ry0 + ly;
        if (gx < 0 || gy < 0 || gx >= W || gy >= H) continue;  // cells beyond the border are filled below
        const float* vr = &v[ly][lx];
        const float* dr = &dv[ly][lx];
        float I = ((0.0f * vr[0] + 0.25f * vr[1]) + 0.5f * vr[2]) + 0.25f * vr[3];
        float Ix = (((-0.125f * vr[0] + -0.25f * vr[1]) + 0.0f * vr[2]) + 0.25f * vr[3]) + 0.125f * vr[4];
        float Iy = ((0.0f * dr[0] + 0.25f * dr[1]) + 0.5f * dr[2]) + 0.25f * dr[3];
        const cs_texel t = cs_pack_texel(I, Ix, Iy);
        if (lx >= CR && lx < CR + FTW && ly >= CR && ly < CR + FTH) out[(size_t)gy * W + gx] = t;
        if (CORNER) {
            float rI, rX, rY;
            cs_unpack_texel(t, rI, rX, rY);
            gxy[ly][lx][0] = rX;
            gxy[ly][lx][1] = rY;
        }
    }
    if (!CORNER) return;
    __syncthreads();
    // CLAMP_TO_EDGE for the detector's taps: a region cell beyond the image border repeats the border texel
    if (rx0 < 0 || ry0 < 0 || rx0 + RW > W || ry0 + RH > H) {
        for (int i = tid; i < RH * RW; i += 256) {
            int ly = i / RW, lx = i - ly * RW;
            const int gx = rx0 + lx, gy = ry0 + ly;
            if (gx < 0 || gy < 0 || gx >= W || gy >= H)
            {
                const int sy = cs_clampi(gy, 0, H - 1) - ry0, sx = cs_clampi(gx, 0, W - 1) - rx0;
                gxy[ly][lx][0] = gxy[sy][sx][0];
                gxy[ly][lx][1] = gxy[sy][sx][1];
            }
        }
        __syncthreads();
    }
    // klt_detector_pass1.cg: vertical taps -3..+3 accumulated in that order
    for (int i = tid; i < FTH * RW; i += 256) {
        int ly = i / RW, lx = i - ly * RW;
        float r0 = 0, r1 = 0, r2 = 0;
#pragma unroll
        for (int k = 0; k < 2 * CR + 1; ++k) {
            const float qx = gxy[ly + k][lx][0], qy = gxy[ly + k][lx][1];
            r0 += qx * qx;
            r1 += qx * qy;
            r2 += qy * qy;
        }
        conv[0][ly][lx] = r0;
        conv[1][ly][lx] = r1;
        conv[2][ly][lx] = r2;
    }
    __syncthreads();
    // klt_detector_pass2.cg:12-33
    const int lx = tid & (FTW - 1), x = x0 + lx;
    for (int ly = tid / FTW; ly < FTH; ly += 256 / FTW) {
        int y = y0 + ly;
        if (x >= W || y >= H) continue;
        float a = 0, b = 0, c = 0;
#pragma unroll
        for (int k = 0; k < 2 * CR + 1; ++k) {
            a += conv[0][ly][lx + k];
            b += conv[1][ly][lx + k];
            c += conv[2][ly][lx + k];
        }
        float amc = a - c;
        float cn = 0.5f * ((a + c) - sqrtf(amc * amc + 4.0f * (b * b)));
        cn = fmaxf(cn - minCornerness, 0.0f);
        float stx = ((float)x + 0.5f) / (float)W, sty = ((float)y + 0.5f) / (float)H;
        bool inside = (stx >= lox && sty >= loy) && (stx <= hix && sty <= hiy);
        corner[(size_t)y * W + x] = inside ? cn : 0.0f;
    }
}

// ---- levels 1..NL (NL <= 3) in one launch -----------------------------------------------------------------
// Tile of the coarsest fused level owned by a block: 16 x 2 texels.  Sixteen texels are 128 bytes -- one full cache line
// per stored row at EVERY level (level 3: 16, level 2: 32, level 1: 64 texels wide).  The first version owned 4 x 4
// tiles: 32- and 64-byte row pieces written by different workgroups (on different XCDs) into the same lines, and the
// counters showed 7.99 MB written per dispatch for 0.81 MB of pyramid (profiles/r01_j_pmc_WRITE_SIZE_session3.md).
constexpr int DTW = 16, DTH = 2;

struct CsDownFused {
    int NL;            // destination levels 1..NL
    int shift;         // tap_mode ? -1 : 0
    int w[4], h[4];    // level sizes 0..NL
    long long off[4];  // level offsets in texels
    int cap1, cap2;    // LDS texel capacity of the level-1 / level-2 regions
};

struct CsRange {
    int lo, hi;  // [lo, hi)
};

__host__ __device__ __forceinline__ int cs_tap_base(int o, int n_dst, int n_src) {
    return (int)(((long long)(2 * o + 1) * n_src) / (2 * (long long)n_dst));
}
// source rows the destination rows [r.lo, r.hi) read (4 taps from tap_base - 1 + shift, CLAMP_TO_EDGE)
__host__ __device__ __forceinline__ CsRange cs_tap_range(CsRange r, int n_dst, int n_src, int shift) {
    int lo = cs_tap_base(r.lo, n_dst, n_src) + shift - 1, hi = cs_tap_base(r.hi - 1, n_dst, n_src) + shift + 2;
    lo = lo < 0 ? 0 : (lo > n_src - 1 ? n_src - 1 : lo);
    hi = hi < 0 ? 0 : (hi > n_src - 1 ? n_src - 1 : hi);
    CsRange q = {lo, hi + 1};
    return q;
}
__host__ __device__ __forceinline__ CsRange cs_union(CsRange a, CsRange b) {
    CsRange q = {a.lo < b.lo ? a.lo : b.lo, a.hi > b.hi ? a.hi : b.hi};
    return q;
}
// texels of level l owned by tile t of the coarsest level (the last tile also owns what floor-halving dropped)
__host__ __device__ __forceinline__ CsRange cs_own(int t, int nTiles, int tile, int n_l, int up) {
    CsRange q = {(t * tile) << up, (t == nTiles - 1) ? n_l : ((t + 1) * tile) << up};
    if (q.hi > n_l) q.hi = n_l;
    if (q.lo > n_l) q.lo = n_l;
    return q;
}

__device__ __forceinline__ cs_texel down_one(const cs_texel* __restrict__ src, int pitch, int ox, int oy, int Ws, int Hs,
                                             int Wd, int Hd, int x, int y, int shift) {
    // src is addressed as src[(row - oy) * pitch + (col - ox)] with row/col clamped to the source level first
    const int by = cs_tap_base(y, Hd, Hs) + shift, bx = cs_tap_base(x, Wd, Ws) + shift;
    int r[4], c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = cs_clampi(by - 1 + k, 0, Hs - 1) - oy;
        c[k] = cs_clampi(bx - 1 + k, 0, Ws - 1) - ox;
    }
    float colI[4], colX[4], colY[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float I[4], X[4], Y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cs_unpack_texel(src[(size_t)r[k] * pitch + c[j]], I[k], X[k], Y[k]);
        colI[j] = cs_h2f(cs_f2h(dec4(I[0], I[1], I[2], I[3])));
        colX[j] = cs_h2f(cs_f2h(dec4(X[0], X[1], X[2], X[3])));
        colY[j] = cs_h2f(cs_f2h(dec4(Y[0], Y[1], Y[2], Y[3])));
    }
    return cs_pack_texel(dec4(colI[0], colI[1], colI[2], colI[3]), dec4(colX[0], colX[1], colX[2], colX[3]),
                         dec4(colY[0], colY[1], colY[2], colY[3]));
}

// one 256-thread group of the fused decimation: tile (tx, ty) of the coarsest fused level; reg1 = this group's LDS
// (cap1 + cap2 texels).  `valid` = false: the group has no tile but still takes part in the workgroup barriers.
__device__ __forceinline__ void cs_down_body(cs_texel* __restrict__ pyr, const CsDownFused& F, int tx, int ty, int tid,
                                             cs_texel* reg1, bool valid) {
    cs_texel* reg2 = reg1 + F.cap1;
    const int NL = F.NL;
    // Ranges per level, statically indexed: a loop over a run-time NL put these arrays (and a copy of F) in scratch --
    // 144 B per lane whose spills the write counters billed to this kernel (4.4 MB per camera for 0.81 MB of pyramid).
    const int w0 = F.w[0], h0 = F.h[0], w1 = F.w[1], h1 = F.h[1], w2 = F.w[2], h2 = F.h[2], w3 = F.w[3], h3 = F.h[3];
    const int wN = NL == 3 ? w3 : (NL == 2 ? w2 : w1), hN = NL == 3 ? h3 : (NL == 2 ? h2 : h1);
    const int ntx = (wN + DTW - 1) / DTW, nty = (hN + DTH - 1) / DTH;
    const CsRange none = {0, 0};
    CsRange ownx3 = none, owny3 = none, ownx2 = none, owny2 = none, needx2 = none, needy2 = none;
    if (NL == 3) {
        ownx3 = cs_own(tx, ntx, DTW, w3, 0);
        owny3 = cs_own(ty, nty, DTH, h3, 0);
    }
    if (NL >= 2) {
        ownx2 = cs_own(tx, ntx, DTW, w2, NL - 2);
        owny2 = cs_own(ty, nty, DTH, h2, NL - 2);
        needx2 = NL == 2 ? ownx2 : cs_union(ownx2, cs_tap_range(ownx3, w3, w2, F.shift));
        needy2 = NL == 2 ? owny2 : cs_union(owny2, cs_tap_range(owny3, h3, h2, F.shift));
    }
    const CsRange ownx1 = cs_own(tx, ntx, DTW, w1, NL - 1), owny1 = cs_own(ty, nty, DTH, h1, NL - 1);
    const CsRange needx1 = NL == 1 ? ownx1 : cs_union(ownx1, cs_tap_range(needx2, w2, w1, F.shift));
    const CsRange needy1 = NL == 1 ? owny1 : cs_union(owny1, cs_tap_range(needy2, h2, h1, F.shift));
    // level 1 from level 0 in HBM
    {
        const cs_texel* src = pyr + F.off[0];
        cs_texel* dst = pyr + F.off[1];
        const int nw = needx1.hi - needx1.lo, nh = needy1.hi - needy1.lo;
        for (int i = valid ? tid : nw * nh; i < nw * nh; i += 256) {
            const int ly = i / nw, lx = i - ly * nw;
            const int x = needx1.lo + lx, y = needy1.lo + ly;
            const cs_texel t = down_one(src, w0, 0, 0, w0, h0, w1, h1, x, y, F.shift);
            if (NL > 1) reg1[i] = t;
            if (x >= ownx1.lo && x < ownx1.hi && y >= owny1.lo && y < owny1.hi) dst[(size_t)y * w1 + x] = t;
        }
    }
    if (NL < 2) return;
    __syncthreads();
    {
        cs_texel* dst = pyr + F.off[2];
        const int sw = needx1.hi - needx1.lo;
        const int nw = needx2.hi - needx2.lo, nh = needy2.hi - needy2.lo;
        for (int i = valid ? tid : nw * nh; i < nw * nh; i += 256) {
            const int ly = i / nw, lx = i - ly * nw;
            const int x = needx2.lo + lx, y = needy2.lo + ly;
            const cs_texel t = down_one(reg1, sw, needx1.lo, needy1.lo, w1, h1, w2, h2, x, y, F.shift);
            if (NL > 2) reg2[i] = t;
            if (x >= ownx2.lo && x < ownx2.hi && y >= owny2.lo && y < owny2.hi) dst[(size_t)y * w2 + x] = t;
        }
    }
    if (NL < 3) return;
    __syncthreads();
    {
        cs_texel* dst = pyr + F.off[3];
        const int sw = needx2.hi - needx2.lo;
        const int nw = ownx3.hi - ownx3.lo, nh = owny3.hi - owny3.lo;
        for (int i = valid ? tid : nw * nh; i < nw * nh; i += 256) {
            const int ly = i / nw, lx = i - ly * nw;
            const int x = ownx3.lo + lx, y = owny3.lo + ly;
            dst[(size_t)y * w3 + x] =
                down_one(reg2, sw, needx2.lo, needy2.lo, w2, h2, w3, h3, x, y, F.shift);
        }
    }
}


// host side: the argument block and launch shape of the fused decimation
inline int cs_down_fused_plan(const CsPyrLayout& lay, int tap_mode, CsDownFused* Fp, int* ntxp, int* ntyp, size_t* ldsp) {
    CsDownFused& F = *Fp;
    memset(&F, 0, sizeof(F));
    F.NL = lay.L - 1 < 3 ? lay.L - 1 : 3;
    F.shift = tap_mode ? -1 : 0;
    for (int l = 0; l <= F.NL; ++l) {
        F.w[l] = lay.w[l];
        F.h[l] = lay.h[l];
        F.off[l] = lay.off[l];
    }
    // LDS capacity: the largest level-1 / level-2 region over all tiles (same range arithmetic as the kernel)
    const int NL = F.NL;
    const int ntx = (F.w[NL] + DTW - 1) / DTW, nty = (F.h[NL] + DTH - 1) / DTH;
    int cap[4] = {0, 0, 0, 0};
    int mw[4] = {0, 0, 0, 0}, mh[4] = {0, 0, 0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        const int nt = pass ? nty : ntx;
        for (int t = 0; t < nt; ++t) {
            CsRange need[4];
            for (int l = NL; l >= 1; --l) {
                const int nl = pass ? F.h[l] : F.w[l];
                CsRange own = cs_own(t, nt, pass ? DTH : DTW, nl, NL - l);
                need[l] = (l == NL) ? own
                                    : cs_union(own, cs_tap_range(need[l + 1], pass ? F.h[l + 1] : F.w[l + 1], nl, F.shift));
                const int ext = need[l].hi - need[l].lo;
                if (pass) {
                    if (ext > mh[l]) mh[l] = ext;
                } else {
                    if (ext > mw[l]) mw[l] = ext;
                }
            }
        }
    }
    for (int l = 1; l <= NL; ++l) cap[l] = mw[l] * mh[l];
    F.cap1 = (NL > 1) ? cap[1] : 0;
    F.cap2 = (NL > 2) ? cap[2] : 0;
    *ntxp = ntx;
    *ntyp = nty;
    *ldsp = (size_t)(F.cap1 + F.cap2) * sizeof(cs_texel);
    if (*ldsp > 32 * 1024) {
        cs_set_error("fused pyramid: %zu B of LDS needed", *ldsp);
        return CS_ERR_INVALID;
    }
    return CS_OK;
}

}  // namespace
