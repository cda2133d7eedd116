// klt_pyramid.hip -- (I, Ix, Iy) pyramid with presmoothing 1, gfx950.
//
// Replaces PyramidWithDerivativesCreator::buildPyramidForGrayscaleImage
// (src/tracking/CGKLT/v3d_gpupyramid.cpp:366-429) and the three Cg passes it schedules
// (Shaders/pyramid_with_derivative_pass1v.cg:63-83, pass1h.cg:96-128, pass2.cg:6-12).
//
// Design: the reference makes 2 + 2(L-1) full-screen passes through an RGBA16F intermediate (8 B/px written and
// re-read) and the detector another two for its cornerness map.  Here the frame front end is TWO launches (see the
// block comment above k_pyr_level0_corner): level 0 together with the cornerness map out of one LDS tile, and levels
// 1..3 together out of a recomputed dependency cone in LDS; each lane stores one 8-byte texel (I, Ix, Iy, 0 as
// binary16).  The vertical and horizontal [1 3 3 1]/8 keep the intermediate binary16 rounding the reference's RGB16F
// temporary imposes, without ever materialising that temporary in HBM.  Levels >= 4 (rare: nLevels 6) take one
// gather kernel each (k_pyr_down).
#include "klt_front_dev.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int tap_base(int o, int n_dst, int n_src) {
    return (int)(((long long)(2 * o + 1) * n_src) / (2 * (long long)n_dst));
}

// one output texel per lane: 4x4 gather from the level above
__global__ __launch_bounds__(256) void k_pyr_down(const cs_texel* __restrict__ src, int Ws, int Hs,
                                                  cs_texel* __restrict__ dst, int Wd, int Hd, int shift) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= Wd || y >= Hd) return;
    const int by = tap_base(y, Hd, Hs) + shift, bx = tap_base(x, Wd, Ws) + shift;
    int r[4], c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = cs_clampi(by - 1 + k, 0, Hs - 1);
        c[k] = cs_clampi(bx - 1 + k, 0, Ws - 1);
    }
    float colI[4], colX[4], colY[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float I[4], X[4], Y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cs_unpack_texel(src[(size_t)r[k] * Ws + c[j]], I[k], X[k], Y[k]);
        // vertical pass result lives in an RGB16F render target: round to binary16 (v3d_gpupyramid.cpp:285-293)
        colI[j] = cs_h2f(cs_f2h(dec4(I[0], I[1], I[2], I[3])));
        colX[j] = cs_h2f(cs_f2h(dec4(X[0], X[1], X[2], X[3])));
        colY[j] = cs_h2f(cs_f2h(dec4(Y[0], Y[1], Y[2], Y[3])));
    }
    dst[(size_t)y * Wd + x] = cs_pack_texel(dec4(colI[0], colI[1], colI[2], colI[3]),
                                            dec4(colX[0], colX[1], colX[2], colX[3]),
                                            dec4(colY[0], colY[1], colY[2], colY[3]));
}



// =====================================================================================================
// Frame front end in TWO launches (a dependent launch costs ~4.7 us on this stack, the five kernels above plus the
// two per-frame memsets were ~30 us of a 130 us frame):
//   k_pyr_level0_corner   level 0 AND the detector's cornerness map (klt_detector_pass1/pass2.cg) out of one LDS
//                         tile: the (Ix, Iy) the 7x7 structure tensor needs are the fp16-rounded texels this block
//                         has just produced (own tile + 3-px halo, recomputed, never re-read from HBM); the same
//                         launch zeroes the frame's counters;
//   k_pyr_down_fused      levels 1..3 in one launch: every block owns an 8x4 tile of the coarsest of them and
//                         recomputes the (one-sided, 2-texel) dependency cone below it in LDS.
// Arithmetic, operation order and the fp16 roundings are those of k_pyr_level0 / k_pyr_down / k_cornerness:
// results are bit-identical to the separate kernels.
// (The bodies live in klt_front_dev.h: the detector tail re-uses them for the next frame's front.)
template <bool CORNER>
__global__ __launch_bounds__(256) void k_pyr_level0_corner(const uint8_t* __restrict__ img, int W, int H,
                                                           cs_texel* __restrict__ out, float* __restrict__ corner,
                                                           float minCornerness, float lox, float loy, float hix,
                                                           float hiy, int* ctr, unsigned long long* gran, int nGran) {
    __shared__ CsLevel0Lds<CORNER> S;
    cs_level0_body<CORNER>(img, W, H, out, corner, minCornerness, lox, loy, hix, hiy, ctr, gran, nGran, blockIdx.x,
                           blockIdx.y, gridDim.x, gridDim.y, threadIdx.x, S);
}

__global__ __launch_bounds__(256) void k_pyr_down_fused(cs_texel* __restrict__ pyr, CsDownFused F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char down_smem[];
    cs_down_body(pyr, F, blockIdx.x, blockIdx.y, threadIdx.x, (cs_texel*)down_smem, true);
}

}  // namespace

// levels >= first (those the fused decimation does not cover): one gather launch each
int cs_launch_pyr_down_from(const CsPyrLayout& lay, cs_texel* d_pyr, int tap_mode, int first, hipStream_t stream) {
    for (int l = first; l < lay.L; ++l) {
        dim3 g((lay.w[l] + 63) / 64, (lay.h[l] + 3) / 4);
        hipLaunchKernelGGL(k_pyr_down, g, dim3(256), 0, stream, d_pyr + lay.off[l - 1], lay.w[l - 1], lay.h[l - 1],
                           d_pyr + lay.off[l], lay.w[l], lay.h[l], tap_mode ? -1 : 0);
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// pyramid (+ cornerness map, + zeroing of the frame's counters) in two launches
int cs_launch_frame_front(const uint8_t* d_img, const CsPyrLayout& lay, cs_texel* d_pyr, int tap_mode, float* corner_out,
                          float minCornerness, float margin, int* ctr, unsigned long long* gran, int nGran,
                          hipStream_t stream) {
    const int W = lay.W, H = lay.H;
    dim3 g0((W + FTW - 1) / FTW, (H + FTH - 1) / FTH);
    if (corner_out) {
        const float lox = margin / (float)W, loy = margin / (float)H;
        const float hix = 1.0f - margin / (float)W, hiy = 1.0f - margin / (float)H;
        hipLaunchKernelGGL(k_pyr_level0_corner<true>, g0, dim3(256), 0, stream, d_img, W, H, d_pyr + lay.off[0], corner_out,
                           minCornerness, lox, loy, hix, hiy, ctr, gran, nGran);
    } else {
        hipLaunchKernelGGL(k_pyr_level0_corner<false>, g0, dim3(256), 0, stream, d_img, W, H, d_pyr + lay.off[0],
                           (float*)nullptr, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, ctr, gran, nGran);
    }
    if (lay.L >= 2) {
        CsDownFused F;
        int ntx, nty;
        size_t lds;
        int rc = cs_down_fused_plan(lay, tap_mode, &F, &ntx, &nty, &lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_pyr_down_fused, dim3(ntx, nty), dim3(256), lds, stream, d_pyr, F);
        rc = cs_launch_pyr_down_from(lay, d_pyr, tap_mode, F.NL + 1, stream);
        if (rc) return rc;
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}
