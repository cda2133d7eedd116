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

// one output texel per lane: 4x4 gather from the level above; blockIdx.z = camera
struct CsPyrPtrs {
    cs_texel* pyr[CS_MAX_CAMS];
};
__global__ __launch_bounds__(256) void k_pyr_down(CsPyrPtrs P, long long offSrc, int Ws, int Hs, long long offDst, int Wd,
                                                  int Hd, int shift) {
    const cs_texel* __restrict__ src = P.pyr[blockIdx.z] + offSrc;
    cs_texel* __restrict__ dst = P.pyr[blockIdx.z] + offDst;
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
struct CsFrontBatch {
    int W, H;
    float minCornerness, lox, loy, hix, hiy;
    CsFrontCam cam[CS_MAX_CAMS];
};

template <bool CORNER>
__global__ __launch_bounds__(256) void k_pyr_level0_corner(CsFrontBatch B) {
    __shared__ CsLevel0Lds<CORNER> S;
    const CsFrontCam& C = B.cam[blockIdx.z];
    cs_level0_body<CORNER>(C.img, B.W, B.H, C.pyr, C.corner, B.minCornerness, B.lox, B.loy, B.hix, B.hiy, C.ctr, blockIdx.x,
                           blockIdx.y, gridDim.x, gridDim.y, threadIdx.x, S);
}

__global__ __launch_bounds__(256) void k_pyr_down_fused(CsPyrPtrs P, CsDownFused F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char down_smem[];
    cs_down_body(P.pyr[blockIdx.z], F, blockIdx.x, blockIdx.y, threadIdx.x, (cs_texel*)down_smem, true);
}

}  // namespace

// levels >= first (those the fused decimation does not cover): one gather launch each, all cameras
static int launch_pyr_down_from(const CsPyrPtrs& P, int n, const CsPyrLayout& lay, int tap_mode, int first,
                                hipStream_t stream) {
    for (int l = first; l < lay.L; ++l) {
        dim3 g((lay.w[l] + 63) / 64, (lay.h[l] + 3) / 4, n);
        hipLaunchKernelGGL(k_pyr_down, g, dim3(256), 0, stream, P, (long long)lay.off[l - 1], lay.w[l - 1], lay.h[l - 1],
                           (long long)lay.off[l], lay.w[l], lay.h[l], tap_mode ? -1 : 0);
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_launch_pyr_down_tail(cs_texel* const* pyrs, int n, const CsPyrLayout& lay, int tap_mode, int first, hipStream_t stream) {
    CsPyrPtrs P;
    memset(&P, 0, sizeof(P));
    for (int c = 0; c < n; ++c) P.pyr[c] = pyrs[c];
    return launch_pyr_down_from(P, n, lay, tap_mode, first, stream);
}

// pyramid (+ cornerness map, + zeroing of the frame's counters) of n cameras in two launches
int cs_launch_frame_front(const CsFrontCam* cams, int n, const CsPyrLayout& lay, int tap_mode, bool withCorner,
                          float minCornerness, float margin, hipStream_t stream) {
    if (n < 1 || n > CS_MAX_CAMS) {
        cs_set_error("frame front: %d cameras (1..%d)", n, CS_MAX_CAMS);
        return CS_ERR_INVALID;
    }
    const int W = lay.W, H = lay.H;
    CsFrontBatch B;
    memset(&B, 0, sizeof(B));
    B.W = W;
    B.H = H;
    CsPyrPtrs P;
    memset(&P, 0, sizeof(P));
    for (int c = 0; c < n; ++c) {
        B.cam[c] = cams[c];
        P.pyr[c] = cams[c].pyr;
    }
    dim3 g0((W + FTW - 1) / FTW, (H + FTH - 1) / FTH, n);
    if (withCorner) {
        B.minCornerness = minCornerness;
        B.lox = margin / (float)W;
        B.loy = margin / (float)H;
        B.hix = 1.0f - margin / (float)W;
        B.hiy = 1.0f - margin / (float)H;
        hipLaunchKernelGGL(k_pyr_level0_corner<true>, g0, dim3(256), 0, stream, B);
    } else {
        hipLaunchKernelGGL(k_pyr_level0_corner<false>, g0, dim3(256), 0, stream, B);
    }
    if (lay.L >= 2) {
        CsDownFused F;
        int ntx, nty;
        size_t lds;
        int rc = cs_down_fused_plan(lay, tap_mode, &F, &ntx, &nty, &lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_pyr_down_fused, dim3(ntx, nty, n), dim3(256), lds, stream, P, F);
        rc = launch_pyr_down_from(P, n, lay, tap_mode, F.NL + 1, stream);
        if (rc) return rc;
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}
