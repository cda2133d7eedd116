// klt_pyramid.hip -- (I, Ix, Iy) pyramid with presmoothing 1, gfx950.
//
// Replaces PyramidWithDerivativesCreator::buildPyramidForGrayscaleImage
// (src/tracking/CGKLT/v3d_gpupyramid.cpp:366-429) and the three Cg passes it schedules
// (Shaders/pyramid_with_derivative_pass1v.cg:63-83, pass1h.cg:96-128, pass2.cg:6-12).
//
// Design: the reference makes 2 + 2(L-1) full-screen passes through an RGBA16F intermediate
// (8 B/px written and re-read).  Here level 0 is ONE kernel: a u8 tile with a 2-px halo is staged in
// LDS, the vertical and horizontal 5-tap filters run out of LDS, and each lane stores one 8-byte texel
// (I,Ix,Iy,0 as binary16) so a wave writes 512 contiguous bytes.  Each coarser level is ONE kernel that
// applies the vertical and the horizontal [1 3 3 1]/8 with the intermediate binary16 rounding the
// reference's RGB16F temporary imposes, without ever materialising that temporary in HBM.
// HBM traffic per frame = W*H bytes in + 8 B per texel out (+ the 4x4 gather of the level above,
// L1/L2-resident).
#include "klt_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int TW = 64;  // tile width  (one wave = one output row segment)
constexpr int TH = 8;   // tile height
constexpr int HALO = 2;

__global__ __launch_bounds__(256) void k_pyr_level0(const uint8_t* __restrict__ img, int W, int H,
                                                    cs_texel* __restrict__ out) {
    __shared__ float g[TH + 2 * HALO][TW + 2 * HALO];  // luminance * 255 (exact integers)
    __shared__ float v[TH][TW + 2 * HALO];             // vertical [1 2 1]/4
    __shared__ float dv[TH][TW + 2 * HALO];            // vertical [-1 -2 0 2 1]/8

    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int tid = threadIdx.x;

    for (int i = tid; i < (TH + 2 * HALO) * (TW + 2 * HALO); i += 256) {
        int ly = i / (TW + 2 * HALO), lx = i - ly * (TW + 2 * HALO);
        int gx = cs_clampi(x0 + lx - HALO, 0, W - 1);  // CLAMP_TO_EDGE, v3d_gpubase.cpp:220-223
        int gy = cs_clampi(y0 + ly - HALO, 0, H - 1);
        // LUMINANCE8 -> [0,1] -> *255 as pass1v.cg:79-80 does; exact for every byte value
        g[ly][lx] = ((float)img[(size_t)gy * W + gx] / 255.0f) * 255.0f;
    }
    __syncthreads();

    for (int i = tid; i < TH * (TW + 2 * HALO); i += 256) {
        int ly = i / (TW + 2 * HALO), lx = i - ly * (TW + 2 * HALO);
        float g0 = g[ly][lx], g1 = g[ly + 1][lx], g2 = g[ly + 2][lx], g3 = g[ly + 3][lx], g4 = g[ly + 4][lx];
        v[ly][lx] = ((0.0f * g0 + 0.25f * g1) + 0.5f * g2) + 0.25f * g3;
        dv[ly][lx] = (((-0.125f * g0 + -0.25f * g1) + 0.0f * g2) + 0.25f * g3) + 0.125f * g4;
    }
    __syncthreads();

    const int lx = tid & (TW - 1);
    const int x = x0 + lx;
    for (int ly = tid / TW; ly < TH; ly += 256 / TW) {
        int y = y0 + ly;
        if (x < W && y < H) {
            const float* vr = &v[ly][lx];  // vr[2] is the centre
            const float* dr = &dv[ly][lx];
            float I = ((0.0f * vr[0] + 0.25f * vr[1]) + 0.5f * vr[2]) + 0.25f * vr[3];
            float Ix = (((-0.125f * vr[0] + -0.25f * vr[1]) + 0.0f * vr[2]) + 0.25f * vr[3]) + 0.125f * vr[4];
            float Iy = ((0.0f * dr[0] + 0.25f * dr[1]) + 0.5f * dr[2]) + 0.25f * dr[3];
            out[(size_t)y * W + x] = cs_pack_texel(I, Ix, Iy);
        }
    }
}

__device__ __forceinline__ int tap_base(int o, int n_dst, int n_src) {
    return (int)(((long long)(2 * o + 1) * n_src) / (2 * (long long)n_dst));
}

__device__ __forceinline__ float dec4(float v1, float v2, float v3, float v4) {
    return (((v1 + 3.0f * v2) + 3.0f * v3) + v4) / 8.0f;  // pass2.cg:12
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

}  // namespace

int cs_launch_pyramid(const uint8_t* d_img, const CsPyrLayout& lay, cs_texel* d_pyr, int tap_mode,
                      hipStream_t stream) {
    dim3 g0((lay.W + TW - 1) / TW, (lay.H + TH - 1) / TH);
    hipLaunchKernelGGL(k_pyr_level0, g0, dim3(256), 0, stream, d_img, lay.W, lay.H, d_pyr + lay.off[0]);
    for (int l = 1; l < lay.L; ++l) {
        dim3 g((lay.w[l] + 63) / 64, (lay.h[l] + 3) / 4);
        hipLaunchKernelGGL(k_pyr_down, g, dim3(256), 0, stream, d_pyr + lay.off[l - 1], lay.w[l - 1], lay.h[l - 1],
                           d_pyr + lay.off[l], lay.w[l], lay.h[l], tap_mode ? -1 : 0);
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}
