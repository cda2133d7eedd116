// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the KLT tracker
// (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel reads (or writes) a known number of bytes exactly once; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// and compare the counter (KB) with the printed byte counts.
//   k_read16       16 B per lane, coalesced streaming (the guide's calibrated case: counter = 1/2 of the bytes)
//   k_read8        8 B per lane, coalesced streaming (a texel per lane, consecutive lanes consecutive texels)
//   k_gather8      8 B per lane, every lane in a different 128-B line of a 2-D array walked tile by tile (a 7x7 window's
//                  bilinear footprint rows: short runs of texels, rows a pitch apart)
//   k_write8       8 B per lane coalesced stores
// hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_read16(const double2* p, size_t n, double* out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x + p[i].y;
    if (acc == 12345.678) out[0] = acc;
}
__global__ void k_read8(const double* p, size_t n, double* out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678) out[0] = acc;
}
// a 2-D array of 8-byte texels, pitch W; a wave reads an 8 x 8 texel footprint (lane = (row, col)), footprints tile the array
__global__ void k_gather8(const double* p, int W, int H, double* out) {
    const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
    const int tilesX = W / 8, tiles = tilesX * (H / 8);
    double acc = 0;
    for (int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < tiles; t += gridDim.x * (blockDim.x >> 6)) {
        const int ty = t / tilesX, tx = t - ty * tilesX;
        acc += p[(size_t)(8 * ty + r) * W + 8 * tx + c];
    }
    if (acc == 12345.678) out[0] = acc;
}
__global__ void k_write8(double* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}

int main() {
    const size_t bytes = 512ull << 20;  // beyond the 256 MiB Infinity Cache
    double *a, *out;
    hipMalloc(&a, bytes);
    hipMalloc(&out, 64);
    hipMemset(a, 0, bytes);
    const int W = 8192, H = (int)(bytes / 8 / W);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const double2*)a, bytes / 16, out);
        hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, a, bytes / 8, out);
        hipLaunchKernelGGL(k_gather8, dim3(4096), dim3(256), 0, 0, a, W, H, out);
        hipLaunchKernelGGL(k_write8, dim3(4096), dim3(256), 0, 0, a, bytes / 8);
    }
    hipDeviceSynchronize();
    printf("every kernel touches %zu bytes (= %.1f KB) exactly once per launch\n", bytes, bytes / 1024.0);
    return 0;
}
