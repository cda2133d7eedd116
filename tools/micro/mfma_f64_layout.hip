// Which element of D = A B + C does each lane / register of v_mfma_f64_16x16x4f64 hold?  (A: lane l supplies A[l % 16][l / 16],
// B: lane l supplies B[l / 16][l % 16] -- assumed, verified by the decoded products.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
    const int l = threadIdx.x;
    const int i = l % 16, kk = l / 16;
    const double a = (kk == 1) ? (double)(i + 1) : 0.0;            // A[i][1] = i + 1
    const double b = (kk == 1) ? (double)(100 * (i + 1)) : 0.0;    // B[1][j] = 100 (j + 1)
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int q = 0; q < 4; ++q) out[4 * l + q] = c[q];
}
int main() {
    double* d;
    hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) {
        printf("lane %2d:", l);
        for (int q = 0; q < 4; ++q) {
            const int v = (int)h[4 * l + q];   // (i + 1) * 100 * (j + 1)
            int fi = -1, fj = -1;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j)
                    if ((i + 1) * 100 * (j + 1) == v && fi < 0) { fi = i; fj = j; }
            printf("  q%d -> %d", q, v);
        }
        printf("\n");
    }
    return 0;
}
