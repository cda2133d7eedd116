// launch_floor.hip -- diagnostic: cost of a dependent kernel launch in a long chain, by kernarg size and by what the
// kernel touches first.  hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { long long v[52]; };  // 416 B, the size of BaDev
__global__ void k_small(int* flag) { if (threadIdx.x == 0 && flag[0] == 12345) flag[1] = 1; }
__global__ void k_big(Big b) { int* flag = (int*)b.v[0]; if (threadIdx.x == 0 && flag[0] == 12345) flag[1] = 1; }
__global__ void k_big_ptr(const Big* b) { int* flag = (int*)b->v[0]; if (threadIdx.x == 0 && flag[0] == 12345) flag[1] = 1; }
__global__ void k_empty() {}
__global__ void k_rw(int* flag) { if (threadIdx.x == 0) flag[2] = flag[2] + 1; }  // read-modify-write a word the previous kernel wrote
template <class F>
static void timeit(const char* name, F launch, int grid, int n = 400) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch(grid);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) launch(grid);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // the same chain from a graph
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    for (int i = 0; i < n; ++i) launch(grid, s);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float msg; hipEventElapsedTime(&msg, e0, e1);
    printf("%-34s grid %4d: eager %.2f us/launch, graph %.2f us/launch\n", name, grid, ms * 1e3 / n, msg * 1e3 / n);
}
int main() {
    int* flag; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
    Big b; for (auto& x : b.v) x = 0; b.v[0] = (long long)flag;
    Big* db; hipMalloc(&db, sizeof(Big)); hipMemcpy(db, &b, sizeof(Big), hipMemcpyHostToDevice);
    for (int grid : {1, 125, 500}) {
        timeit("empty kernel", [&](int g, hipStream_t s = 0) { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s); }, grid);
        timeit("8-byte kernarg, reads a flag", [&](int g, hipStream_t s = 0) { hipLaunchKernelGGL(k_small, dim3(g), dim3(256), 0, s, flag); }, grid);
        timeit("416-byte kernarg, reads a flag", [&](int g, hipStream_t s = 0) { hipLaunchKernelGGL(k_big, dim3(g), dim3(256), 0, s, b); }, grid);
        timeit("pointer to 416-byte struct", [&](int g, hipStream_t s = 0) { hipLaunchKernelGGL(k_big_ptr, dim3(g), dim3(256), 0, s, db); }, grid);
        timeit("RMW of the previous kernel's word", [&](int g, hipStream_t s = 0) { hipLaunchKernelGGL(k_rw, dim3(g), dim3(256), 0, s, flag); }, grid);
    }
    return 0;
}
