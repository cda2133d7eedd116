// launch_floor2.hip -- diagnostic: why do the BA's trivial kernels take ~4.5 us each inside its graph when a trivial
// kernel chain runs at 1.6 us/launch?  Variants: state-word kernels like k_outer_end, alternating distinct kernels,
// wide grids, and a producer with many workgroups before a single-workgroup consumer.
// hipcc --offload-arch=gfx950 -O3 -o launch_floor2 launch_floor2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct St { double lambda, cost; int a, b, c, d, e, f; };
struct Big { St* st; double* part; long long pad[50]; };
__global__ void k_state(Big B) { St* st = B.st; if (st->a) return; st->b += 1; st->c = 0; if (!st->d) st->e = 1; }
__global__ void k_state2(Big B) { St* st = B.st; if (st->a) return; st->f += 1; st->c = 1; if (st->d) st->e = 0; }
__global__ __launch_bounds__(256) void k_producer(Big B) {  // every workgroup writes one partial
    if (B.st->a) return;
    if (threadIdx.x == 0) B.part[blockIdx.x] = (double)blockIdx.x + B.st->lambda;
}
__global__ __launch_bounds__(256) void k_consumer(Big B, int n) {  // one workgroup sums them, updates the state
    __shared__ double red[4];
    if (B.st->a) return;
    double c = 0;
    for (int q = threadIdx.x; q < n; q += 256) c += B.part[q];
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { B.st->cost = red[0] + red[1] + red[2] + red[3]; B.st->lambda *= 1.0000001; }
}
template <class F>
static void graph_time(const char* name, F enqueue, int nLaunches) {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    enqueue(s);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %.2f us/launch\n", name, ms * 1e3 / (5.0 * nLaunches));
}
int main() {
    St* st; hipMalloc(&st, sizeof(St)); hipMemset(st, 0, sizeof(St));
    double* part; hipMalloc(&part, 8 * 1024);
    Big B; B.st = st; B.part = part; for (auto& x : B.pad) x = 0;
    const int N = 200;
    graph_time("state-word kernel, 1 thread", [&](hipStream_t s) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_state, dim3(1), dim3(1), 0, s, B); }, N);
    graph_time("two alternating state-word kernels", [&](hipStream_t s) { for (int i = 0; i < N; ++i) { if (i & 1) hipLaunchKernelGGL(k_state2, dim3(1), dim3(1), 0, s, B); else hipLaunchKernelGGL(k_state, dim3(1), dim3(1), 0, s, B); } }, N);
    graph_time("state-word kernel, grid 125 x 256", [&](hipStream_t s) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_state, dim3(125), dim3(256), 0, s, B); }, N);
    for (int g : {1, 8, 125, 500})
        for (int rep = 0; rep < 1; ++rep) {
            char nm[96]; snprintf(nm, sizeof nm, "producer (%d workgroups) -> consumer (1 workgroup), per launch", g);
            graph_time(nm, [&](hipStream_t s) { for (int i = 0; i < N / 2; ++i) { hipLaunchKernelGGL(k_producer, dim3(g), dim3(256), 0, s, B); hipLaunchKernelGGL(k_consumer, dim3(1), dim3(256), 0, s, B, g); } }, N);
        }
    return 0;
}
