// handoff_mesh.hip -- diagnostic micro-benchmark: the hand-off pattern of the persistent gain tracker with no
// arithmetic.  N waves on a fw x fh slot grid, each pass every wave waits for its 8 neighbours' granules of the
// previous pass and publishes its own.  Reports microseconds per pass for several publish / poll primitives.
//   hipcc --offload-arch=gfx950 -O3 -o handoff_mesh handoff_mesh.hip && ./handoff_mesh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

enum { PUB_SC1 = 0, PUB_ATOMIC = 1, PUB_SYS = 2 };
enum { POLL_SC1 = 0, POLL_ATOMIC = 1, POLL_SYS = 2 };

template <int PUB>
__device__ __forceinline__ void publish(u64* p, unsigned tag, unsigned val) {
    u64 g = ((u64)tag << 32) | val;
    if (PUB == PUB_SC1) __hip_atomic_store((gu64*)p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PUB == PUB_ATOMIC) __hip_atomic_exchange((gu64*)p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PUB == PUB_SYS) __hip_atomic_store((gu64*)p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int POLL>
__device__ __forceinline__ u64 poll(const u64* p) {
    if (POLL == POLL_SC1) return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (POLL == POLL_ATOMIC) return __hip_atomic_fetch_or((gu64*)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// work: dependent VALU filler between publish and first poll (cycles ~ 8 * work)
template <int PUB, int POLL>
__global__ __launch_bounds__(256) void k_mesh(u64* gran, int fw, int fh, int N, int passes, int nNb, int work,
                                              int stride, int* err, float* sink) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= N) return;
    u64* g0 = gran;
    u64* g1 = gran + (size_t)N * stride;
    if (lane == 0) publish<PUB>(g0 + (size_t)k * stride, 1u, 1u);
    const int si = k % fw, sj = k / fw;
    int nb = k;
    if (lane < nNb) {
        const int dxs[8] = {1, -1, 0, 0, 1, -1, 1, -1}, dys[8] = {0, 0, 1, -1, 1, -1, -1, 1};
        int x = min(max(si + dxs[lane], 0), fw - 1), y = min(max(sj + dys[lane], 0), fh - 1);
        nb = y * fw + x;
    }
    const bool polls = nb != k;
    float acc = (float)k;
    for (unsigned pass = 1; pass <= (unsigned)passes; ++pass) {
        for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
        const u64* src = (((pass - 1) & 1u) ? g1 : g0) + (size_t)nb * stride;
        unsigned spins = 0;
        u64 got;
        for (;;) {
            got = poll<POLL>(src);
            if (__all(!polls || (unsigned)(got >> 32) >= pass)) break;
            if (++spins > (1u << 18)) {
                if (lane == 0) atomicExch(err, 1);
                break;
            }
        }
        acc += (float)(unsigned)got;
        if (lane == 0) publish<PUB>(((pass & 1u) ? g1 : g0) + (size_t)k * stride, pass + 1u, (unsigned)pass);
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int PUB, int POLL>
static void run(const char* name, int nNb, int work, int stride, int fw = 50, int fh = 40, int passes = 40) {
    const int N = fw * fh;
    u64* gran;
    int* err;
    float* sink;
    hipMalloc(&gran, sizeof(u64) * 2 * N * stride);
    hipMalloc(&err, 4);
    hipMalloc(&sink, 4);
    hipMemset(err, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
        hipMemsetAsync(gran, 0, sizeof(u64) * 2 * N * stride, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mesh<PUB, POLL>), dim3((N + 3) / 4), dim3(256), 0, 0, gran, fw, fh, N, passes, nNb, work, stride,
                           err, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 3) {
            sum += ms;
            best = ms < best ? ms : best;
        }
    }
    int herr = 0;
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("%-28s nb=%d work=%4d stride=%d grid=%dx%d: avg %.2f us/pass (best %.2f)%s\n", name, nNb, work, stride, fw, fh,
           sum / reps * 1e3 / passes, best * 1e3 / passes, herr ? "  TIMEOUT" : "");
    hipFree(gran);
    hipFree(err);
    hipFree(sink);
}

// push model: every wave owns one 64-byte inbox line per parity (8 granules, one per neighbour direction); a producer
// stores its granule into each consumer's inbox (8 stores from lanes 0..7), the consumer polls ONE line.
// Neighbours outside the grid do not take part (no clamping) so that pull and push solve the same problem.
template <int MODE>  // 0 = pull (as k_mesh, no clamping), 1 = push
__global__ __launch_bounds__(256) void k_mesh2(u64* gran, int fw, int fh, int N, int passes, int work, int stride,
                                               int* err, float* sink, int post) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= N) return;
    const size_t parity = (size_t)N * stride * (MODE ? 8 : 1);
    const int si = k % fw, sj = k / fw;
    const int dxs[8] = {1, -1, 0, 0, 1, -1, 1, -1}, dys[8] = {0, 0, 1, -1, 1, -1, -1, 1};
    bool polls = false;
    int nb = k, cons = k;
    if (lane < 8) {
        int x = si + dxs[lane], y = sj + dys[lane];
        if (x >= 0 && x < fw && y >= 0 && y < fh) {
            polls = true;
            nb = y * fw + x;
        }
        int cx = si - dxs[lane], cy = sj - dys[lane];
        cons = (cx >= 0 && cx < fw && cy >= 0 && cy < fh) ? cy * fw + cx : -1;
    }
    // pull: my granule at gran[k*stride]; push: consumer c's inbox slot `lane` at gran[(c*8 + lane)*stride]
    if (MODE == 0) {
        if (lane == 0) publish<PUB_SC1>(gran + (size_t)k * stride, 1u, 1u);
    } else if (lane < 8 && cons >= 0) {
        publish<PUB_SC1>(gran + ((size_t)cons * 8 + lane) * stride, 1u, 1u);
    }
    float acc = (float)k;
    for (unsigned pass = 1; pass <= (unsigned)passes; ++pass) {
        for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
        const u64* base = gran + ((pass - 1) & 1u) * parity;
        const u64* src = MODE ? base + ((size_t)k * 8 + (lane & 7)) * stride : base + (size_t)nb * stride;
        unsigned spins = 0;
        u64 got;
        for (;;) {
            got = poll<POLL_SC1>(src);
            if (__all(!polls || (unsigned)(got >> 32) >= pass)) break;
            if (++spins > (1u << 18)) {
                if (lane == 0) atomicExch(err, 1);
                break;
            }
        }
        acc += (float)(unsigned)got;
        for (int w = 0; w < post; ++w) acc = acc * 1.0001f + 0.5f;
        u64* ob = gran + (pass & 1u) * parity;
        if (MODE == 0) {
            if (lane == 0) publish<PUB_SC1>(ob + (size_t)k * stride, pass + 1u, (unsigned)pass);
        } else if (lane < 8 && cons >= 0) {
            publish<PUB_SC1>(ob + ((size_t)cons * 8 + lane) * stride, pass + 1u, (unsigned)pass);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
static void run2(const char* name, int work, int stride, int fw = 50, int fh = 40, int passes = 40, int post = 0) {
    const int N = fw * fh;
    const size_t words = (size_t)2 * N * stride * (MODE ? 8 : 1);
    u64* gran;
    int* err;
    float* sink;
    hipMalloc(&gran, sizeof(u64) * words);
    hipMalloc(&err, 4);
    hipMalloc(&sink, 4);
    hipMemset(err, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
        hipMemsetAsync(gran, 0, sizeof(u64) * words, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mesh2<MODE>), dim3((N + 3) / 4), dim3(256), 0, 0, gran, fw, fh, N, passes, work, stride, err, sink, post);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 3) {
            sum += ms;
            best = ms < best ? ms : best;
        }
    }
    int herr = 0;
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("post=%d ", post);
    printf("%-28s work=%4d stride=%d grid=%dx%d: avg %.2f us/pass (best %.2f)%s\n", name, work, stride, fw, fh,
           sum / reps * 1e3 / passes, best * 1e3 / passes, herr ? "  TIMEOUT" : "");
    hipFree(gran);
    hipFree(err);
    hipFree(sink);
}

// hierarchical pull: a workgroup = 4 waves = 4 consecutive slots of a row; ONE wave polls the 14 external neighbours of
// the strip (3 rows x 6 columns minus the strip) and hands them to its mates through LDS; mates' own values go through
// LDS too.  Chip-wide pollers: N/4 instead of N.
__global__ __launch_bounds__(256) void k_mesh_hier(u64* gran, int fw, int fh, int N, int passes, int work, int* err,
                                                  float* sink, int post) {
    __shared__ volatile unsigned extTag;            // pass number whose external neighbours are complete in LDS
    __shared__ volatile unsigned ownTag[2][4];      // [parity][wave]: tag of the mates' published value
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + wv;
    const bool live = k < N;
    u64* g0 = gran;
    u64* g1 = gran + N;
    if (threadIdx.x == 0) extTag = 0;
    if (lane == 0) { ownTag[0][wv] = 1; ownTag[1][wv] = 0; }
    if (live && lane == 0) publish<PUB_SC1>(g0 + k, 1u, 1u);
    __syncthreads();
    // external neighbour list of the strip (wave 0 polls): rows sj-1..sj+1, cols si0-1..si0+4, outside the strip, inside the grid
    const int k0 = blockIdx.x * 4, si0 = k0 % fw, sj = k0 / fw;
    int nb = -1;
    if (wv == 0 && lane < 18) {
        const int r = lane / 6, c = lane % 6;
        const int x = si0 - 1 + c, y = sj - 1 + r;
        const bool inStrip = (r == 1 && c >= 1 && c <= 4);
        if (!inStrip && x >= 0 && x < fw && y >= 0 && y < fh && (y * fw + x) < N) nb = y * fw + x;
    }
    float acc = (float)k;
    for (unsigned pass = 1; pass <= (unsigned)passes; ++pass) {
        for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
        const u64* src = ((pass - 1) & 1u) ? g1 : g0;
        unsigned spins = 0;
        if (wv == 0) {
            for (;;) {
                u64 got = (nb >= 0) ? poll<POLL_SC1>(src + nb) : ((u64)pass << 32);
                if (__all((unsigned)(got >> 32) >= pass)) break;
                if (++spins > (1u << 18)) { if (lane == 0) atomicExch(err, 1); break; }
            }
            if (lane == 0) extTag = pass;
        } else {
            while (extTag < pass) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 20)) { if (lane == 0) atomicExch(err, 2); break; }
            }
        }
        // mates' values of the previous pass (LDS)
        for (int m = 0; m < 4; ++m) {
            while (ownTag[(pass - 1) & 1u][m] < pass) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 20)) { if (lane == 0) atomicExch(err, 3); break; }
            }
        }
        for (int w = 0; w < post; ++w) acc = acc * 1.0001f + 0.5f;
        if (lane == 0) {
            if (live) publish<PUB_SC1>(((pass & 1u) ? g1 : g0) + k, pass + 1u, (unsigned)pass);
            ownTag[pass & 1u][wv] = pass + 1u;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static void run_hier(int work, int post, int fw = 50, int fh = 40, int passes = 40) {
    const int N = fw * fh;
    u64* gran; int* err; float* sink;
    hipMalloc(&gran, sizeof(u64) * 2 * N); hipMalloc(&err, 4); hipMalloc(&sink, 4); hipMemset(err, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0; const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
        hipMemsetAsync(gran, 0, sizeof(u64) * 2 * N, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mesh_hier, dim3((N + 3) / 4), dim3(256), 0, 0, gran, fw, fh, N, passes, work, err, sink, post);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 3) { sum += ms; best = ms < best ? ms : best; }
    }
    int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("post=%d hierarchical (1 poller / 4 waves)   work=%4d grid=%dx%d: avg %.2f us/pass (best %.2f)%s\n", post, work, fw, fh,
           sum / reps * 1e3 / passes, best * 1e3 / passes, herr ? "  TIMEOUT/ERR" : "");
    hipFree(gran); hipFree(err); hipFree(sink);
}

// XCD-local variant: the slot grid is cut into 8 regions (2 x 4), region r is served by the workgroups that run on
// XCD r (block b -> XCD b % 8 is assumed for the mapping and CHECKED with XCC_ID: mismatching blocks are counted).
// Every wave publishes twice: a plain store into a "local" granule array (stays in its XCD's L2; same-XCD readers
// poll it with L1-bypassing loads served by that L2) and a write-through sc1 store into the "global" array for
// readers on other XCDs.  A reader picks the array per neighbour by comparing regions.
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf; }
__global__ __launch_bounds__(256) void k_mesh_xcd(u64* granG, u64* granL, int fw, int fh, int N, int passes, int work,
                                                 int* err, float* sink, int post, int* mismatch) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // block -> (region, item): region = blockIdx % 8, item = blockIdx / 8; region r = 2 x 4 layout of (fw/2) x (fh/4) slots
    const int region = blockIdx.x & 7, item = blockIdx.x >> 3;
    const int rw = fw / 2, rh = fh / 4;                  // 25 x 10
    const int perRegion = rw * rh;                       // 250 slots -> 63 items of 4 waves (last one ragged)
    const int local = item * 4 + wv;
    if (local >= perRegion) return;
    const int rx = region & 1, ry = region >> 1;
    const int si = rx * rw + local % rw, sj = ry * rh + local / rw;
    const int k = sj * fw + si;
    if (threadIdx.x == 0 && xcc_id() != (unsigned)region) atomicAdd(mismatch, 1);
    u64* G0 = granG; u64* G1 = granG + N; u64* L0 = granL; u64* L1 = granL + N;
    if (lane == 0) {
        publish<PUB_SC1>(G0 + k, 1u, 1u);
        ((volatile u64*)L0)[k] = ((u64)1u << 32) | 1u;
    }
    const int dxs[8] = {1, -1, 0, 0, 1, -1, 1, -1}, dys[8] = {0, 0, 1, -1, 1, -1, -1, 1};
    bool polls = false, sameX = false;
    int nb = k;
    if (lane < 8) {
        int x = si + dxs[lane], y = sj + dys[lane];
        if (x >= 0 && x < fw && y >= 0 && y < fh) {
            polls = true;
            nb = y * fw + x;
            sameX = ((x / rw) + 2 * (y / rh)) == region;
        }
    }
    float acc = (float)k;
    for (unsigned pass = 1; pass <= (unsigned)passes; ++pass) {
        for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
        const u64* srcG = (((pass - 1) & 1u) ? G1 : G0) + nb;
        const u64* srcL = (((pass - 1) & 1u) ? L1 : L0) + nb;
        const u64* src = sameX ? srcL : srcG;
        unsigned spins = 0;
        u64 got;
        for (;;) {
            got = poll<POLL_SC1>(src);   // sc1 load: bypasses L1, served by this XCD's L2 (local array) or the fabric
            if (__all(!polls || (unsigned)(got >> 32) >= pass)) break;
            if (++spins > (1u << 18)) { if (lane == 0) atomicExch(err, 1); break; }
        }
        acc += (float)(unsigned)got;
        for (int w = 0; w < post; ++w) acc = acc * 1.0001f + 0.5f;
        if (lane == 0) {
            const u64 g = ((u64)(pass + 1u) << 32) | pass;
            ((volatile u64*)((pass & 1u) ? L1 : L0))[k] = g;                       // plain store: stays in this XCD's L2
            publish<PUB_SC1>(((pass & 1u) ? G1 : G0) + k, pass + 1u, (unsigned)pass);  // write-through for the others
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static void run_xcd(int work, int post, int passes = 40) {
    const int fw = 50, fh = 40, N = fw * fh;
    u64 *gG, *gL; int *err, *mm; float* sink;
    hipMalloc(&gG, sizeof(u64) * 2 * N); hipMalloc(&gL, sizeof(u64) * 2 * N); hipMalloc(&err, 4); hipMalloc(&mm, 4); hipMalloc(&sink, 4);
    hipMemset(err, 0, 4); hipMemset(mm, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0; const int reps = 20;
    for (int r = 0; r < reps + 3; ++r) {
        hipMemsetAsync(gG, 0, sizeof(u64) * 2 * N, 0); hipMemsetAsync(gL, 0, sizeof(u64) * 2 * N, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mesh_xcd, dim3(8 * 63), dim3(256), 0, 0, gG, gL, fw, fh, N, passes, work, err, sink, post, mm);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 3) { sum += ms; best = ms < best ? ms : best; }
    }
    int herr = 0, hmm = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hmm, mm, 4, hipMemcpyDeviceToHost);
    printf("post=%d XCD-local regions (dual publish)    work=%4d grid=50x40: avg %.2f us/pass (best %.2f)  xcc mismatches %d%s\n", post, work,
           sum / reps * 1e3 / passes, best * 1e3 / passes, hmm, herr ? "  TIMEOUT/STALE" : "");
}

int main() {
    for (int work : {0, 50, 75}) {
        const int post = work ? 20 : 0;
        run2<0>("pull", work, 1, 50, 40, 40, post);
        run_xcd(work, post);
    }
    return 0;
}
