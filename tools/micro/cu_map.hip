// cu_map.hip -- diagnostic: which physical (XCC, SE, CU) the bits of a hipExtStreamCreateWithCUMask mask select.
// hipcc --offload-arch=gfx950 -O3 -o cu_map cu_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <map>
__global__ void k_where(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // burn a little time so that the dispatcher has to spread the grid
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
}
static void probe(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int nb = 4096;
    unsigned* d; hipMalloc(&d, 8 * nb);
    hipLaunchKernelGGL(k_where, dim3(nb), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nb); hipMemcpy(h.data(), d, 8 * nb, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;  // xcc -> set of (se, sh, cu)
    for (int i = 0; i < nb; ++i) {
        unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per[xcc].insert(se * 100 + sh * 16 + cu);
    }
    int total = 0;
    printf("%s:\n", name);
    for (auto& kv : per) {
        printf("  xcc %u: %zu CUs:", kv.first, kv.second.size());
        for (unsigned v : kv.second) printf(" se%u.cu%u", v / 100, v % 16);
        printf("\n");
        total += (int)kv.second.size();
    }
    printf("  total %d CUs seen\n", total);
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int T = p.multiProcessorCount, W = (T + 31) / 32;
    auto range = [&](int a, int n) { std::vector<uint32_t> m(W, 0u); for (int c = a; c < a + n; ++c) m[c >> 5] |= 1u << (c & 31); return m; };
    probe("bits [0,8)", range(0, 8));
    probe("bits [0,32)", range(0, 32));
    probe("bits [8,16)", range(8, 8));
    probe("bits [192,256)", range(192, 64));
    probe("bits [0,192)", range(0, 192));
    { std::vector<uint32_t> m(W, 0u); for (int c = 0; c < T; ++c) if (c % 8 == 0) m[c >> 5] |= 1u << (c & 31); probe("bits c%8==0", m); }
    { std::vector<uint32_t> m(W, 0u); for (int c = 0; c < T; ++c) if (c % 8 >= 6) m[c >> 5] |= 1u << (c & 31); probe("bits c%8>=6", m); }
    return 0;
}
