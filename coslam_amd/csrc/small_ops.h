// small_ops.h -- fills and device-to-device copies of the frame loop's streams as ONE kernel launch for a whole list.
//
// hipMemsetAsync / hipMemcpyAsync between kernels cost far more than the bytes they move: each is its own dispatch with a
// barrier on either side, measured on the pose stream 4-5 us of execution plus 12-13 us of idle stream before the next operation
// (profiles/r04_ab_runs.txt), against back-to-back dispatch for ordinary kernels.  The key-frame path alone issued 14
// of them.  cs_small_ops collects up to CS_SMALL_OPS operations (copy or fill, any byte count and alignment) and runs them as
// one launch: blockIdx.y = operation, the x blocks stride over its 16-byte words, the unaligned head and tail byte by byte.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace cs_small {
namespace {

constexpr int CS_SMALL_OPS = 12;
struct Op {
    unsigned char* dst;
    const unsigned char* src;   // null: fill with `value`
    size_t bytes;
    unsigned value;             // fill byte, replicated
};
struct Ops {
    Op op[CS_SMALL_OPS];
};

__global__ __launch_bounds__(256) void k_small_ops(Ops L) {
    const Op o = L.op[blockIdx.y];
    if (o.bytes == 0) return;
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nThreads = (size_t)gridDim.x * 256;
    const bool aligned = ((reinterpret_cast<uintptr_t>(o.dst) | (o.src ? reinterpret_cast<uintptr_t>(o.src) : 0)) & 15) == 0;
    if (aligned) {
        const size_t nVec = o.bytes / 16;
        uint4* d = reinterpret_cast<uint4*>(o.dst);
        if (o.src) {
            const uint4* s = reinterpret_cast<const uint4*>(o.src);
            for (size_t q = tid; q < nVec; q += nThreads) d[q] = s[q];
        } else {
            const unsigned v = o.value * 0x01010101u;
            const uint4 vv = make_uint4(v, v, v, v);
            for (size_t q = tid; q < nVec; q += nThreads) d[q] = vv;
        }
        for (size_t q = nVec * 16 + tid; q < o.bytes; q += nThreads) o.dst[q] = o.src ? o.src[q] : (unsigned char)o.value;
    } else {
        for (size_t q = tid; q < o.bytes; q += nThreads) o.dst[q] = o.src ? o.src[q] : (unsigned char)o.value;
    }
}

// a list under construction; run() launches it (nothing when empty)
struct List {
    Ops L;
    int n = 0;
    size_t maxBytes = 0;
    bool copy(void* dst, const void* src, size_t bytes) { return add(dst, src, bytes, 0); }
    bool fill(void* dst, unsigned byteValue, size_t bytes) { return add(dst, nullptr, bytes, byteValue & 0xffu); }
    bool add(void* dst, const void* src, size_t bytes, unsigned value) {
        if (n >= CS_SMALL_OPS) return false;
        L.op[n].dst = static_cast<unsigned char*>(dst), L.op[n].src = static_cast<const unsigned char*>(src);
        L.op[n].bytes = bytes, L.op[n].value = value;
        if (bytes > maxBytes) maxBytes = bytes;
        ++n;
        return true;
    }
    hipError_t run(hipStream_t s) {
        if (n == 0) return hipSuccess;
        for (int k = n; k < CS_SMALL_OPS; ++k) L.op[k].dst = nullptr, L.op[k].src = nullptr, L.op[k].bytes = 0, L.op[k].value = 0;
        size_t blocks = (maxBytes / 16 + 255) / 256;   // one 16-byte word per thread for the largest operation, at most 512 workgroups
        if (blocks < 1) blocks = 1;
        if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(k_small_ops, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, s, L);
        n = 0, maxBytes = 0;
        return hipGetLastError();
    }
};

}  // namespace
}  // namespace cs_small
