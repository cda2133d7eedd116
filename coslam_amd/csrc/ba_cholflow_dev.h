// ba_cholflow_dev.h -- Cholesky solve of a LARGE reduced camera system (order > SB_MAX_ORDER: it no longer fits one
// workgroup's LDS) as ONE launch: a dataflow over block columns (included by ba.hip inside its anonymous namespace, after the
// sb_* helpers of the LDS solver).
//
// The round-1 path is right-looking in HBM with two launches per 32-column block (k_chol_panel re-factors the diagonal block in
// every workgroup, k_chol_trail updates 64 x 64 tiles) plus a one-workgroup triangular solve: 43 dependent launches, 2.0 ms at
// order 672 (BASELINE cfg5) for 101 MFLOP.  Here workgroup j OWNS block column j (16 columns): its blocks (i, j), i >= j, and
// its slice of the right-hand side -- one more block row, so the forward substitution is part of the factorisation -- live in
// LDS for the whole launch (order 672: <= 43 blocks = 97 KB).  Left-looking, driven by flags in HBM:
//   for k < j, as column k is published:  A_ij -= L_ik L_jk^T   every wave keeps ITS blocks' accumulators in registers
//                                         (block b of the column belongs to wave b mod 16) and takes the MFMA operands of
//                                         L_ik / L_jk straight from the published column (stored in operand order: one
//                                         coalesced 512-byte load per operand) -- no LDS traffic, no barrier per column;
//   then  L_jj = chol(A_jj), L_jj^-1 (wave 0, f64 DPP: sb_factor_diag), L_ij = A_ij L_jj^-T (one MFMA product per block,
//         sb_panel_mfma), publish, release flag j.
// The chain is one link per block column: flag -> operand loads -> update of the diagonal block -> factor + inverse -> panel ->
// publish.  Workgroups far behind the front consume published columns at load latency and wait at the front.  Back
// substitution runs the same way in reverse over the blocks each workgroup already holds: x_j = L_jj^-T (y_j - sum_{i>j}
// L_ij^T x_i), the partial sums taken as the x_i arrive.
// Workgroups are dispatched in index order and column j only waits for lower indices (then, in the back substitution, for
// workgroups that are already running), so the launch cannot deadlock on residency; the polls are bounded all the same.
// Same arithmetic building blocks as k_solve_blocked; the summation order over k is the same left-to-right order.

constexpr int CF_NT = 1024, CF_WAVES = 16;
constexpr int CF_MAXB = 5;  // blocks of a column per wave: columns of up to 80 blocks (incl. the right-hand side), LDS permitting
constexpr int CF_MAX_BLOCKS = 66;  // 66 x 2304 B = 152 KB of LDS -> orders up to 16 x 65 = 1040

struct CholFlow {
    double* pub;   // [NB][NB + 1][256]: column k, block row i (row NB = the right-hand side's block), MFMA operand order
    double* xpub;  // [NB][16]
    int* flagL;    // [NB]
    int* flagX;    // [NB]
    int NB;
};

__host__ __device__ constexpr size_t cf_lds_bytes(int nbTot) { return sizeof(double) * ((size_t)nbTot * SBLK + CF_WAVES * 16 + 16) + 16; }

// Hand-off without cache maintenance: the published values and the flags are relaxed agent-scope atomics (sc1: performed at
// the coherent level, never served from a stale L1 / non-coherent L2 line), the producer waits for its stores' acknowledgements
// (s_waitcnt vmcnt(0)) before it raises the flag, the consumer's loads are issued after the flag was seen.  An acquire / release
// fence pair at agent scope costs an L2 write-back and an invalidate per link (17 us per block column instead of 8).
__device__ __forceinline__ double cf_ld(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void cf_st(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void cf_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bool cf_wait(const int* flag) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 21)) return false;  // ~ a second: the producer is gone; give up instead of hanging the queue
    }
    asm volatile("" ::: "memory");
    return true;
}

__global__ void k_cholflow_begin(BaDev D, CholFlow F) {
    if (!BA_ACTIVE(D)) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) D.st->chol_ok = 1;
    if (t < F.NB) {
        F.flagL[t] = 0;
        F.flagX[t] = 0;
    }
}

__global__ __launch_bounds__(CF_NT) void k_cholflow(BaDev D, CholFlow F) {
    if (CS_SOLVE_PRIO) __builtin_amdgcn_s_setprio(CS_SOLVE_PRIO);
    if (!BA_ACTIVE(D)) return;
#ifdef CF_PROBE
    const unsigned long long cfT0 = wall_clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) double cf_sm[];
    __shared__ int okSh;
    const int n = D.n, NB = F.NB, j = blockIdx.x, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int nb = NB - j, nbTot = nb + 1;  // blocks (j..NB-1, j) and the right-hand side's block (index nb)
    double* blk = cf_sm;                      // [nbTot][SBLK]
    double* wpart = cf_sm + (size_t)nbTot * SBLK;  // [16 waves][16]
    double* tvec = wpart + CF_WAVES * 16;          // [16]
    if (tid == 0) okSh = 1;
    // ---- this wave's blocks, as MFMA accumulators (D layout: register q = row lg + 4 q, column lr) ----
    sb_d4 acc[CF_MAXB];
#pragma unroll
    for (int u = 0; u < CF_MAXB; ++u) {
        const int b = wv + CF_WAVES * u;
        acc[u] = (sb_d4){0.0, 0.0, 0.0, 0.0};
        if (b < nb) {
            const int gc = 16 * j + lr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gr = 16 * (j + b) + lg + 4 * q;
                double v = (gr == gc) ? 1.0 : 0.0;  // identity padding beyond n
                if (gr < n && gc < n) v = D.S[(size_t)gr * n + gc];
                acc[u][q] = v;
            }
        } else if (b == nb) {  // right-hand side: row 0 of the block
            const int gc = 16 * j + lr;
            if (lg == 0 && gc < n) acc[u][0] = D.rhs[gc];
        }
    }
    // ---- left-looking updates from the published columns ----
    bool alive = true;
    for (int k = 0; k < j && alive; ++k) {
        alive = cf_wait(F.flagL + k);
        if (!alive) break;
        const double* col = F.pub + (size_t)k * (NB + 1) * 256;
        const double* PJ = col + (size_t)j * 256;
        double pj[4], pi[CF_MAXB][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) pj[s] = cf_ld(PJ + 64 * s + lane);
#pragma unroll
        for (int u = 0; u < CF_MAXB; ++u) {
            const int b = wv + CF_WAVES * u;
            if (b < nbTot) {
                const double* PI = col + (size_t)(b == nb ? NB : j + b) * 256;
#pragma unroll
                for (int s = 0; s < 4; ++s) pi[u][s] = cf_ld(PI + 64 * s + lane);
            }
        }
#pragma unroll
        for (int u = 0; u < CF_MAXB; ++u) {
            const int b = wv + CF_WAVES * u;
            if (b < nbTot) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pi[u][s], pj[s], acc[u], 0, 0, 0);
            }
        }
    }
    // ---- the column is up to date: to LDS, factor, panel ----
#pragma unroll
    for (int u = 0; u < CF_MAXB; ++u) {
        const int b = wv + CF_WAVES * u;
        if (b < nbTot) {
#pragma unroll
            for (int q = 0; q < 4; ++q) blk[(size_t)b * SBLK + (lg + 4 * q) * SP + lr] = acc[u][q];
        }
    }
    if (wv == 0) {  // (block 0 was written by this wave itself: no workgroup barrier in front of the factorisation)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!sb_factor_diag(blk, lane) || !alive) okSh = 0;  // block 0 now holds L_jj^-1
    }
    __syncthreads();
    double* mycol = F.pub + (size_t)j * (NB + 1) * 256;
#pragma unroll
    for (int u = 0; u < CF_MAXB; ++u) {
        const int b = wv + CF_WAVES * u;
        if (b >= 1 && b < nbTot) {
            double* P = blk + (size_t)b * SBLK;
            // P <- P L_jj^-T (sb_panel_mfma with explicit pointers)
            double pa[4], pb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                pa[s] = P[lr * SP + 4 * s + lg];
                pb[s] = blk[lr * SP + 4 * s + lg];
            }
            sb_d4 c = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[s], pb[s], c, 0, 0, 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) P[(lg + 4 * q) * SP + lr] = c[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // publish in operand order: element (lane, s) = L[lr][4 s + lg]
            double* out = mycol + (size_t)(b == nb ? NB : j + b) * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) cf_st(out + 64 * s + lane, P[lr * SP + 4 * s + lg]);
        }
    }
    cf_stores_done();
    __syncthreads();
    if (tid == 0) {
        if (!okSh) D.st->chol_ok = 0;
        if (!alive) D.st->solverTimeout = 1;  // a scheduling stall, not a non-positive-definite system: reported as such
        __hip_atomic_store(F.flagL + j, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CF_PROBE
        if ((j == NB - 1 || j == NB / 2 || j == 1) && D.st->nIterTotal == 2) printf("cholflow column %d published at %llu (x 10 ns)\n", j, wall_clock64() - cfT0);
#endif
    }
    // ---- back substitution: x_j = L_jj^-T (y_j - sum_{i > j} L_ij^T x_i); this wave takes the blocks it owns, lanes 0..15 =
    //      the 16 entries of the partial sum ----
    double part = 0.0;
    for (int u = CF_MAXB - 1; u >= 0; --u) {
        const int b = wv + CF_WAVES * u;
        if (b < 1 || b >= nb) continue;
        const int i = j + b;
        if (alive) alive = cf_wait(F.flagX + i);
        const double* Lb = blk + (size_t)b * SBLK;
        const double* xi = F.xpub + 16 * (size_t)i;
        if (lane < 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) part += Lb[r * SP + lane] * cf_ld(xi + r);
        }
    }
    if (lane < 16) wpart[wv * 16 + lane] = part;
    __syncthreads();
    if (wv == 0) {
        const double* yb = blk + (size_t)nb * SBLK;  // row 0 = y_j
        if (lane < 16) {
            double t = yb[lane];
#pragma unroll
            for (int w = 0; w < CF_WAVES; ++w) t -= wpart[w * 16 + lane];
            tvec[lane] = t;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 16) {
            double x = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) x += blk[r * SP + lane] * tvec[r];  // (L_jj^-1)^T t
            cf_st(F.xpub + 16 * (size_t)j + lane, x);
            if (16 * j + lane < n) D.rhs[16 * j + lane] = x;
        }
        cf_stores_done();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            if (!alive) {
                D.st->chol_ok = 0;
                D.st->solverTimeout = 1;
            }
            __hip_atomic_store(F.flagX + j, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CF_PROBE
            if ((j == NB - 1 || j == NB / 2 || j == 0) && D.st->nIterTotal == 2) printf("cholflow x_%d published at %llu (x 10 ns)\n", j, wall_clock64() - cfT0);
#endif
        }
    }
}
