// ba_persist_dev.h -- one run of LM steps (an inner loop of bundleAdjustRobust) as ONE cooperative launch on a handful of
// compute units (included by ba.hip inside its anonymous namespace, after ba_packed_dev.h).
//
// Why one launch.  Next to the persistent tracker every SIMD of the chip holds tracker waves; a chain of short dependent kernels
// queues behind them launch after launch (an LM step of the joint local BA: 68 us alone, 100-115 us in the frame loop, and 150 us
// when its stream is confined to 64 CUs by a CU mask -- masked queues dispatch slowly; profiles/r03_*).  A launch that STAYS
// resident owns its compute units for the whole run: G workgroups of 512 threads, each with more LDS than a tracker workgroup
// leaves free, so nothing else lands on their CUs; the tracker is budgeted for the other 256 - G (cs_klt_set_cu_count).  No CU
// mask is involved.
// Phases of a step, separated by grid barriers (one monotonic counter, relaxed agent-scope atomics, lane 0 of every workgroup):
//   L  lin_wave       all workgroups, waves strided over the lane plan
//   S  schur_team     all workgroups, teams of CS_SCHUR_WPP waves strided over the camera pairs
//   V  sb_solve_body  workgroup 0 (8 waves; the others wait in the barrier: one polling lane each)
//   U  update_wave    all workgroups; per-workgroup partial cost / squared step
//   D  the LM decision, by EVERY workgroup from the same partials in the same order -- the LM state is replicated in registers,
//      no state word is read or written inside the loop; workgroup 0 writes it back once, behind the last step, together with
//      the estimate (Rs / Ts / pts) if the current one sits in the other buffer.
// Data that crosses workgroups inside the launch (estimates, Jc / e / W / Y / V^-1 / g, S || rhs, the step, the partials) moves
// as relaxed agent-scope atomics (COH = true: sc1 -- never served from a stale L1 or another XCD's L2 line), every lane drains
// its stores (s_waitcnt vmcnt(0)) before its workgroup arrives at a barrier; the topology (index arrays, measurements, K, the
// outlier flags) is read-only during the launch and read plainly.
// Deadlock: the launch needs its G workgroups resident together.  It never waits for anything but itself, and whatever holds
// the CUs it is waiting for (tracker launches) terminates without it, so residency is reached; every poll is bounded all the
// same and a run that times out is reported (solverTimeout -> CS_ERR_NUMERIC), not hung.

constexpr int LP_NW = 8, LP_NT = LP_NW * 64;

struct LmPersist {
    int* bar;   // [0] barrier counter (zeroed before the launch), [1] chol_ok of the step, [2] time-out flag
    int maxSteps;
};


// all threads of the workgroup call; false: timed out (the grid is not resident / a workgroup died)
__device__ __forceinline__ bool lp_barrier(int* bar, int G, int& epoch, int* aliveSh) {
    cf_stores_done();  // this lane's stores have been acknowledged
    __syncthreads();
    epoch += 1;
    if (threadIdx.x == 0) {
        int ok = 1;
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = epoch * G;
        int spins = 0;
        while (lp_ld_i(bar) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) {
                ok = 0;
                break;
            }
            if ((spins & 1023) == 0 && lp_ld_i(bar + 2)) {  // somebody else gave up
                ok = 0;
                break;
            }
        }
        if (!ok) lp_st_i(bar + 2, 1);
        *aliveSh = ok;
    }
    __syncthreads();
    return *aliveSh != 0;
}

// COH = false: ONE workgroup (G = 1) -- nothing crosses workgroups, every access is a plain one and the "grid" barrier is the
// workgroup's own: the whole run of LM steps of a SMALL problem (the inter-camera solve: order 48, 2 k measurements) without a
// single launch boundary or coherent access.
template <bool COH>
__global__ __launch_bounds__(LP_NT) void k_lm_persist(BaDev D, LmPersist Q) {
    CS_BA_SETPRIO();
    extern __shared__ __attribute__((aligned(16))) double lp_sm[];  // the solver's blocks; the other phases' scratch aliases it
    __shared__ double red[2 * LP_NW];
    __shared__ int okFlag, aliveSh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = gridDim.x, gw0 = blockIdx.x * LP_NW + wv, nW = G * LP_NW;
    BaState* st = D.st;
    // the replicated LM state (identical in every workgroup: read from the same word, advanced by the same rule)
    if (st->all_done || st->inner_done) return;
    double lambda = st->lambda, cost = st->cost;
    int inner_it = st->inner_it, cur = st->cur;
    int nSteps = 0, nCholFail = 0, nAcc = 0, done = 0, timedOut = 0;
    int epoch = 0;
    const int nPairs = D.nc * (D.nc + 1) / 2;
    constexpr int WPP = CS_SCHUR_WPP, TEAMS = LP_NW / WPP;
    double* wl = lp_sm + (size_t)wv * 10 * 64;                     // L / U: this wave's segment scratch
    double* tot = lp_sm;                                           // S: [LP_NW][72]
    for (int it = 0; it < Q.maxSteps && !done; ++it) {
        // ---- L ----
        for (int w = gw0; w < D.nPackWaves; w += nW) lin_wave<COH>(D, w, lane, cur, lambda, wl);
        if (!lp_barrier(Q.bar, G, epoch, &aliveSh)) {
            timedOut = 1;
            break;
        }
        // ---- S ----
        for (int p0 = blockIdx.x * TEAMS; p0 < nPairs; p0 += G * TEAMS) {  // (workgroup-uniform trip count)
            const int team = wv / WPP, sub = wv % WPP, pi = p0 + team;
            if (pi < nPairs) schur_pair_part<COH, WPP>(D, pi, sub, lane, tot + (size_t)wv * 72);
            __syncthreads();
            if (pi < nPairs && sub == 0) schur_finish<COH, WPP>(D, pi, lane, tot + (size_t)team * WPP * 72, lambda);
            __syncthreads();
        }
        if (!lp_barrier(Q.bar, G, epoch, &aliveSh)) {
            timedOut = 1;
            break;
        }
        // ---- V ----
        if (blockIdx.x == 0) {
            if (D.n > 0) {
                sb_solve_body<LP_NW, COH>(D, lp_sm, &okFlag, false);
            } else if (tid == 0) {
                okFlag = 1;
            }
            cf_stores_done();
            __syncthreads();
            if (tid == 0) lp_st_i(Q.bar + 1, okFlag);
        }
        if (!lp_barrier(Q.bar, G, epoch, &aliveSh)) {
            timedOut = 1;
            break;
        }
        const int chol_ok = lp_ld_i(Q.bar + 1);
        // ---- U ----
        double c = 0, s2 = 0;
        for (int w = gw0; w < D.nPackWaves; w += nW) update_wave<COH>(D, w, lane, cur, wl, c, s2);
        for (int j = blockIdx.x * LP_NT + tid; j < D.C; j += G * LP_NT) update_cam<COH>(D, j, cur, s2);
        c = wsum(c);
        s2 = wsum(s2);
        if (lane == 0) {
            red[wv] = c;
            red[LP_NW + wv] = s2;
        }
        __syncthreads();
        if (tid == 0) {
            double a = red[0], b = red[LP_NW];
#pragma unroll
            for (int u = 1; u < LP_NW; ++u) {
                a += red[u];
                b += red[LP_NW + u];
            }
            stm<COH>(D.costPart + blockIdx.x, a);
            stm<COH>(D.stepPart + blockIdx.x, b);
        }
        if (!lp_barrier(Q.bar, G, epoch, &aliveSh)) {
            timedOut = 1;
            break;
        }
        // ---- D ---- (every workgroup, same partials, same order)
        double pc = 0, ps = 0;
        for (int q = tid; q < G; q += LP_NT) {
            pc += ldm<COH>(D.costPart + q);
            ps += ldm<COH>(D.stepPart + q);
        }
        pc = wsum(pc);
        ps = wsum(ps);
        if (lane == 0) {
            red[wv] = pc;
            red[LP_NW + wv] = ps;
        }
        __syncthreads();
        double cost_sum = red[0], step2 = red[LP_NW];
#pragma unroll
        for (int u = 1; u < LP_NW; ++u) {
            cost_sum += red[u];
            step2 += red[LP_NW + u];
        }
        __syncthreads();  // (red is rewritten by the next step)
        const LmRule r = lm_rule(chol_ok, cost, lambda, inner_it, D.innerMaxIter, cost_sum, step2);
        nSteps += 1;
        inner_it += 1;
        if (!chol_ok) nCholFail += 1;
        if (r.acc) {
            nAcc += 1;
            cur ^= 1;
        }
        cost = r.cost;
        lambda = r.lambda;
        done = r.done;
    }
    // ---- the state word and the estimate, once (the launch ends here: plain stores, flushed at the kernel boundary) ----
    if (blockIdx.x != 0) return;
    if (cur == 1 && !timedOut) {
        // (the other workgroups' tentative points were drained and their barrier passed before this workgroup got here)
        for (int q = tid; q < 9 * D.C; q += LP_NT) D.Rs[q] = ldm<COH>(D.Rn + q);
        for (int q = tid; q < 3 * D.C; q += LP_NT) D.Ts[q] = ldm<COH>(D.Tn + q);
        for (int q = tid; q < 3 * D.P; q += LP_NT) D.pts[q] = ldm<COH>(D.Mn + q);
    }
    if (tid == 0) {
        st->nIterTotal += nSteps;
        st->inner_it = inner_it;
        st->nCholFail += nCholFail;
        st->nAccepted += nAcc;
        st->cost = cost;
        st->lambda = lambda;
        st->inner_done = (done || timedOut) ? 1 : 0;
        st->pending = 0;
        st->cur = 0;
        if (timedOut) {
            st->solverTimeout = 1;
            st->all_done = 1;
        }
    }
}
