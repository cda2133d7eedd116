// ba_packed_dev.h -- the LM step of a SMALL reduced system (order 37..176: the 8-camera rig's joint local BA and its inter-camera
// solve) reshaped so that it fits a FEW compute units (included by ba.hip inside its anonymous namespace).
//
// Why.  The wave-per-point / workgroup-per-camera-pair kernels above need ~300-360 workgroups resident at once: alone on the
// chip an LM step is 68 us, but next to the persistent tracker (every SIMD holds two of its waves) the same kernels take
// 100-115 us per step and the 20-step joint BA is the frame loop's critical path (profiles/r03_*).  These functions do the same
// arithmetic with a quarter of the waves, as building blocks of two schedules: one launch per phase (the k_*_packed kernels
// below) and ONE launch for the whole LM loop (ba_persist_dev.h):
//   lin_wave        lane = measurement, a wave holds WHOLE points back to back (a plan made at upload: waveStart[]); the
//                   per-point sums (V, g, the inlier count) are taken over the point's lanes through the wave's LDS scratch in
//                   measurement order -- the serial order of the oracle's loop, and the same bits in every lane of the point,
//                   so every lane inverts V itself and no broadcast follows.  Also writes Y = W V^-1, which the Schur step
//                   would otherwise recompute per pair entry.
//   schur_team      WPP waves per camera pair over the pair's list {oa, ob, point}: 42 (+27 on the diagonal) sums folded with
//                   the transposed butterfly inside each wave, the waves' totals added in wave order.
//   update_wave     tentative step and its cost, same lane layout as lin_wave.
// In the launch-per-phase schedule the LM control of the PREVIOUS step (accept / reject, lambda, stop tests) is the head of
// k_lin_packed: every workgroup sums the partial costs itself (fixed order) and picks the estimate to linearise at -- there is
// no control launch and no commit copy: "current" and "tentative" are two buffers and an index; k_control_final settles the
// decision that is still pending when the inner loop ends and puts the estimate back into Rs / Ts / pts.
// LM state hand-over without a race: k_lin_packed reads state A and (workgroup 0) writes the decided state to B; the Schur
// kernel, the solver and k_update_packed run on B; k_update_packed's workgroup 0 copies B back to A with `pending` set.  Nobody
// reads a state word that another workgroup of the same launch writes.
// COH (template flag of every function here): plain loads / stores when a kernel boundary separates producer and consumer,
// relaxed agent-scope atomics when both run inside one launch.

__device__ __forceinline__ int lp_ld_i(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lp_st_i(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct LmView {
    int active, cur;
    double lambda;
};

// the LM rules of k_control_step as a pure function of (state, cost of the tentative step, squared step)
struct LmRule {
    int acc, done;
    double lambda, cost;
};
__device__ __forceinline__ LmRule lm_rule(int chol_ok, double cost_old, double lambda, int inner_it, int innerMaxIter, double cost_sum,
                                          double step2) {
    LmRule r;
    const double cost_new = chol_ok ? cost_sum : 1e300;
    r.acc = (chol_ok && cost_new <= cost_old) ? 1 : 0;
    r.done = 0;
    r.lambda = lambda;
    r.cost = cost_old;
    if (r.acc) {
        const double dec = cost_old - cost_new;
        r.cost = cost_new;
        r.lambda = lambda / 10;
        if (dec < 1e-9 * cost_new + 1e-15 || step2 < 1e-20) r.done = 1;
    } else {
        r.lambda = lambda * 10;
        if (lambda * 10 > 1e12) r.done = 1;
    }
    if (inner_it + 1 >= innerMaxIter) r.done = 1;
    return r;
}

// LM control of the previous step, by every thread of a 256-thread workgroup (all arrive at the same answer); workgroup 0 records
// it in *D.stn.
// takesStep: the caller goes on to take an LM step when the answer is "active" (k_lin_packed: yes; k_control_final: no)
__device__ __forceinline__ LmView lm_head(const BaDev& D, double* red /* shared, 8 doubles */, const bool takesStep) {
    const int tid = threadIdx.x;
    const BaState* st = D.st;
    const int all_done = st->all_done, inner_done = st->inner_done, pending = st->pending, chol_ok = st->chol_ok,
              inner_it = st->inner_it, cur = st->cur;
    const double cost_old = st->cost, lambda = st->lambda;
    LmView v;
    v.active = 0;
    v.cur = cur;
    v.lambda = lambda;
    if (all_done || inner_done) return v;
    if (!pending) {  // first step of an LM run: nothing to decide
        if (blockIdx.x == 0 && tid == 0) {
            BaState s = *st;
            if (takesStep) s.seq += 1;  // (one more LM step is being taken)
            *D.stn = s;
        }
        v.active = 1;
        return v;
    }
    double c = 0, s2 = 0;
    for (int q = tid; q < D.nUpdBlocks; q += 256) {
        c += D.costPart[q];
        s2 += D.stepPart[q];
    }
    c = wsum(c);
    s2 = wsum(s2);
    if ((tid & 63) == 0) {
        red[tid >> 6] = c;
        red[4 + (tid >> 6)] = s2;
    }
    __syncthreads();
    const double cost_sum = ((red[0] + red[1]) + red[2]) + red[3];
    const double step2 = ((red[4] + red[5]) + red[6]) + red[7];
    const LmRule r = lm_rule(chol_ok, cost_old, lambda, inner_it, D.innerMaxIter, cost_sum, step2);
    if (blockIdx.x == 0 && tid == 0) {
        BaState s = *st;
        s.nIterTotal += 1;
        s.inner_it = inner_it + 1;
        if (!chol_ok) s.nCholFail += 1;
        if (r.acc) s.nAccepted += 1;
        s.cost = r.cost;
        s.lambda = r.lambda;
        s.inner_done = r.done;
        s.pending = 0;
        s.cur = r.acc ? (cur ^ 1) : cur;
        if (!r.done && takesStep) s.seq += 1;
        *D.stn = s;
    }
    v.active = r.done ? 0 : 1;
    v.cur = r.acc ? (cur ^ 1) : cur;
    v.lambda = r.lambda;
    return v;
}

// every lane of a point gets the sum of v[k] over the point's lanes [segStart, segStart + segLen) of its wave, added in lane
// order from zero (a serial loop over the point's measurements); wl = this wave's K x 64 doubles of LDS
template <int K>
__device__ __forceinline__ void seg_allreduce(double (&v)[K], double* wl, int lane, int segStart, int segLen) {
#pragma unroll
    for (int k = 0; k < K; ++k) wl[k * 64 + lane] = v[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    for (int q = 0; __builtin_amdgcn_ballot_w64(q < segLen) != 0ull; ++q) {
        if (q < segLen) {
            const double* p = wl + segStart + q;
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] += p[k * 64];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = s[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the scratch may be rewritten by the caller's next reduction)
}

struct PackLane {
    int o, i, j, segStart, segLen;
    bool has, in;
};
__device__ __forceinline__ PackLane pack_lane(const BaDev& D, int w, int lane) {
    PackLane L;
    const int o0 = D.waveStart[w], o1 = D.waveStart[w + 1];
    L.o = o0 + lane;
    L.has = L.o < o1;
    L.i = L.j = L.segStart = L.segLen = 0;
    L.in = false;
    if (L.has) {
        L.i = D.obs_pt[L.o];
        L.j = D.obs_cam[L.o];
        L.in = !D.outlier[L.o];
        const int p0 = D.obs_ptr[L.i];
        L.segStart = p0 - o0;
        L.segLen = D.obs_ptr[L.i + 1] - p0;
    }
    return L;
}

// linearisation of the measurements of wave w at the estimate `cur` with damping `lambda`; wl: 10 x 64 doubles of LDS
// COH_CAM: the camera poses were written by ANOTHER workgroup of this launch (fused update + linearisation): coherent loads
template <bool COH, bool COH_CAM = COH>
__device__ __forceinline__ void lin_wave(const BaDev& D, int w, int lane, int cur, double lambda, double* wl) {
    const PackLane L = pack_lane(D, w, lane);
    const double* Rc = cur ? D.Rn : D.Rs;
    const double* Tc = cur ? D.Tn : D.Ts;
    const double* Mc = cur ? D.Mn : D.pts;
    double M[3] = {0, 0, 0}, R[9], T[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = 0;
    T[0] = T[1] = T[2] = 0;
    if (L.has) {
#pragma unroll
        for (int q = 0; q < 3; ++q) M[q] = ldm<COH>(Mc + 3 * (size_t)L.i + q);
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = ldm<COH_CAM>(Rc + 9 * L.j + q);
#pragma unroll
        for (int q = 0; q < 3; ++q) T[q] = ldm<COH_CAM>(Tc + 3 * L.j + q);
    }
    double e[2] = {0, 0}, Jc[12], Jp[6];
#pragma unroll
    for (int q = 0; q < 12; ++q) Jc[q] = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) Jp[q] = 0;
    if (L.in) residual<true>(D.Ks + 9 * L.j, R, T, M, D.obs_xy + 2 * (size_t)L.o, e, Jc, Jp);
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // V upper (6) + g (3) + inlier count
    if (L.in && L.i >= D.nPtsCon) {
        acc[0] = Jp[0] * Jp[0] + Jp[3] * Jp[3];
        acc[1] = Jp[0] * Jp[1] + Jp[3] * Jp[4];
        acc[2] = Jp[0] * Jp[2] + Jp[3] * Jp[5];
        acc[3] = Jp[1] * Jp[1] + Jp[4] * Jp[4];
        acc[4] = Jp[1] * Jp[2] + Jp[4] * Jp[5];
        acc[5] = Jp[2] * Jp[2] + Jp[5] * Jp[5];
        acc[6] = Jp[0] * e[0] + Jp[3] * e[1];
        acc[7] = Jp[1] * e[0] + Jp[4] * e[1];
        acc[8] = Jp[2] * e[0] + Jp[5] * e[1];
    }
    acc[9] = L.in ? 1.0 : 0.0;
    seg_allreduce<10>(acc, wl, lane, L.segStart, L.segLen);
    // a point seen by fewer than two inlier measurements has no depth constraint: hold it (DESIGN.md "Robust BA")
    const bool freeP = L.has && (L.i >= D.nPtsCon) && (acc[9] >= 2.0);
    double Vi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    if (freeP) {
        double Vm[9] = {acc[0] + lambda, acc[1], acc[2], acc[1], acc[3] + lambda, acc[4], acc[2], acc[4], acc[5] + lambda};
        if (!inv33(Vm, Vi)) {
#pragma unroll
            for (int q = 0; q < 9; ++q) Vi[q] = 0;
        }
        g[0] = acc[6];
        g[1] = acc[7];
        g[2] = acc[8];
    }
    if (!L.has) return;
    const bool wf = L.in && freeP && (L.j >= D.nCamsCon);
    double Wm[18], Y[18];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Wm[3 * r + c] = wf ? (Jc[r] * Jp[c] + Jc[6 + r] * Jp[3 + c]) : 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Y[3 * r + c] = Wm[3 * r] * Vi[c] + Wm[3 * r + 1] * Vi[3 + c] + Wm[3 * r + 2] * Vi[6 + c];
    double* Jo = D.Jc + 12 * (size_t)L.o;
    double* Wo = D.W + 18 * (size_t)L.o;
    double* Yo = D.Y + 18 * (size_t)L.o;
#pragma unroll
    for (int q = 0; q < 12; ++q) stm<COH>(Jo + q, Jc[q]);
    stm<COH>(D.e + 2 * (size_t)L.o, e[0]);
    stm<COH>(D.e + 2 * (size_t)L.o + 1, e[1]);
#pragma unroll
    for (int q = 0; q < 18; ++q) stm<COH>(Wo + q, Wm[q]);
#pragma unroll
    for (int q = 0; q < 18; ++q) stm<COH>(Yo + q, Y[q]);
    if (lane == L.segStart) {
#pragma unroll
        for (int q = 0; q < 9; ++q) stm<COH>(D.Vinv + 9 * (size_t)L.i + q, Vi[q]);
#pragma unroll
        for (int q = 0; q < 3; ++q) stm<COH>(D.gp + 3 * (size_t)L.i + q, g[q]);
    }
}

__device__ __forceinline__ void schur_pair_of(const BaDev& D, int pi, int& ja, int& jb) {  // the diagonal pairs first
    if (pi < D.nc) {
        ja = jb = pi;
    } else {
        int pair = pi - D.nc;
        ja = 0;
        while (pair >= D.nc - 1 - ja) {
            pair -= D.nc - 1 - ja;
            ++ja;
        }
        jb = ja + 1 + pair;
    }
}

// One wave's share (sub of WPP) of camera pair pi: its entries' sums, folded inside the wave, into tt[72] (42 Schur sums, then on
// the diagonal the 27 sums of U_j, g_j).  The caller synchronises the team and calls schur_finish on its first wave.
template <bool COH, int WPP>
__device__ __forceinline__ void schur_pair_part(const BaDev& D, int pi, int sub, int lane, double* tt) {
    int ja, jb;
    schur_pair_of(D, pi, ja, jb);
    const int ca = ja + D.nCamsCon, cb = jb + D.nCamsCon;
    const bool diag = (ja == jb);
    const size_t pid = (size_t)ca * D.C - (size_t)ca * (ca - 1) / 2 + (size_t)(cb - ca);
    const int eBeg = D.pairPtr[pid], eEnd = D.pairPtr[pid + 1];
    double acc[42];
#pragma unroll
    for (int q = 0; q < 42; ++q) acc[q] = 0;
    for (int en = eBeg + sub * 64 + lane; en < eEnd; en += 64 * WPP) {
        const int4 E = D.pairEnt[en];
        const double* Ya = D.Y + 18 * (size_t)E.x;
        const double* Wb = D.W + 18 * (size_t)E.y;
        double Y[18], Wv[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) Y[q] = ldm<COH>(Ya + q);
#pragma unroll
        for (int q = 0; q < 18; ++q) Wv[q] = ldm<COH>(Wb + q);
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                acc[6 * r + c] += Y[3 * r] * Wv[3 * c] + Y[3 * r + 1] * Wv[3 * c + 1] + Y[3 * r + 2] * Wv[3 * c + 2];
        if (diag) {
            const double* g = D.gp + 3 * (size_t)E.z;
            const double g0 = ldm<COH>(g), g1 = ldm<COH>(g + 1), g2 = ldm<COH>(g + 2);
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[36 + r] += Y[3 * r] * g0 + Y[3 * r + 1] * g1 + Y[3 * r + 2] * g2;
        }
    }
    {
        cs_reduce_many<42>(acc, lane);
        const int q = cs_reduce_index<42>(lane);
        if (q >= 0) tt[q] = acc[0];
    }
    if (diag) {  // U_j, g_j: every measurement of the camera (outliers carry Jc = e = 0), fixed points included
        double u[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) u[q] = 0;
        for (int en = eBeg + sub * 64 + lane; en < eEnd; en += 64 * WPP) {
            const int oa = D.pairEnt[en].x;
            const double* Jg = D.Jc + 12 * (size_t)oa;
            double J[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) J[q] = ldm<COH>(Jg + q);
            const double e0 = ldm<COH>(D.e + 2 * (size_t)oa), e1 = ldm<COH>(D.e + 2 * (size_t)oa + 1);
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) u[q++] += J[r] * J[c] + J[6 + r] * J[6 + c];
#pragma unroll
            for (int r = 0; r < 6; ++r) u[21 + r] += J[r] * e0 + J[6 + r] * e1;
        }
        cs_reduce_many<27>(u, lane);
        const int q = cs_reduce_index<27>(lane);
        if (q >= 0) tt[42 + q] = u[0];
    }
}
// the team's first wave: the WPP waves' totals (tt0[sub * 72 + q]) added in wave order, the pair's block of S || rhs written
template <bool COH, int WPP>
__device__ __forceinline__ void schur_finish(const BaDev& D, int pi, int lane, const double* tt0, double lambda) {
    if (lane >= 42) return;
    int ja, jb;
    schur_pair_of(D, pi, ja, jb);
    const bool diag = (ja == jb);
    const int q = lane, n = D.n;
    auto total = [&](int k) {
        double s = tt0[k];
#pragma unroll
        for (int u = 1; u < WPP; ++u) s += tt0[u * 72 + k];
        return s;
    };
    const double s = total(q);
    if (q < 36) {
        const int r = q / 6, c = q - 6 * r;
        if (diag) {
            const int rr = r < c ? r : c, cc = r < c ? c : r;
            const int uq = rr * 6 - (rr * (rr - 1)) / 2 + (cc - rr);
            stm<COH>(&D.S[(size_t)(6 * ja + r) * n + 6 * ja + c], (total(42 + uq) + ((r == c && D.addLambda) ? lambda : 0.0)) - s);
        } else {
            stm<COH>(&D.S[(size_t)(6 * ja + r) * n + 6 * jb + c], -s);
            stm<COH>(&D.S[(size_t)(6 * jb + c) * n + 6 * ja + r], -s);
        }
    } else if (diag) {
        stm<COH>(&D.rhs[6 * ja + (q - 36)], total(42 + 21 + (q - 36)) - s);
    }
}

// tentative step of the points of wave w and the tentative cost of its measurements (per lane, to be summed by the caller);
// rhs = the solved camera step; wl: 3 x 64 doubles of LDS
template <bool COH, bool COH_RHS = COH>
__device__ __forceinline__ void update_wave(const BaDev& D, int w, int lane, int cur, double* wl, double& cost, double& step) {
    const double* Rc = cur ? D.Rn : D.Rs;
    const double* Tc = cur ? D.Tn : D.Ts;
    const double* Mc = cur ? D.Mn : D.pts;
    double* Mt = cur ? D.pts : D.Mn;
    const double* rhs = D.rhs;
    const PackLane L = pack_lane(D, w, lane);
    double b[3] = {0, 0, 0};
    double dc[6] = {0, 0, 0, 0, 0, 0};
    const int jf = L.j - D.nCamsCon;
    if (L.has && jf >= 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) dc[q] = ldm<COH_RHS>(rhs + 6 * jf + q);
    }
    if (L.has && L.in && jf >= 0 && L.i >= D.nPtsCon) {
        const double* Wo = D.W + 18 * (size_t)L.o;
        double Wm[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) Wm[q] = ldm<COH>(Wo + q);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 6; ++r) b[c] -= Wm[3 * r + c] * dc[r];
    }
    seg_allreduce<3>(b, wl, lane, L.segStart, L.segLen);
    if (!L.has) return;
    double Mn[3], d[3] = {0, 0, 0};
    const double* Mp = Mc + 3 * (size_t)L.i;
    if (L.i >= D.nPtsCon) {
        const double* Vi = D.Vinv + 9 * (size_t)L.i;
        const double* gp = D.gp + 3 * (size_t)L.i;
        double V9[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) V9[q] = ldm<COH>(Vi + q);
        const double g0 = ldm<COH>(gp) + b[0], g1 = ldm<COH>(gp + 1) + b[1], g2 = ldm<COH>(gp + 2) + b[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) d[r] = V9[3 * r] * g0 + V9[3 * r + 1] * g1 + V9[3 * r + 2] * g2;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) Mn[r] = ldm<COH>(Mp + r) + d[r];
    if (lane == L.segStart) {
#pragma unroll
        for (int r = 0; r < 3; ++r) stm<COH>(Mt + 3 * (size_t)L.i + r, Mn[r]);
        step += d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    }
    if (L.in) {
        double Rn[9], Tn[3], Rcur[9];
        const double* Rj = Rc + 9 * L.j;
        const double* Tj = Tc + 3 * L.j;
#pragma unroll
        for (int q = 0; q < 9; ++q) Rcur[q] = ldm<COH>(Rj + q);
        if (jf >= 0) {
            double wv3[3] = {dc[0], dc[1], dc[2]}, dR[9];
            so3_exp(wv3, dR);
            mat33AB(Rcur, dR, Rn);
#pragma unroll
            for (int q = 0; q < 3; ++q) Tn[q] = ldm<COH>(Tj + q) + dc[3 + q];
        } else {
#pragma unroll
            for (int q = 0; q < 9; ++q) Rn[q] = Rcur[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) Tn[q] = ldm<COH>(Tj + q);
        }
        double e[2];
        residual<false>(D.Ks + 9 * L.j, Rn, Tn, Mn, D.obs_xy + 2 * (size_t)L.o, e, nullptr, nullptr);
        cost += e[0] * e[0] + e[1] * e[1];
    }
}

// tentative pose of camera j and its squared step
// COH_OUT: the tentative pose is read by OTHER workgroups of this launch (fused update + linearisation): coherent stores
template <bool COH, bool COH_RHS = COH, bool COH_OUT = COH>
__device__ __forceinline__ void update_cam(const BaDev& D, int j, int cur, double& step) {
    const double* Rc = cur ? D.Rn : D.Rs;
    const double* Tc = cur ? D.Tn : D.Ts;
    double* Rt = cur ? D.Rs : D.Rn;
    double* Tt = cur ? D.Ts : D.Tn;
    double Rcur[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Rcur[q] = ldm<COH>(Rc + 9 * j + q);
    if (j >= D.nCamsCon) {
        const double* dcj = D.rhs + 6 * (j - D.nCamsCon);
        double dc[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dc[q] = ldm<COH_RHS>(dcj + q);
        double wv3[3] = {dc[0], dc[1], dc[2]}, dR[9], Rn[9];
        so3_exp(wv3, dR);
        mat33AB(Rcur, dR, Rn);
#pragma unroll
        for (int q = 0; q < 9; ++q) stm<COH_OUT>(Rt + 9 * j + q, Rn[q]);
#pragma unroll
        for (int q = 0; q < 3; ++q) stm<COH_OUT>(Tt + 3 * j + q, ldm<COH>(Tc + 3 * j + q) + dc[3 + q]);
        double s2 = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) s2 += dc[q] * dc[q];
        step += s2;
    } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) stm<COH_OUT>(Rt + 9 * j + q, Rcur[q]);
#pragma unroll
        for (int q = 0; q < 3; ++q) stm<COH_OUT>(Tt + 3 * j + q, ldm<COH>(Tc + 3 * j + q));
    }
}

// ---- the launch-per-phase schedule ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lin_packed(BaDev D) {
    CS_BA_SETPRIO();
    __shared__ double red[8];
    __shared__ double segl[4][10 * 64];
    const LmView V = lm_head(D, red, true);
    if (!V.active) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = blockIdx.x * 4 + wv;
    if (w >= D.nPackWaves) return;
    lin_wave<false>(D, w, lane, V.cur, V.lambda, segl[wv]);
}

#ifndef CS_SCHUR_WPP
#define CS_SCHUR_WPP 4  // waves per camera pair (measured in the frame loop: 1 -> 1920, 2 -> 2150, 4 -> 2200 frames/s)
#endif
// CS_SCHUR_WAVES_PER_EU (A/B builds): compile the pair kernel for that many waves per SIMD (3: 168 VGPRs instead of 206 -- a wave
// then fits a SIMD that holds two tracker waves of 160)
#ifdef CS_SCHUR_WAVES_PER_EU
#define CS_SCHUR_ATTR __attribute__((amdgpu_waves_per_eu(CS_SCHUR_WAVES_PER_EU, CS_SCHUR_WAVES_PER_EU)))
#else
#define CS_SCHUR_ATTR
#endif
__global__ __launch_bounds__(256) CS_SCHUR_ATTR void k_schur_wave(BaDev D) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    constexpr int WPP = CS_SCHUR_WPP, TEAMS = 4 / WPP;
    __shared__ double tot[4][72];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, team = wv / WPP, sub = wv % WPP;
    const int pi = blockIdx.x * TEAMS + team, nPairs = D.nc * (D.nc + 1) / 2;
    if (pi < nPairs) schur_pair_part<false, WPP>(D, pi, sub, lane, tot[wv]);
    __syncthreads();
    if (pi < nPairs && sub == 0) schur_finish<false, WPP>(D, pi, lane, tot[team * WPP], D.st->lambda);
}

// tentative step + its cost; D.st = the state k_lin_packed decided (B), D.stn = the state the next k_lin_packed reads (A)
__global__ __launch_bounds__(256) void k_update_packed(BaDev D) {
    CS_BA_SETPRIO();
    __shared__ double red[8];
    __shared__ double segl[4][3 * 64];
    const BaState* st = D.st;
    const bool active = !st->all_done && !st->inner_done;
    const int cur = st->cur;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // (nobody reads *D.stn during this launch; chol_ok is the solver's, in *D.st)
        BaState s = *st;
        if (active) s.pending = 1;
        *D.stn = s;
    }
    if (!active) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = blockIdx.x * 4 + wv;
    double cost = 0, step = 0;
    if (w < D.nPackWaves) update_wave<false>(D, w, lane, cur, segl[wv], cost, step);  // (wave-uniform)
    {   // tentative camera poses: camera t of this launch's first ceil(C / 256) workgroups
        const int j = blockIdx.x * 256 + threadIdx.x;
        if (j < D.C) update_cam<false>(D, j, cur, step);
    }
    cost = wsum(cost);
    step = wsum(step);
    if (lane == 0) {
        red[wv] = cost;
        red[4 + wv] = step;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        D.costPart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
        D.stepPart[blockIdx.x] = ((red[4] + red[5]) + red[6]) + red[7];
    }
}

// end of a run of LM steps: the decision on the last tentative step (if one is pending), the estimate back in Rs / Ts / pts,
// and both state words equal (a speculative launch behind this one must see "done" in the word its solver and update read)
__global__ __launch_bounds__(256) void k_control_final(BaDev D) {
    __shared__ double red[8];
    if (D.st->all_done) return;
    BaDev D2 = D;
    D2.stn = D.st;  // in place: one workgroup, every thread has read the state before thread 0 rewrites it (barrier inside)
    const LmView V = lm_head(D2, red, false);
    __syncthreads();
    const int tid = threadIdx.x;
    if (V.cur == 1) {
        for (int q = tid; q < 9 * D.C; q += 256) D.Rs[q] = D.Rn[q];
        for (int q = tid; q < 3 * D.C; q += 256) D.Ts[q] = D.Tn[q];
        for (int q = tid; q < 3 * D.P; q += 256) D.pts[q] = D.Mn[q];
        __syncthreads();
    }
    if (tid == 0) {
        D.st->cur = 0;
        BaState s = *D.st;
        *D.stn = s;
        ba_publish_state(D, s.inner_done, s.all_done);
    }
}

// both estimates equal at the start of a solve: entries no step ever writes (points without measurements) must not differ
__global__ __launch_bounds__(256) void k_mirror_estimate(BaDev D) {
    const int t0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int q = t0; q < 9 * D.C; q += stride) D.Rn[q] = D.Rs[q];
    for (int q = t0; q < 3 * D.C; q += stride) D.Tn[q] = D.Ts[q];
    for (int q = t0; q < 3 * D.P; q += stride) D.Mn[q] = D.pts[q];
}
