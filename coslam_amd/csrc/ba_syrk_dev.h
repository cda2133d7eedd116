// ba_syrk_dev.h -- the reduced camera system of a LARGE bundle adjustment as a symmetric rank-k update on the f64 matrix
// cores (included by ba.hip inside its anonymous namespace; uses BaDev and the helpers defined there).
//
// S = U + lambda I - sum_p W_p V_p^-1 W_p^T.  With V_p^-1 = L_p L_p^T (3x3 Cholesky) and Z_p = W_p L_p the sum is Z Z^T for the
// (6 nc) x (3 P) matrix Z = [Z_1 ... Z_P]: a dense contraction when most cameras see most points -- BASELINE cfg5, 112 free
// poses x 5000 points x 600 k measurements: 672^2 x 15000 = 6.8 GFLOP per LM step.  The pair-per-workgroup kernel (k_schur)
// walks that as 6216 camera pairs x 5000 points with 45 doubles of operands per 324 flops: bound by L2 bandwidth, 2.34 ms per
// step, 6.6 TFLOP/s.  Here:
//   k_syrk_pack    one wave per point: L_p, then per measurement Z_ap = W_ap L_p into Zt (K-major: row 3p + c, column
//                  6 a + r -- six contiguous doubles per store) and t_o = W_ap V_p^-1 g_p (the measurement's share of the
//                  right-hand side).  Entries of cameras that do not see the point stay zero (Zt is cleared once per solve;
//                  every present entry is rewritten by every step).
//   k_syrk_mfma    C = Z Z^T, lower 128x128 tiles x K slices, one workgroup each: panels of 16 K-rows staged through LDS
//                  (double buffered, row stride 144 doubles so the four 16-lane groups of a ds_read_b64 hit disjoint banks),
//                  4 waves x (4x4 tiles of v_mfma_f64_16x16x4f64) = 64 accumulators per lane.  Operands come straight out of
//                  the K-major panel: lane l needs row k + l/16, column i0 + l%16 for A and for B alike.
//   k_schur_diag_u one workgroup per free camera: U_a, g_a over its measurement list (k_schur's diagonal branch) + sum t_o.
//   k_syrk_reduce  adds the K slices in slice order (deterministic), forms S (both triangles) and rhs.
// Selected for orders above SB_MAX_ORDER when the pair lists are too large to build (COSLAM_BA_SYRK=0 keeps k_schur).

constexpr int SY_TB = 128;    // output tile
constexpr int SY_KC = 16;     // K rows per LDS panel
constexpr int SY_US = 8;      // slices of a camera's measurement list in k_schur_diag_u
constexpr int SY_LDP = 144;   // panel row stride in doubles: = 16 mod 32 -> conflict-free ds_read_b64 for the MFMA operands

typedef double sy_v4 __attribute__((ext_vector_type(4)));

struct SyrkDev {
    double* Zt;      // [Kpad][ldz]
    double* Tobs;    // [nObs][6]
    double* Cpart;   // [nSlices][nTiles][128 * 128]
    double* Udiag;   // [nc][SY_US][33]: U upper (21), g (6), sum t (6), in SY_US slices of the camera's measurement list
    int ldz, Kpad, Kslice, nSlices, nT, nTiles;
};

// one wave per point
__global__ __launch_bounds__(256) void k_syrk_pack(BaDev D, SyrkDev Y) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= D.P || i < D.pLo || i >= D.pHi) return;
    double Vi[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Vi[q] = D.Vinv[9 * (size_t)i + q];
    const double g0 = D.gp[3 * (size_t)i], g1 = D.gp[3 * (size_t)i + 1], g2 = D.gp[3 * (size_t)i + 2];
    // L L^T = V^-1 (lower).  V^-1 is symmetric positive definite or zero (held point); a pivot that rounding made
    // non-positive drops its column.
    double L00 = 0, L10 = 0, L20 = 0, L11 = 0, L21 = 0, L22 = 0;
    if (Vi[0] > 0) {
        L00 = sqrt(Vi[0]);
        L10 = Vi[3] / L00;
        L20 = Vi[6] / L00;
    }
    const double d1 = Vi[4] - L10 * L10;
    if (d1 > 0) {
        L11 = sqrt(d1);
        L21 = (Vi[7] - L20 * L10) / L11;
    }
    const double d2 = Vi[8] - L20 * L20 - L21 * L21;
    if (d2 > 0) L22 = sqrt(d2);
    const double vg0 = Vi[0] * g0 + Vi[1] * g1 + Vi[2] * g2, vg1 = Vi[3] * g0 + Vi[4] * g1 + Vi[5] * g2,
                 vg2 = Vi[6] * g0 + Vi[7] * g1 + Vi[8] * g2;
    const int o0 = D.obs_ptr[i], o1 = D.obs_ptr[i + 1];
    double* z0 = Y.Zt + (size_t)(3 * i) * Y.ldz;
    for (int o = o0 + lane; o < o1; o += 64) {
        const double* Wo = D.W + 18 * (size_t)o;
        const int ja = D.obs_cam[o] - D.nCamsCon;
        double w[18];
#pragma unroll
        for (int q = 0; q < 18; ++q) w[q] = Wo[q];
        double* To = Y.Tobs + 6 * (size_t)o;
#pragma unroll
        for (int r = 0; r < 6; ++r) To[r] = w[3 * r] * vg0 + w[3 * r + 1] * vg1 + w[3 * r + 2] * vg2;
        if (ja < 0) continue;  // fixed camera: W = 0, no column in Z
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            z0[6 * ja + r] = w[3 * r] * L00 + w[3 * r + 1] * L10 + w[3 * r + 2] * L20;
            z0[(size_t)Y.ldz + 6 * ja + r] = w[3 * r + 1] * L11 + w[3 * r + 2] * L21;
            z0[2 * (size_t)Y.ldz + 6 * ja + r] = w[3 * r + 2] * L22;
        }
    }
}

__device__ __forceinline__ void sy_tile_of(int t, int& I, int& J) {  // lower triangle, row by row: t = I (I + 1) / 2 + J
    int r = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    I = r;
    J = t - r * (r + 1) / 2;
}

__global__ __launch_bounds__(256) void k_syrk_mfma(BaDev D, SyrkDev Y) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    extern __shared__ __attribute__((aligned(16))) double sy_lds[];  // [2 buffers][A | B][SY_KC][SY_LDP]
    const int tile = blockIdx.x % Y.nTiles, slice = blockIdx.x / Y.nTiles;
    int I, J;
    sy_tile_of(tile, I, J);
    const bool diagTile = (I == J);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = (wv & 1) * 64, wc = (wv >> 1) * 64;  // this wave's 64 x 64 corner of the tile
    const int kBeg = slice * Y.Kslice, nChunks = Y.Kslice / SY_KC;
    const double* gA = Y.Zt + (size_t)kBeg * Y.ldz + (size_t)I * SY_TB;
    const double* gB = Y.Zt + (size_t)kBeg * Y.ldz + (size_t)J * SY_TB;
    constexpr int PANEL = SY_KC * SY_LDP;
    // this thread's four 16-byte pieces of a 16 x 128 panel: rows tid / 64 + 4 u, doubles 2 (tid % 64)
    const int pr0 = tid >> 6, pc = (tid & 63) * 2;
    const size_t gRow = (size_t)pr0 * Y.ldz + pc, gStep = 4 * (size_t)Y.ldz;  // + chunk * SY_KC * ldz
    const int lRow = pr0 * SY_LDP + pc;                                      // + u * 4 * SY_LDP
    double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define SY_GLOAD(chunk)                                                        \
    {                                                                          \
        const size_t o_ = (size_t)(chunk)*SY_KC * Y.ldz + gRow;                \
        ra0 = *(const double2*)(gA + o_);                                      \
        ra1 = *(const double2*)(gA + o_ + gStep);                              \
        ra2 = *(const double2*)(gA + o_ + 2 * gStep);                          \
        ra3 = *(const double2*)(gA + o_ + 3 * gStep);                          \
        if (!diagTile) {                                                       \
            rb0 = *(const double2*)(gB + o_);                                  \
            rb1 = *(const double2*)(gB + o_ + gStep);                          \
            rb2 = *(const double2*)(gB + o_ + 2 * gStep);                      \
            rb3 = *(const double2*)(gB + o_ + 3 * gStep);                      \
        }                                                                      \
    }
#define SY_LSTORE(buf)                                                         \
    {                                                                          \
        double* a_ = sy_lds + (size_t)(buf)*2 * PANEL + lRow;                  \
        *(double2*)(a_) = ra0;                                                 \
        *(double2*)(a_ + 4 * SY_LDP) = ra1;                                    \
        *(double2*)(a_ + 8 * SY_LDP) = ra2;                                    \
        *(double2*)(a_ + 12 * SY_LDP) = ra3;                                   \
        if (!diagTile) {                                                       \
            *(double2*)(a_ + PANEL) = rb0;                                     \
            *(double2*)(a_ + PANEL + 4 * SY_LDP) = rb1;                        \
            *(double2*)(a_ + PANEL + 8 * SY_LDP) = rb2;                        \
            *(double2*)(a_ + PANEL + 12 * SY_LDP) = rb3;                       \
        }                                                                      \
    }
    rb0 = rb1 = rb2 = rb3 = make_double2(0.0, 0.0);
    sy_v4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (sy_v4){0.0, 0.0, 0.0, 0.0};
    SY_GLOAD(0);
    SY_LSTORE(0);
    __syncthreads();
    const int kq = lane >> 4, ln = lane & 15;
    for (int ch = 0; ch < nChunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nChunks) SY_GLOAD(ch + 1);  // in flight while this panel is multiplied
        const double* a = sy_lds + (size_t)buf * 2 * PANEL;
        const double* b = diagTile ? a : a + PANEL;
#pragma unroll
        for (int kk = 0; kk < SY_KC; kk += 4) {
            double fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = a[(kk + kq) * SY_LDP + wr + 16 * t + ln];
                fb[t] = b[(kk + kq) * SY_LDP + wc + 16 * t + ln];
            }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ta], fb[tb], acc[ta][tb], 0, 0, 0);
        }
        if (ch + 1 < nChunks) {
            SY_LSTORE(buf ^ 1);  // the other buffer: its readers finished before the barrier that ended the previous chunk
            __syncthreads();
        }
    }
#undef SY_GLOAD
#undef SY_LSTORE
    // partial tile of this K slice: D lane l, register q = row l/16 + 4q, column l%16 of each 16 x 16 block
    double* out = Y.Cpart + ((size_t)slice * Y.nTiles + tile) * (SY_TB * SY_TB);
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int q = 0; q < 4; ++q) out[(size_t)(wr + 16 * ta + kq + 4 * q) * SY_TB + wc + 16 * tb + ln] = acc[ta][tb][q];
}

// U_a, g_a and the camera's share of the right-hand side: one workgroup per free camera over its own measurement list
__global__ __launch_bounds__(256) void k_schur_diag_u(BaDev D, SyrkDev Y) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    __shared__ double red[4][33];
    const int ja = blockIdx.x / SY_US, sl = blockIdx.x % SY_US, ca = ja + D.nCamsCon;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double u[33];
#pragma unroll
    for (int q = 0; q < 33; ++q) u[q] = 0;
    const int s0 = D.cam_ptr[ca], s1 = D.cam_ptr[ca + 1], len = (s1 - s0 + SY_US - 1) / SY_US;
    const int sBeg = s0 + sl * len, sEnd = min(s1, sBeg + len);
    for (int s = sBeg + (int)threadIdx.x; s < sEnd; s += 256) {
        const int oa = D.cam_obs[s];
        const int ip = D.obs_pt[oa];
        if (ip < D.pLo || ip >= D.pHi) continue;  // another rank's point
        const double* T = Y.Tobs + 6 * (size_t)oa;
#pragma unroll
        for (int r = 0; r < 6; ++r) u[27 + r] += T[r];  // zero for outliers / held points (W = 0)
        if (D.outlier[oa]) continue;
        const double* Jm = D.Jc + 12 * (size_t)oa;
        const double e0 = D.e[2 * (size_t)oa], e1 = D.e[2 * (size_t)oa + 1];
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) u[q++] += Jm[r] * Jm[c] + Jm[6 + r] * Jm[6 + c];
#pragma unroll
        for (int r = 0; r < 6; ++r) u[21 + r] += Jm[r] * e0 + Jm[6 + r] * e1;
    }
    cs_reduce_many<33>(u, lane);
    const int q = cs_reduce_index<33>(lane);
    if (q >= 0) red[wv][q] = u[0];
    __syncthreads();
    if (threadIdx.x < 33)
        Y.Udiag[33 * ((size_t)ja * SY_US + sl) + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__device__ __forceinline__ double sy_udiag(const SyrkDev& Y, int ja, int q) {  // slices added in slice order
    const double* p = Y.Udiag + 33 * (size_t)ja * SY_US + q;
    double v = 0;
#pragma unroll
    for (int k = 0; k < SY_US; ++k) v += p[33 * k];
    return v;
}

// S = U + lambda I - sum over the K slices (slice order), both triangles; rhs = g - sum t
__global__ __launch_bounds__(256) void k_syrk_reduce(BaDev D, SyrkDev Y) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    const int n = D.n;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < (size_t)n) {
        const int ja = (int)idx / 6, r = (int)idx % 6;
        D.rhs[idx] = sy_udiag(Y, ja, 21 + r) - sy_udiag(Y, ja, 27 + r);
    }
    if (idx >= (size_t)n * n) return;
    const int r = (int)(idx / n), c = (int)(idx % n);
    if (c > r) return;
    const int I = r / SY_TB, J = c / SY_TB, tile = I * (I + 1) / 2 + J;
    const double* p = Y.Cpart + (size_t)tile * (SY_TB * SY_TB) + (size_t)(r % SY_TB) * SY_TB + (c % SY_TB);
    const size_t sliceStride = (size_t)Y.nTiles * (SY_TB * SY_TB);
    double s = 0;
    for (int k = 0; k < Y.nSlices; ++k) s += p[k * sliceStride];
    double v = -s;
    if (r / 6 == c / 6) {
        const int ja = r / 6, rr = c % 6, cc = r % 6;  // rr <= cc inside the block (c <= r)
        const int uq = rr * 6 - (rr * (rr - 1)) / 2 + (cc - rr);
        v = (sy_udiag(Y, ja, uq) + ((r == c && D.addLambda) ? D.st->lambda : 0.0)) - s;
    }
    D.S[(size_t)r * n + c] = v;
    D.S[(size_t)c * n + r] = v;
}
