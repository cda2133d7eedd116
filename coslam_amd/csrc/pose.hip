// pose.hip -- intraCamEstimate on gfx950: Tukey-IRLS around Levenberg-Marquardt, binary64.
//
// Replaces bool intraCamEstimate(...) (src/slam/SL_IntraCamPose.h:92-95, SL_IntraCamPose.cpp:626-709) and
// the functions under it: intraCamWeightedLMProc (:475-549), intraCamWeightedLMStep (:259-303), the
// forward-difference Jacobians (:43-117, eps = 1e-8), getSO3ExpMap (:10-39), intraCamUpdatePose (:367-380).
//
// Design: the whole estimate -- up to 5 re-weighting rounds x up to 100 LM steps -- is ONE launch of ONE
// workgroup per camera (a batch of cameras = a grid of workgroups).  The reference is a serial loop
// over <= 192 points with 7 projections each; here lane i owns point i, the 21+6 entries of the
// weighted normal equations AND the weighted error of the same pose are folded together with one
// transposed 64-lane butterfly per wave and one LDS exchange between the waves (one barrier), and
// every lane then runs the 6x6 Gauss-Jordan and the LM accept/reject logic redundantly on identical
// inputs, so the control flow stays uniform with no host round trip per iteration.  An LM step is ONE
// pass over the points: the reference projects every point at a tentative pose for its error and, once
// it has accepted the pose, again for the next step's Jacobians -- the same numbers; here the pass at
// the tentative pose yields both (lm_pass), and a round's Tukey weights are computed inside its first pass.
// The arithmetic per point is the reference's, operation for operation (numeric Jacobians included):
// only the order of the sum over points differs from the serial CPU loop.
#include "cs_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int WS_LDS_MAX = 4096;  // points whose IRLS weights fit in LDS next to the reduction scratch
constexpr int SMALL_NPTS = 64;    // up to here ONE wave owns the problem (no LDS exchange, no barrier).  Beyond, four waves
                                  // with one point per lane win although they meet in LDS every LM step: an LM step is
                                  // seven projections per point, and three points per lane made it 73.6 vs 64.1 us at 192
                                  // points

__device__ __forceinline__ void so3_exp(const double w[3], double R[9]) {  // SL_IntraCamPose.cpp:10-39
    double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (theta == 0) {
        R[0] = 1, R[1] = 0, R[2] = 0, R[3] = 0, R[4] = 1, R[5] = 0, R[6] = 0, R[7] = 0, R[8] = 1;
        return;
    }
    double hw0 = w[0] / theta, hw1 = w[1] / theta, hw2 = w[2] / theta;
    double st, ct;
    if (theta < 0.5) {
        // an LM step's rotation is a fraction of a degree: sin and cos from their Taylor series (truncation below 2^-56 of the
        // value for theta < 0.5; the library's routines carry a range reduction for arguments this never sees)
        const double x2 = theta * theta;
        double ps = 1.0 / 355687428096000.0;   // 1 / 17!
        ps = ps * x2 - 1.0 / 1307674368000.0;
        ps = ps * x2 + 1.0 / 6227020800.0;
        ps = ps * x2 - 1.0 / 39916800.0;
        ps = ps * x2 + 1.0 / 362880.0;
        ps = ps * x2 - 1.0 / 5040.0;
        ps = ps * x2 + 1.0 / 120.0;
        ps = ps * x2 - 1.0 / 6.0;
        st = theta + theta * (x2 * ps);
        double pc = 1.0 / 20922789888000.0;    // 1 / 16!
        pc = pc * x2 - 1.0 / 87178291200.0;
        pc = pc * x2 + 1.0 / 479001600.0;
        pc = pc * x2 - 1.0 / 3628800.0;
        pc = pc * x2 + 1.0 / 40320.0;
        pc = pc * x2 - 1.0 / 720.0;
        pc = pc * x2 + 1.0 / 24.0;
        const double c = 1.0 - (0.5 * x2 - x2 * x2 * pc);   // cos(theta), rounded: the reference forms 1 - cos(theta) from that (:20)
        ct = 1 - c;
    } else {
        st = sin(theta);
        ct = 1 - cos(theta);
    }
    double hw0hw0 = hw0 * hw0, hw0hw1 = hw0 * hw1, hw0hw2 = hw0 * hw2;
    double hw1hw1 = hw1 * hw1, hw1hw2 = hw1 * hw2, hw2hw2 = hw2 * hw2;
    R[0] = -ct * hw1hw1 - ct * hw2hw2 + 1;
    R[1] = ct * hw0hw1 - st * hw2;
    R[2] = st * hw1 + ct * hw0hw2;
    R[3] = st * hw2 + ct * hw0hw1;
    R[4] = -ct * hw0hw0 - ct * hw2hw2 + 1;
    R[5] = ct * hw1hw2 - st * hw0;
    R[6] = ct * hw0hw2 - st * hw1;
    R[7] = st * hw0 + ct * hw1hw2;
    R[8] = -ct * hw0hw0 - ct * hw1hw1 + 1;
}

__device__ __forceinline__ void mat33AB(const double* A, const double* B, double* C) {
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) C[i] = T[i];
}

__device__ __forceinline__ void project(const double* K, const double* R, const double* t, const double* M, double* m) {
    double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
    double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
    double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
    double u = K[0] * X + K[1] * Y + K[2] * Z;
    double v = K[3] * X + K[4] * Y + K[5] * Z;
    double w = K[6] * X + K[7] * Y + K[8] * Z;
    m[0] = u / w;
    m[1] = v / w;
}

// (A + lambda I) p = B by Gauss-Jordan elimination with partial pivoting on [A | B] (the reference forms the LAPACK inverse and
// multiplies: same solution), ONE MATRIX ENTRY PER LANE: lane 7 r + j holds M[r][j] (r < 6, j < 7; column 6 = B).  Every lane of a
// wave running all 42 entries' arithmetic redundantly cost ~800 wave instructions per LM step, issued at 4 cycles each -- a third of
// the step; here a pivot step is one divide and one multiply-subtract across the lanes plus the exchanges that feed them (the pivot
// column through v_readlane, the row swap and the two factors through ds_bpermute).  Element for element the same operations as
// the register version: the pivot of column c is the first row >= c with the largest |M[r][c]|, the pivot row is divided by it,
// every other row loses f times the pivot row, columns <= c are left as they are.
__device__ __forceinline__ double cs_shfl_d(double v, int srcLane) {
    int lo = __shfl(__double2loint(v), srcLane, 64), hi = __shfl(__double2hiint(v), srcLane, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void solve66_lanes(double m, int lane, double* param) {
    const int r = lane / 7, j = lane - 7 * r;   // lanes >= 42 (r >= 6) carry nothing
    const bool in = r < 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        double best = fabs(cs_readlane_d(m, 7 * c + c));
        int p = c;
#pragma unroll
        for (int rr = c + 1; rr < 6; ++rr) {
            const double v = fabs(cs_readlane_d(m, 7 * rr + c));
            if (v > best) best = v, p = rr;
        }
        const int srcRow = r == c ? p : (r == p ? c : r);
        const double sw = cs_shfl_d(m, in ? 7 * srcRow + j : lane);
        m = sw;
        const double d = cs_readlane_d(m, 7 * c + c);
        const double q = m / d;
        if (r == c && j > c) m = q;
        const double f = cs_shfl_d(m, in ? 7 * r + c : lane);
        const double pr = cs_shfl_d(m, in ? 7 * c + j : lane);
        const double e = m - f * pr;
        if (in && r != c && j > c) m = e;
    }
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) param[rr] = cs_readlane_d(m, 7 * rr + 6);
}

__device__ __forceinline__ double tukey(double e, double tau) {  // :646-653
    if (e >= tau) return 0;
    e /= tau;
    e = 1 - e * e;
    return e * e;
}

struct PoseCtx {
    const double* K;
    const double* Ms;
    const double* ms;
    double* Ws;  // LDS or global scratch, npts
    int npts;
    double* red;          // LDS [PB/64][27]
    double pM[3], pm[2];  // the lane's FIRST point (i = threadIdx.x), read once instead of in every LM pass (170 points per camera: one
                          // point per lane; 2193-2200 against 2184-2186 frames/s in the loop, three alternating runs each)
    double dR[3][9];      // exp(eps e_k): independent of the iterate (SL_IntraCamPose.cpp:57-58)
    int myEntry;          // which of the 28 sums this lane holds in the lane-parallel solve
};

// The 27 sums of the weighted normal equations and the weighted squared error over the workgroup: the waves' transposed butterflies
// leave each total in ONE lane, which stores it; ONE barrier; then a lane fetches what IT needs -- the entry of [A | B] it holds in
// the lane-parallel solve (`mine`) and the error -- adding the waves' totals in wave order.  The scratch is double-buffered (`par`
// flips per call), so no second barrier protects it from the next call's stores.
constexpr int NSUM = 28;
// which of the 28 sums lane 7 r + j of the solve needs: A[r][j] = upper-triangle entry (min, max), B[r] = 21 + r
__device__ __forceinline__ int solve_entry_of_lane(int lane) {
    const int r = lane / 7, j = lane - 7 * r;
    if (r >= 6) return 27;
    if (j == 6) return 21 + r;
    const int a = r < j ? r : j, b = r < j ? j : r;
    return a * 6 - (a * (a - 1)) / 2 + (b - a);
}
template <int PB>
__device__ __forceinline__ void block_sum28(double (&v)[NSUM], double* lds /* [2][PB/64][NSUM] */, int& par, int myEntry, double& mine,
                                            double& err) {
    constexpr int NW = PB / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    cs_reduce_many<NSUM>(v, lane);
    if (NW == 1) {
        constexpr CsOwnerTable<NSUM> T{};
        int owner = 0;   // the lane that holds the total of entry myEntry
#pragma unroll
        for (int q = 0; q < NSUM; ++q) owner = myEntry == q ? T.lane[q] : owner;
        mine = cs_shfl_d(v[0], owner);
        err = cs_readlane_d(v[0], T.lane[27]);
        return;
    }
    const int have = cs_reduce_index<NSUM>(lane);
    double* buf = lds + par * (NW * NSUM);
    par ^= 1;
    if (have >= 0) buf[wv * NSUM + have] = v[0];
    __syncthreads();
    double s = buf[myEntry], e = buf[27];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += buf[w * NSUM + myEntry], e += buf[w * NSUM + 27];
    mine = s, err = e;
}

// ONE pass over the points at the pose (R, t): acc[0..20] the upper triangle of sum J^T J, acc[21..26] sum J^T r
// (intraCamWeightedLMStep, :259-303: forward-difference Jacobians, eps = 1e-8, the weight on J and on r) and acc[27] the weighted
// squared reprojection error (:439-456: the weight once).  The reference evaluates the error of a tentative pose and, when it
// accepts it, the Jacobians at that same pose in its next step: the same projections twice.  Here the tentative pose gets both in
// one pass -- accepted: the next step's normal equations are there; rejected: the previous ones are still in registers -- so an LM
// step is ONE pass and ONE reduction.  reweight: first the Tukey weights from the residuals at (R, t) (:687-701), which is where
// the reference computes them: at the pose the next round starts from.
template <int PB>
__device__ void lm_pass(const PoseCtx& c, const double* R, const double* t, bool reweight, double tau, int& par, double& mine, double& err) {
    const double eps = 1e-8;
    // The forward differences are SCALED by 1e8 where the reference divides by 1e-8 (:271-287): 12 of a point's 26 f64 divisions per pass
    // (each ~30 instructions with a quarter-rate reciprocal) become multiplications -- at most 1 ulp per Jacobian entry, the size of what
    // the tree sums below already differ from the reference's serial ones by; the tolerances of tests/test_pose_ba_gpu.py are unchanged.
    const double ieps = 1e8;
    // the perturbed rotations R * exp(eps e_k) do not depend on the point (:57-59)
    double R1[3][9];
#pragma unroll
    for (int a = 0; a < 3; ++a) mat33AB(R, c.dR[a], R1[a]);
    double acc[NSUM];
#pragma unroll
    for (int q = 0; q < NSUM; ++q) acc[q] = 0;
    for (int i = threadIdx.x; i < c.npts; i += PB) {
        double pM[3], pm[2];
        if (i < PB) {   // (uniform: the first trip of every lane)
            pM[0] = c.pM[0], pM[1] = c.pM[1], pM[2] = c.pM[2], pm[0] = c.pm[0], pm[1] = c.pm[1];
        } else {
            pM[0] = c.Ms[3 * i], pM[1] = c.Ms[3 * i + 1], pM[2] = c.Ms[3 * i + 2], pm[0] = c.ms[2 * i], pm[1] = c.ms[2 * i + 1];
        }
        double rm0[2], rm[2], J[12];
        project(c.K, R, t, pM, rm0);
        const double dx = rm0[0] - pm[0], dy = rm0[1] - pm[1];
        double w;
        if (reweight) {
            w = tukey(sqrt(dx * dx + dy * dy), tau);
            c.Ws[i] = w;   // (a point's weight is read by the lane that wrote it: no barrier)
        } else {
            w = c.Ws[i];
        }
        acc[27] += (dx * dx + dy * dy) * w;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            project(c.K, R1[a], t, pM, rm);
            J[a] = (rm[0] - rm0[0]) * ieps;
            J[6 + a] = (rm[1] - rm0[1]) * ieps;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double t1[3] = {t[0], t[1], t[2]};
            t1[a] = t[a] + eps;
            project(c.K, R, t1, pM, rm);
            J[3 + a] = (rm[0] - rm0[0]) * ieps;
            J[9 + a] = (rm[1] - rm0[1]) * ieps;
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) J[q] = w * J[q];
        const double r0 = (-rm0[0] + pm[0]) * w, r1 = (-rm0[1] + pm[1]) * w;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = r; cc < 6; ++cc) acc[q++] += J[r] * J[cc] + J[6 + r] * J[6 + cc];
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[21 + r] += J[r] * r0 + J[6 + r] * r1;
    }
    block_sum28<PB>(acc, c.red, par, c.myEntry, mine, err);
}

// (sum J^T J + lambda I) p = sum J^T r, the lane's entry of [A | B] in `mine`
__device__ __forceinline__ void lm_solve(double mine, double lambda, double* param) {
    const int lane = threadIdx.x & 63, r = lane / 7, j = lane - 7 * r;
    solve66_lanes(r == j ? mine + lambda : mine, lane, param);
}

__device__ __forceinline__ void update_pose(const double* R, const double* t, const double* p, double* Rn, double* tn) {
    double dR[9];
    so3_exp(p, dR);
    mat33AB(R, dR, Rn);
    tn[0] = t[0] + p[3];
    tn[1] = t[1] + p[4];
    tn[2] = t[2] + p[5];
}

template <int PB>
__device__ bool weighted_lm(const PoseCtx& c, const double* R0, const double* t0, double* R_opt, double* t_opt,
                            cs_pose_option& opt, bool reweight, double tau, int& par) {  // :475-549
    double param[6];
    double ne, cand, err;   // this lane's entry of the normal equations at the accepted pose / at the tentative one
    opt.npts = c.npts;
    opt.lambda = opt.lambda0;
    lm_pass<PB>(c, R0, t0, reweight, tau, par, ne, err);
    opt.err0 = err;
    opt.err = opt.err0;
    double R[9], t[3], R_tmp[9], t_tmp[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = R_tmp[i] = R0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = t_tmp[i] = t0[i];
    opt.retTypeLM = 1;
    int i = 0;
    for (; i < opt.maxIterLM; ++i) {
        lm_solve(ne, opt.lambda, param);
        update_pose(R, t, param, R_opt, t_opt);
        double p2 = param[0] * param[0] + param[1] * param[1] + param[2] * param[2] + param[3] * param[3] +
                    param[4] * param[4] + param[5] * param[5];
        if (p2 < opt.epsParamChangeLM) {
            opt.retTypeLM = 0;
            break;
        }
        lm_pass<PB>(c, R_opt, t_opt, false, tau, par, cand, err);
        if (fabs(err - opt.err) < opt.epsErrorChangeLM) {
            opt.retTypeLM = 0;
            break;
        }
        if (err <= opt.err) {
#pragma unroll
            for (int q = 0; q < 9; ++q) R[q] = R_tmp[q] = R_opt[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) t[q] = t_tmp[q] = t_opt[q];
            ne = cand;
            opt.err = err;
            opt.lambda /= 10;
        } else {
            opt.lambda *= 10;
            if (opt.lambda > 1e+18) {
                opt.retTypeLM = -1;
                break;
            }
        }
    }
    if (opt.retTypeLM == -1) {
#pragma unroll
        for (int q = 0; q < 9; ++q) R_opt[q] = R_tmp[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) t_opt[q] = t_tmp[q];
    }
    opt.err = err;
    opt.nIterLM = i;
    opt.verboseRW += i < opt.maxIterLM ? i + 1 : i;   // (diagnostic, see coslam_hip.h: LM steps taken over all rounds)
    return opt.retTypeLM >= 0;
}

// CS_IC_WAVES_PER_EU (A/B builds): compile for that many waves per SIMD (3: 168 VGPRs + 460 bytes of scratch per lane instead of 255
// VGPRs: a workgroup then fits a compute unit that holds two tracker waves per SIMD)
#if defined(CS_IC_NUM_VGPR)
#define CS_IC_ATTR __attribute__((amdgpu_num_vgpr(CS_IC_NUM_VGPR)))
#elif defined(CS_IC_WAVES_PER_EU)
#define CS_IC_ATTR __attribute__((amdgpu_waves_per_eu(CS_IC_WAVES_PER_EU, CS_IC_WAVES_PER_EU)))
#else
#define CS_IC_ATTR
#endif
template <int PB>
__global__ __launch_bounds__(PB) CS_IC_ATTR void k_intracam(int ptsStride, const double* __restrict__ Kall,
                                                 const double* __restrict__ R0all, const double* __restrict__ t0all,
                                                 const int* __restrict__ nptsAll, const double* __restrict__ prevErrs,
                                                 const double* __restrict__ MsAll, const double* __restrict__ msAll,
                                                 double tau, double* __restrict__ RoptAll, double* __restrict__ toptAll,
                                                 cs_pose_option* __restrict__ optAll, int* __restrict__ okAll,
                                                 double* __restrict__ wsScratch) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    CS_POSE_STREAM_PRIO();
    const int pb = blockIdx.x;
    constexpr int NW = PB / 64;
    double* red = smem;                  // [2][NW][NSUM]
    double* sK = smem + 2 * NW * NSUM;   // 9 (+3 pad)
    double* WsL = sK + 12;         // npts (when it fits)
    const int npts = nptsAll[pb];
    if (threadIdx.x < 9) sK[threadIdx.x] = Kall[9 * pb + threadIdx.x];
    PoseCtx c;
    c.K = sK;
    c.Ms = MsAll + (size_t)3 * ptsStride * pb;
    c.ms = msAll + (size_t)2 * ptsStride * pb;
    c.npts = npts;
    c.red = red;
    c.myEntry = solve_entry_of_lane(threadIdx.x & 63);
    c.Ws = wsScratch ? (wsScratch + (size_t)ptsStride * pb) : WsL;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double w[3] = {0, 0, 0};
        w[a] = 1e-8;
        so3_exp(w, c.dR[a]);
    }
    if ((int)threadIdx.x < npts) {
        const int i = threadIdx.x;
        c.pM[0] = c.Ms[3 * i], c.pM[1] = c.Ms[3 * i + 1], c.pM[2] = c.Ms[3 * i + 2], c.pm[0] = c.ms[2 * i], c.pm[1] = c.ms[2 * i + 1];
    } else {
        c.pM[0] = c.pM[1] = c.pM[2] = c.pm[0] = c.pm[1] = 0;
    }
    for (int i = threadIdx.x; i < npts; i += PB)
        c.Ws[i] = prevErrs ? tukey(fabs(prevErrs[(size_t)ptsStride * pb + i]), tau) : 1.0;  // :641-655
    __syncthreads();

    cs_pose_option opt = optAll[pb];
    double R[9], t[3], R_opt[9], t_opt[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = R_opt[i] = R0all[9 * pb + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = t_opt[i] = t0all[3 * pb + i];
    bool ret = true;
    int k = 0, par = 0;
    opt.errRW = -1;
    opt.verboseRW = 0;
    for (; k < opt.maxIterRW; ++k) {  // :664
        // (round k > 0: the Tukey weights from the residuals at the pose it starts from, :687-701, inside its first pass)
        if (!weighted_lm<PB>(c, R, t, R_opt, t_opt, opt, k > 0, tau, par)) {
            ret = false;
            break;
        }
        opt.lambda0 = opt.lambda;
        if (opt.errRW < 0) {
            opt.errRW = opt.err;
        } else {
            if (fabs(opt.err - opt.errRW) < opt.epsErrorChangeRW) {
                ret = true;
                break;
            }
            opt.errRW = opt.err;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = R_opt[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = t_opt[i];
    }
    opt.nIterRW = k;
    if (threadIdx.x == 0) {
        for (int i = 0; i < 9; ++i) RoptAll[9 * pb + i] = R_opt[i];
        for (int i = 0; i < 3; ++i) toptAll[3 * pb + i] = t_opt[i];
        optAll[pb] = opt;
        okAll[pb] = ret ? 1 : 0;
    }
}

// Host-pointer entry point: every input in ONE pinned staging block / one device block / one H2D copy, every output
// in one block and one D2H copy behind a single synchronisation (a dozen small pageable copies cost more than the
// solve itself).  Layout in doubles: in  = K 9 | R0 9 | t0 3 | opt 12 | npts (as int) 1 | Ms 3n | ms 2n | prev n
//                                    out = Ropt 9 | topt 3 | opt 12 | ok (as int) 1
struct PoseScratch {
    int device = -1;
    size_t cap = 0;  // points
    double *d_in = nullptr, *d_out = nullptr, *d_ws = nullptr;
    double *h_in = nullptr, *h_out = nullptr;  // pinned
    hipStream_t stream = nullptr;
};
constexpr size_t PS_IN_HEAD = 9 + 9 + 3 + 12 + 1, PS_OUT = 9 + 3 + 12 + 1;
static_assert(sizeof(cs_pose_option) == 96, "cs_pose_option is 12 doubles");
thread_local PoseScratch g_ps;

int ensure_scratch(int device, size_t npts) {
    CS_HIP(hipSetDevice(device));
    PoseScratch& s = g_ps;
    if (s.device != device) {
        s = PoseScratch();
        s.device = device;
        CS_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        CS_HIP(hipMalloc((void**)&s.d_out, PS_OUT * 8));
        CS_HIP(hipHostMalloc((void**)&s.h_out, PS_OUT * 8, hipHostMallocDefault));
    }
    if (npts > s.cap || !s.d_in) {
        size_t cap = npts < 1024 ? 1024 : npts;
        if (s.d_in) {
            (void)hipFree(s.d_in);
            (void)hipFree(s.d_ws);
            (void)hipHostFree(s.h_in);
        }
        CS_HIP(hipMalloc((void**)&s.d_in, (PS_IN_HEAD + 6 * cap) * 8));
        CS_HIP(hipMalloc((void**)&s.d_ws, cap * 8));
        CS_HIP(hipHostMalloc((void**)&s.h_in, (PS_IN_HEAD + 6 * cap) * 8, hipHostMallocDefault));
        s.cap = cap;
    }
    return CS_OK;
}

int launch_intracam(hipStream_t stream, int nProb, int ptsStride, const double* K, const double* R0, const double* t0,
                    const int* npts, const double* prevErrs, const double* Ms, const double* ms, double tau,
                    double* R_opt, double* t_opt, cs_pose_option* opt, int* ok, double* wsScratch) {
    if (ptsStride <= SMALL_NPTS) {
        size_t lds = sizeof(double) * (2 * 1 * NSUM + 12 + (wsScratch ? 0 : ptsStride));
        hipLaunchKernelGGL(k_intracam<64>, dim3(nProb), dim3(64), lds, stream, ptsStride, K, R0, t0, npts, prevErrs, Ms,
                           ms, tau, R_opt, t_opt, opt, ok, wsScratch);
    } else {
        size_t lds = sizeof(double) * (2 * 4 * NSUM + 12 + (wsScratch ? 0 : ptsStride));
        hipLaunchKernelGGL(k_intracam<256>, dim3(nProb), dim3(256), lds, stream, ptsStride, K, R0, t0, npts, prevErrs,
                           Ms, ms, tau, R_opt, t_opt, opt, ok, wsScratch);
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}

}  // namespace

extern "C" {

void cs_pose_option_default(cs_pose_option* o) {  // SL_IntraCamPose.h:42-46
    memset(o, 0, sizeof(*o));
    o->maxIterLM = 100;
    o->maxIterRW = 5;
    o->epsErrorChangeLM = 1e-7;
    o->epsParamChangeLM = 1e-6;
    o->epsErrorChangeRW = 1e-6;
    o->lambda0 = 1e-3;
}

int cs_pose_intracam(const double K[9], const double R0[9], const double t0[3], int npts, const double* prevErrs,
                     const double* Ms, const double* ms, double tau, double R_opt[9], double t_opt[3],
                     cs_pose_option* opt, int device) {
    if (!K || !R0 || !t0 || npts < 0 || (npts > 0 && (!Ms || !ms)) || !R_opt || !t_opt || !opt) {
        cs_set_error("cs_pose_intracam: bad arguments");
        return CS_ERR_INVALID;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        cs_set_error("cs_pose_intracam: no usable HIP device %d; there is no CPU fallback", device);
        return CS_ERR_NO_DEVICE;
    }
    int rc = ensure_scratch(device, (size_t)npts);
    if (rc) return rc;
    PoseScratch& s = g_ps;
    const size_t np = (size_t)npts;
    double* h = s.h_in;
    memcpy(h, K, 72);
    memcpy(h + 9, R0, 72);
    memcpy(h + 18, t0, 24);
    memcpy(h + 21, opt, sizeof(*opt));
    memcpy(h + 33, &npts, sizeof(int));
    if (npts > 0) {
        memcpy(h + PS_IN_HEAD, Ms, np * 24);
        memcpy(h + PS_IN_HEAD + 3 * np, ms, np * 16);
        if (prevErrs) memcpy(h + PS_IN_HEAD + 5 * np, prevErrs, np * 8);
    }
    const size_t inDoubles = PS_IN_HEAD + (prevErrs ? 6 : 5) * np;
    CS_HIP(hipMemcpyAsync(s.d_in, h, inDoubles * 8, hipMemcpyHostToDevice, s.stream));
    CS_HIP(hipMemcpyAsync(s.d_out + 12, s.d_in + 21, sizeof(*opt), hipMemcpyDeviceToDevice, s.stream));  // opt is in/out
    const int stride = npts > 0 ? npts : 1;
    double* d = s.d_in;
    rc = launch_intracam(s.stream, 1, stride, d, d + 9, d + 18, (const int*)(d + 33), prevErrs ? d + PS_IN_HEAD + 5 * np : nullptr,
                         d + PS_IN_HEAD, d + PS_IN_HEAD + 3 * np, tau, s.d_out, s.d_out + 9, (cs_pose_option*)(s.d_out + 12),
                         (int*)(s.d_out + 24), stride > WS_LDS_MAX ? s.d_ws : nullptr);
    if (rc) return rc;
    CS_HIP(hipMemcpyAsync(s.h_out, s.d_out, PS_OUT * 8, hipMemcpyDeviceToHost, s.stream));
    CS_HIP(hipStreamSynchronize(s.stream));
    memcpy(R_opt, s.h_out, 72);
    memcpy(t_opt, s.h_out + 9, 24);
    memcpy(opt, s.h_out + 12, sizeof(*opt));
    int ok = 0;
    memcpy(&ok, s.h_out + 24, sizeof(int));
    return ok;
}

int cs_pose_intracam_batch_dev(int device, void* hip_stream, int nProb, int ptsStride, const double* K,
                               const double* R0, const double* t0, const int* npts, const double* prevErrs,
                               const double* Ms, const double* ms, double tau, double* R_opt, double* t_opt,
                               cs_pose_option* opt, int* ok) {
    if (nProb <= 0 || ptsStride <= 0 || !K || !R0 || !t0 || !npts || !Ms || !ms || !R_opt || !t_opt || !opt || !ok) {
        cs_set_error("cs_pose_intracam_batch_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (ptsStride > WS_LDS_MAX) {
        cs_set_error("cs_pose_intracam_batch_dev: ptsStride %d > %d (weights must fit LDS in the batched form)", ptsStride,
                     WS_LDS_MAX);
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(device));
    return launch_intracam((hipStream_t)hip_stream, nProb, ptsStride, K, R0, t0, npts, prevErrs, Ms, ms, tau, R_opt, t_opt,
                           opt, ok, nullptr);
}

}  // extern "C"
