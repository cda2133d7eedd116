// ba.hip -- robust multi-camera bundle adjustment on gfx950 (binary64): the bundleAdjustRobust contract.
//
// Replaces the external LibVisualSLAM call bundleAdjustRobust(nCamsCon,Ks,Rs,Ts,nPtsCon,pts,meas,maxErr,
// maxIter,innerMaxIter) reached from src/app/SL_CoSLAMRobustBA.cpp:170-180 (local BA),
// SL_InterCamPoseEstimator.cpp:92-95 (inter-camera pose) and SL_MergeCameraGroup.cpp:646-647.  The
// algorithm is the one defined in DESIGN.md "Robust BA" (the library is not vendored: parity unpinned).
//
// Design: measurements live in HBM grouped by point (CSR), with a second index by camera and a dense
// (point, camera) -> measurement table.  One LM step is a short chain of kernels and the LM / outlier
// control flow stays on the device (a state word every kernel checks on entry), so the host enqueues the
// whole maxIter x innerMaxIter schedule without a single synchronisation (four launches per LM step):
//   k_linearize   one wave per point: residual + analytic Jacobians per measurement (lane = measurement),
//                 W_ij = Jc^T Jp to HBM, V_i and g_i folded across the wave with butterflies, V_i^-1 stored;
//                 k_linearize_seg8 / k_update_seg8: EIGHT lanes per point (eight points per wave) when no point has more
//                 than 8 measurements -- the local-BA case, where a wave per point leaves 59 lanes idle
//                 (COSLAM_BA_SEG8=0 forces the wave-per-point kernels, for A/B runs);
//   k_schur_part  (orders <= 36) one WAVE per (camera pair, point slice): partial S_jk, rhs_j, U_j, g_j, the 69 sums
//                 folded with a transposed butterfly; k_schur (larger systems): one workgroup per camera pair writes
//                 S_jk = [j==k](U_j + lambda I) - sum_i W_ij V_i^-1 W_ik^T through the dense table;
//   k_update<N>   N > 0: every workgroup adds the slice partials in LDS and factorises the reduced system out of the
//                 registers of one wave; then the tentative step (cameras R exp(w), t + dt; points by
//                 back-substitution, wave per point) and the tentative cost of the wave's own measurements;
//                 N = 0: the system was solved by k_solve_blocked (one workgroup, LDS, orders 37..176), k_cholflow (one
//                 dataflow launch, a workgroup per block column, orders 177..1040: ba_cholflow_dev.h) or, beyond that,
//                 k_chol_panel / k_chol_trail / k_chol_trsv (HBM, two launches per block);
//   k_control_step  one workgroup: fixed-order sum of the partials, accept/reject, lambda, commit, stop flags
//                 (k_control: the same at the start of an outer round);
//   k_cost / k_flag  start-of-round cost; outlier flags (residual > maxErr) and the "flags changed" bit.
// Every sum has a fixed order (no atomics): results are run-to-run identical.
// MFMA: the f64 matrix peak of MI355X equals its f64 vector peak, and the Schur products of the local BA are 6x3 blocks --
// wave-shuffle territory.  The matrix cores are used where they save instructions on a latency chain (the block updates and
// panels of k_solve_blocked / k_cholflow) and where the contraction is large and dense (ba_syrk_dev.h: the Schur sum of a
// sliding-window BA as Z Z^T, 0.72 of the f64 matrix peak at cfg5).
// The distributed solve (points sliced by rank, all-reduce of S || rhs per LM step) reuses these kernels through
// the cs_ba_dist_* phase API at the end of this file.
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include <chrono>
#include <string>

#include "cs_common.h"

#ifndef CS_BA_PRIO
#define CS_BA_PRIO 1  // s_setprio of the LM-step kernels' waves: the persistent tracker's base level (it runs at 1..3); 0 costs 2 % of the loop, 3 is erratic
#endif
#define CS_BA_SETPRIO() do { if (CS_BA_PRIO) __builtin_amdgcn_s_setprio(CS_BA_PRIO); } while (0)
#pragma clang fp contract(off)

struct cs_ba_stats_dev {
    double cost0, cost;
    int nIterTotal, nOuter, nOutliers, flags;  // flags: CS_BA_FLAG_* (coslam_hip.h)
};

namespace {

struct BaState {
    double lambda, cost, cost_new, step2, cost0;
    int inner_it, inner_done, all_done, chol_ok, changed, nIterTotal, nOuter, nOutliers, first_cost;
    int seq;            // packed path: LM steps started so far in this solve (diagnostic)
    int pending;        // packed path: a tentative step awaits its accept / reject (decided by the next k_lin_packed)
    int cur;            // packed path: which estimate is current: 0 = Rs / Ts / pts, 1 = Rn / Tn / Mn
    int nCholFail;      // LM steps whose reduced system could not be factorised (not positive definite, NaN, time-out)
    int nAccepted;      // LM steps that were accepted
    int solverTimeout;  // the dataflow Cholesky gave up waiting for a block column (scheduling stall, not arithmetic)
    int fuseEpoch;      // packed path: fused update + linearisation launches that went through their grid barrier in this solve
};

struct BaDev {
    int C, P, nObs, nCamsCon, nPtsCon, nc, n;
    const double* Ks;
    double *Rs, *Ts, *pts;  // current
    double *Rn, *Tn, *Mn;   // tentative
    const int *obs_ptr, *obs_cam, *obs_pt, *cam_ptr, *cam_obs, *obs_of;
    const double* obs_xy;
    int* outlier;
    double *Jc, *e, *W, *Vinv, *gp, *S, *rhs, *costPart, *stepPart;
    BaState* st;
    int nCostBlocks, nUpdBlocks, nSlices;
    int pLo, pHi, addLambda;  // point slice owned by this rank ([0, P) and 1 in the single-process solve)
    int maxObsPerPoint;       // largest measurement count of a point (computed at upload)
    double* scal;             // [4] cost / point-step / flags-changed / outlier-count partials (distributed solve)
    double* schurPart;  // [nPairs][nSlices][72] partial Schur blocks (orders <= 36)
    // camera-pair lists (built at upload when they are small enough, else null): for cameras ca <= cb the measurements
    // {oa, ob, point} of the points both see (ca == cb: every measurement of the camera), pair index
    // ca C - ca (ca - 1) / 2 + (cb - ca)
    const int* pairPtr;
    const int4* pairEnt;
    double maxErr;
    int innerMaxIter;
    // packed path (ba_packed_dev.h): whole points back to back in the lanes of a wave
    const int* waveStart;  // [nPackWaves + 1] first measurement of every wave
    int nPackWaves;
    double* Y;             // [nObs][18]  W V^-1
    BaState* stn;          // the state the launch writes (see ba_packed_dev.h)
    int* hostState;        // worker-run solves: pinned host {inner_done, all_done}, written by the last launch of a chunk / tail
                           // (a copy node between the chunks cost 3.6 us + ~12 us of gap on the chain); null: nobody listens
};

// the word the worker's host thread reads between segments (system scope: the store must have left the device when the event
// behind the launch fires)
__device__ __forceinline__ void ba_publish_state(const BaDev& D, int inner_done, int all_done) {
    if (!D.hostState) return;
    __hip_atomic_store(D.hostState, inner_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(D.hostState + 1, all_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ double wsum(double v) { return cs_wave_sum_d(v); }

__device__ __forceinline__ void so3_exp(const double w[3], double R[9]) {
    double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (th == 0) {
        R[0] = 1, R[1] = 0, R[2] = 0, R[3] = 0, R[4] = 1, R[5] = 0, R[6] = 0, R[7] = 0, R[8] = 1;
        return;
    }
    double h0 = w[0] / th, h1 = w[1] / th, h2 = w[2] / th;
    double st = sin(th), ct = 1 - cos(th);
    R[0] = -ct * h1 * h1 - ct * h2 * h2 + 1;
    R[1] = ct * h0 * h1 - st * h2;
    R[2] = st * h1 + ct * h0 * h2;
    R[3] = st * h2 + ct * h0 * h1;
    R[4] = -ct * h0 * h0 - ct * h2 * h2 + 1;
    R[5] = ct * h1 * h2 - st * h0;
    R[6] = ct * h0 * h2 - st * h1;
    R[7] = st * h0 + ct * h1 * h2;
    R[8] = -ct * h0 * h0 - ct * h1 * h1 + 1;
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

// same arithmetic as oracle oba_residual()
template <bool JAC>
__device__ __forceinline__ bool residual(const double* K, const double* R, const double* t, const double* M,
                                         const double* m, double* e, double* Jc, double* Jp) {
    double X[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) X[r] = R[3 * r] * M[0] + R[3 * r + 1] * M[1] + R[3 * r + 2] * M[2] + t[r];
    double u = K[0] * X[0] + K[1] * X[1] + K[2] * X[2];
    double v = K[3] * X[0] + K[4] * X[1] + K[5] * X[2];
    double w = K[6] * X[0] + K[7] * X[1] + K[8] * X[2];
    if (!(w > 1e-12)) {
        e[0] = e[1] = 1e150;
        if (JAC) {
#pragma unroll
            for (int q = 0; q < 12; ++q) Jc[q] = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) Jp[q] = 0;
        }
        return false;
    }
    double mx = u / w, my = v / w;
    e[0] = m[0] - mx;
    e[1] = m[1] - my;
    if (JAC) {
        double A[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A[c] = (K[c] - mx * K[6 + c]) / w;
            A[3 + c] = (K[3 + c] - my * K[6 + c]) / w;
        }
        double Mx[9] = {0, -M[2], M[1], M[2], 0, -M[0], -M[1], M[0], 0};
        double RMx[9];
        mat33AB(R, Mx, RMx);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Jc[6 * r + c] = -(A[3 * r] * RMx[c] + A[3 * r + 1] * RMx[3 + c] + A[3 * r + 2] * RMx[6 + c]);
                Jc[6 * r + 3 + c] = A[3 * r + c];
                Jp[3 * r + c] = A[3 * r] * R[c] + A[3 * r + 1] * R[3 + c] + A[3 * r + 2] * R[6 + c];
            }
    }
    return true;
}

__device__ __forceinline__ bool inv33(const double* V, double* Vi) {
    double a = V[0], b = V[1], c = V[2], d = V[3], e = V[4], f = V[5], g = V[6], h = V[7], i = V[8];
    double A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
    double det = a * A + b * B + c * Cc;
    if (!(fabs(det) > 0)) return false;
    double r = 1.0 / det;
    Vi[0] = A * r;
    Vi[1] = -(b * i - c * h) * r;
    Vi[2] = (b * f - c * e) * r;
    Vi[3] = B * r;
    Vi[4] = (a * i - c * g) * r;
    Vi[5] = -(a * f - c * d) * r;
    Vi[6] = Cc * r;
    Vi[7] = -(a * h - b * g) * r;
    Vi[8] = (a * e - b * d) * r;
    return true;
}

#define BA_ACTIVE(D) (!((D).st->all_done) && !((D).st->inner_done))

// ---- one wave per point --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_linearize(BaDev D) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= D.P || i < D.pLo || i >= D.pHi) return;
    const int o0 = D.obs_ptr[i], o1 = D.obs_ptr[i + 1];
    int nIn = 0;
    for (int o = o0 + lane; o < o1; o += 64) nIn += D.outlier[o] ? 0 : 1;
    nIn = cs_wave_sum_i(nIn);
    // a point seen by fewer than two inlier measurements has no depth constraint: hold it (DESIGN.md "Robust BA")
    const bool freeP = (i >= D.nPtsCon) && (nIn >= 2);
    const double lambda = D.st->lambda;
    const double M[3] = {D.pts[3 * i], D.pts[3 * i + 1], D.pts[3 * i + 2]};
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // V upper (6) + g (3)
    for (int o = o0 + lane; o < o1; o += 64) {
        double* Wo = D.W + 18 * (size_t)o;
        double* Jo = D.Jc + 12 * (size_t)o;
        double e[2], Jc[12], Jp[6];
        const int j = D.obs_cam[o];
        const bool in = !D.outlier[o];
        if (in) {
            residual<true>(D.Ks + 9 * j, D.Rs + 9 * j, D.Ts + 3 * j, M, D.obs_xy + 2 * (size_t)o, e, Jc, Jp);
        } else {
            e[0] = e[1] = 0;
#pragma unroll
            for (int q = 0; q < 12; ++q) Jc[q] = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) Jp[q] = 0;
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) Jo[q] = Jc[q];
        D.e[2 * (size_t)o] = e[0];
        D.e[2 * (size_t)o + 1] = e[1];
        const bool freeC = (j >= D.nCamsCon);
        if (in && freeP) {
            acc[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3];
            acc[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
            acc[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5];
            acc[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
            acc[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5];
            acc[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
            acc[6] += Jp[0] * e[0] + Jp[3] * e[1];
            acc[7] += Jp[1] * e[0] + Jp[4] * e[1];
            acc[8] += Jp[2] * e[0] + Jp[5] * e[1];
        }
        const bool w = in && freeP && freeC;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Wo[3 * r + c] = w ? (Jc[r] * Jp[c] + Jc[6 + r] * Jp[3 + c]) : 0.0;
    }
    cs_wave_sum_many_d<9>(acc);
    if (lane == 0) {
        double Vi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (freeP) {
            double V[9] = {acc[0] + lambda, acc[1], acc[2], acc[1], acc[3] + lambda, acc[4], acc[2], acc[4], acc[5] + lambda};
            if (!inv33(V, Vi)) {
#pragma unroll
                for (int q = 0; q < 9; ++q) Vi[q] = 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) D.Vinv[9 * (size_t)i + q] = Vi[q];
        D.gp[3 * (size_t)i] = acc[6];
        D.gp[3 * (size_t)i + 1] = acc[7];
        D.gp[3 * (size_t)i + 2] = acc[8];
    }
}

// ---- eight lanes per point ---------------------------------------------------------------------------------------
// A local-BA point has a handful of measurements (one per key frame that sees it), so a wave per point leaves 59 of 64
// lanes idle through ~400 f64 instructions per measurement -- and with two such workgroups per CU (the 64-CU partition)
// the launch is bound by exactly that: 3.4 us of arithmetic on the whole chip, 5.4 us on the partition (s_memtime).
// When no point has more than 8 measurements (maxObsPerPoint, known at upload) eight points share a wave: lane = (point,
// measurement), the per-point sums are three DPP steps inside the 8-lane segment, one lane per segment inverts V.
__device__ __forceinline__ double seg8_sum(double v) {
    v += cs_dpp_d<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += cs_dpp_d<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += cs_dpp_d<0x141, 0xf>(v);  // row_half_mirror: the other quad of the segment
    return v;
}
__global__ __launch_bounds__(256) void k_linearize_seg8(BaDev D) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), k = threadIdx.x & 7;
    const bool hasP = !(i >= D.P || i < D.pLo || i >= D.pHi);
    const double lambda = D.st->lambda;
    int o = -1;
    double M[3] = {0, 0, 0};
    if (hasP) {
        const int o0 = D.obs_ptr[i], o1 = D.obs_ptr[i + 1];
        if (o0 + k < o1) o = o0 + k;
        M[0] = D.pts[3 * (size_t)i];
        M[1] = D.pts[3 * (size_t)i + 1];
        M[2] = D.pts[3 * (size_t)i + 2];
    }
    int j = 0;
    bool in = false;
    if (o >= 0) {
        j = D.obs_cam[o];
        in = !D.outlier[o];
    }
    const int nIn = (int)seg8_sum(in ? 1.0 : 0.0);
    // a point seen by fewer than two inlier measurements has no depth constraint: hold it (DESIGN.md "Robust BA")
    const bool freeP = hasP && (i >= D.nPtsCon) && (nIn >= 2);
    double e[2] = {0, 0}, Jc[12], Jp[6];
#pragma unroll
    for (int q = 0; q < 12; ++q) Jc[q] = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) Jp[q] = 0;
    if (in) residual<true>(D.Ks + 9 * j, D.Rs + 9 * j, D.Ts + 3 * j, M, D.obs_xy + 2 * (size_t)o, e, Jc, Jp);
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // V upper (6) + g (3)
    if (in && freeP) {
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
    if (o >= 0) {
        double* Wo = D.W + 18 * (size_t)o;
        double* Jo = D.Jc + 12 * (size_t)o;
#pragma unroll
        for (int q = 0; q < 12; ++q) Jo[q] = Jc[q];
        D.e[2 * (size_t)o] = e[0];
        D.e[2 * (size_t)o + 1] = e[1];
        const bool w = in && freeP && (j >= D.nCamsCon);
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Wo[3 * r + c] = w ? (Jc[r] * Jp[c] + Jc[6 + r] * Jp[3 + c]) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] = seg8_sum(acc[q]);
    if (hasP && k == 0) {
        double Vi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (freeP) {
            double V[9] = {acc[0] + lambda, acc[1], acc[2], acc[1], acc[3] + lambda, acc[4], acc[2], acc[4], acc[5] + lambda};
            if (!inv33(V, Vi)) {
#pragma unroll
                for (int q = 0; q < 9; ++q) Vi[q] = 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) D.Vinv[9 * (size_t)i + q] = Vi[q];
        D.gp[3 * (size_t)i] = acc[6];
        D.gp[3 * (size_t)i + 1] = acc[7];
        D.gp[3 * (size_t)i + 2] = acc[8];
    }
}

// ---- one workgroup per camera pair (ja <= jb) ----------------------------------------------------
// Diagonal pairs also form U_j = sum Jc^T Jc + lambda I and g_j = sum Jc^T e over the camera's own measurement
// list, so the whole reduced system S, rhs is written by this one launch (no read-modify-write between kernels).
__global__ __launch_bounds__(256) void k_schur(BaDev D) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    __shared__ double red[4][42];
    __shared__ double redU[4][27];
    // decode the pair from the linear block index over the upper triangle
    // the diagonal pairs first: they carry the longest lists (+ U_j, g_j), so they must not be the launch's last workgroups
    int ja, jb;
    if ((int)blockIdx.x < D.nc) {
        ja = jb = blockIdx.x;
    } else {
        int pair = blockIdx.x - D.nc;
        ja = 0;
        while (pair >= D.nc - 1 - ja) {
            pair -= D.nc - 1 - ja;
            ++ja;
        }
        jb = ja + 1 + pair;
    }
    const int ca = ja + D.nCamsCon, cb = jb + D.nCamsCon;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool diag = (ja == jb);
    // The points both cameras measure are found by walking camera a's OWN measurement list (ascending point index) and
    // looking camera b up in the dense (point, camera) table: one dependent lookup per entry of a list a few hundred
    // long, where walking all P points cost two lookups each, most of them misses.  The diagonal pair takes U_j and
    // g_j from the same entries.
    double u[27], acc[42];
#pragma unroll
    for (int q = 0; q < 27; ++q) u[q] = 0;
#pragma unroll
    for (int q = 0; q < 42; ++q) acc[q] = 0;
    const int sBeg = D.cam_ptr[ca], sEnd = D.cam_ptr[ca + 1];
    for (int s0 = sBeg; s0 < sEnd; s0 += 4 * 256) {
        // the index chain of four entries per thread goes out as one batch (entry -> point -> partner measurement)
        int oa[4], ip[4], ob[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int s = s0 + threadIdx.x + 256 * t;
            oa[t] = (s < sEnd) ? D.cam_obs[s] : -1;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ip[t] = -1;
            if (oa[t] >= 0 && !D.outlier[oa[t]]) ip[t] = D.obs_pt[oa[t]];
            if (ip[t] < D.pLo || ip[t] >= D.pHi) ip[t] = -1;  // another rank's point
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ob[t] = -1;
            if (ip[t] >= D.nPtsCon) ob[t] = diag ? oa[t] : D.obs_of[(size_t)ip[t] * D.C + cb];
            if (ob[t] >= 0 && D.outlier[ob[t]]) ob[t] = -1;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (diag && ip[t] >= 0) {  // U_j, g_j: every inlier measurement of the camera, fixed points included
                const double* J = D.Jc + 12 * (size_t)oa[t];
                const double e0 = D.e[2 * (size_t)oa[t]], e1 = D.e[2 * (size_t)oa[t] + 1];
                int q = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = r; c < 6; ++c) u[q++] += J[r] * J[c] + J[6 + r] * J[6 + c];
#pragma unroll
                for (int r = 0; r < 6; ++r) u[21 + r] += J[r] * e0 + J[6 + r] * e1;
            }
            if (ob[t] < 0) continue;
            const double* Wa = D.W + 18 * (size_t)oa[t];
            const double* Wb = D.W + 18 * (size_t)ob[t];
            const double* Vi = D.Vinv + 9 * (size_t)ip[t];
            double Y[18];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Y[3 * r + c] = Wa[3 * r] * Vi[c] + Wa[3 * r + 1] * Vi[3 + c] + Wa[3 * r + 2] * Vi[6 + c];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    acc[6 * r + c] += Y[3 * r] * Wb[3 * c] + Y[3 * r + 1] * Wb[3 * c + 1] + Y[3 * r + 2] * Wb[3 * c + 2];
            if (diag) {
                const double* g = D.gp + 3 * (size_t)ip[t];
#pragma unroll
                for (int r = 0; r < 6; ++r) acc[36 + r] += Y[3 * r] * g[0] + Y[3 * r + 1] * g[1] + Y[3 * r + 2] * g[2];
            }
        }
    }
    if (diag) {
        cs_reduce_many<27>(u, lane);  // transposed butterfly: total q ends up in lane cs_reduce_owner<27>(q)
        const int q = cs_reduce_index<27>(lane);
        if (q >= 0) redU[wv][q] = u[0];
    }
    {
        cs_reduce_many<42>(acc, lane);
        const int q = cs_reduce_index<42>(lane);
        if (q >= 0) red[wv][q] = acc[0];
    }
    __syncthreads();
    if (threadIdx.x < 42) {
        const int q = threadIdx.x;
        const double s = ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q];
        const int n = D.n;
        if (q < 36) {
            const int r = q / 6, c = q - 6 * r;
            if (diag) {
                // U_j entry (upper-triangular rank of (min,max)) + lambda on the diagonal - Schur sum
                const int rr = r < c ? r : c, cc = r < c ? c : r;
                const int uq = rr * 6 - (rr * (rr - 1)) / 2 + (cc - rr);
                const double uv = ((redU[0][uq] + redU[1][uq]) + redU[2][uq]) + redU[3][uq];
                D.S[(size_t)(6 * ja + r) * n + 6 * ja + c] = (uv + ((r == c && D.addLambda) ? D.st->lambda : 0.0)) - s;
            } else {
                D.S[(size_t)(6 * ja + r) * n + 6 * jb + c] = -s;
                D.S[(size_t)(6 * jb + c) * n + 6 * ja + r] = -s;
            }
        } else if (diag) {
            const int r = q - 36;
            const double gv = ((redU[0][21 + r] + redU[1][21 + r]) + redU[2][21 + r]) + redU[3][21 + r];
            D.rhs[6 * ja + r] = gv - s;
        }
    }
}

// ---- the same reduced system from the camera-pair lists built at upload ------------------------------------------------
// One workgroup per pair of free cameras; a thread takes entries {oa, ob, point} of the pair's list -- ONE load instead of
// k_schur's chain of three, and every entry is a hit.  Nothing else has to be tested: k_linearize leaves W = 0 for a
// measurement that is an outlier, belongs to a fixed camera or to a point that is held, Jc = e = 0 for an outlier and
// V^-1 = 0 for a held point, so those entries add zeros; only another rank's points (not linearised here: stale) are skipped.
__global__ __launch_bounds__(256) void k_schur_pairs(BaDev D) {
    CS_BA_SETPRIO();
#ifdef CS_SCHUR_PROBE
    const unsigned long long p0 = __builtin_amdgcn_s_memtime();
#endif
    if (!BA_ACTIVE(D)) return;
    __shared__ double red[4][42];
    __shared__ double redU[4][27];
    // the diagonal pairs first: they carry the longest lists (+ U_j, g_j), so they must not be the launch's last workgroups
    int ja, jb;
    if ((int)blockIdx.x < D.nc) {
        ja = jb = blockIdx.x;
    } else {
        int pair = blockIdx.x - D.nc;
        ja = 0;
        while (pair >= D.nc - 1 - ja) {
            pair -= D.nc - 1 - ja;
            ++ja;
        }
        jb = ja + 1 + pair;
    }
    const int ca = ja + D.nCamsCon, cb = jb + D.nCamsCon;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool diag = (ja == jb);
    const size_t pid = (size_t)ca * D.C - (size_t)ca * (ca - 1) / 2 + (size_t)(cb - ca);
    const int eBeg = D.pairPtr[pid], eEnd = D.pairPtr[pid + 1];
    double u[27], acc[42];
#pragma unroll
    for (int q = 0; q < 27; ++q) u[q] = 0;
#pragma unroll
    for (int q = 0; q < 42; ++q) acc[q] = 0;
    for (int en = eBeg + (int)threadIdx.x; en < eEnd; en += 256) {
        const int4 E = D.pairEnt[en];
        const int oa = E.x, ob = E.y, ip = E.z;
        if (ip < D.pLo || ip >= D.pHi) continue;  // another rank's point
        if (diag) {  // U_j, g_j: every measurement of the camera (outliers carry Jc = e = 0), fixed points included
            const double* J = D.Jc + 12 * (size_t)oa;
            const double e0 = D.e[2 * (size_t)oa], e1 = D.e[2 * (size_t)oa + 1];
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) u[q++] += J[r] * J[c] + J[6 + r] * J[6 + c];
#pragma unroll
            for (int r = 0; r < 6; ++r) u[21 + r] += J[r] * e0 + J[6 + r] * e1;
        }
        const double* Wa = D.W + 18 * (size_t)oa;
        const double* Wb = D.W + 18 * (size_t)ob;
        const double* Vi = D.Vinv + 9 * (size_t)ip;
        double Y[18];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Y[3 * r + c] = Wa[3 * r] * Vi[c] + Wa[3 * r + 1] * Vi[3 + c] + Wa[3 * r + 2] * Vi[6 + c];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                acc[6 * r + c] += Y[3 * r] * Wb[3 * c] + Y[3 * r + 1] * Wb[3 * c + 1] + Y[3 * r + 2] * Wb[3 * c + 2];
        if (diag) {
            const double* g = D.gp + 3 * (size_t)ip;
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[36 + r] += Y[3 * r] * g[0] + Y[3 * r + 1] * g[1] + Y[3 * r + 2] * g[2];
        }
    }
#ifdef CS_SCHUR_PROBE
    asm volatile("" : "+v"(acc[0]));
    const unsigned long long p1 = __builtin_amdgcn_s_memtime();
#endif
    if (diag) {
        cs_reduce_many<27>(u, lane);
        const int q = cs_reduce_index<27>(lane);
        if (q >= 0) redU[wv][q] = u[0];
    }
    {
        cs_reduce_many<42>(acc, lane);
        const int q = cs_reduce_index<42>(lane);
        if (q >= 0) red[wv][q] = acc[0];
    }
#ifdef CS_SCHUR_PROBE
    const unsigned long long p2 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef CS_SCHUR_PROBE
    if (threadIdx.x == 0 && (diag ? ja < 3 : blockIdx.x == 7) && D.st->nIterTotal == 3)
        printf("k_schur_pairs block %d diag %d (%d entries): state+loop %llu reduce %llu\n", (int)blockIdx.x, (int)diag, eEnd - eBeg, p1 - p0, p2 - p1);
#endif
    if (threadIdx.x < 42) {
        const int q = threadIdx.x;
        const double s = ((red[0][q] + red[1][q]) + red[2][q]) + red[3][q];
        const int n = D.n;
        if (q < 36) {
            const int r = q / 6, c = q - 6 * r;
            if (diag) {
                const int rr = r < c ? r : c, cc = r < c ? c : r;
                const int uq = rr * 6 - (rr * (rr - 1)) / 2 + (cc - rr);
                const double uv = ((redU[0][uq] + redU[1][uq]) + redU[2][uq]) + redU[3][uq];
                D.S[(size_t)(6 * ja + r) * n + 6 * ja + c] = (uv + ((r == c && D.addLambda) ? D.st->lambda : 0.0)) - s;
            } else {
                D.S[(size_t)(6 * ja + r) * n + 6 * jb + c] = -s;
                D.S[(size_t)(6 * jb + c) * n + 6 * ja + r] = -s;
            }
        } else if (diag) {
            const int r = q - 36;
            const double gv = ((redU[0][21 + r] + redU[1][21 + r]) + redU[2][21 + r]) + redU[3][21 + r];
            D.rhs[6 * ja + r] = gv - s;
        }
    }
}

#include "ba_syrk_dev.h"

// ---- small reduced systems (order <= 36): one WAVE per (camera pair, point slice) -------------------------------
// The pair-per-workgroup kernel above walks all points with 256 threads and then folds 42 sums over four waves; at
// the reference's local-BA sizes (3..6 free cameras) that is 6..21 workgroups and two rounds of dependent loads.
// Here the points are cut into D.nSlices contiguous slices and every (pair, slice) is one wave (lane = point); the
// wave totals go to schurPart[pair][slice][72] and the register Cholesky adds the slices in slice order while it
// loads its rows -- deterministic, no atomics, no extra launch.
__global__ __launch_bounds__(64) void k_schur_part(BaDev D) {
    // The launch is a chain of dependent loads (state -> measurement index -> outlier flag -> blocks), a few waves
    // wide: every load whose address is known is issued before the first value is looked at.
    const int lane = threadIdx.x;
    const int pairIdx = blockIdx.x / D.nSlices, slice = blockIdx.x - pairIdx * D.nSlices;
    int ja = 0, pr = pairIdx;
    while (pr >= D.nc - ja) {
        pr -= D.nc - ja;
        ++ja;
    }
    const int jb = ja + pr;
    const int ca = ja + D.nCamsCon, cb = jb + D.nCamsCon;
    const bool diag = (ja == jb);
    const int nFree = D.P - D.nPtsCon;
    const int per = (nFree + D.nSlices - 1) / D.nSlices;
    const int lo = D.nPtsCon + slice * per, hi = min(lo + per, D.P);
    // round trip 1: the state word, this lane's first point and (diagonal pairs) the camera's measurement list bounds
    const int all_done = D.st->all_done, inner_done = D.st->inner_done;
    const int i0 = lo + lane;
    const bool has0 = i0 < hi;
    int oa0 = -1, ob0 = -1;
    double Vi0[9], g0[3];
    if (has0) {
        oa0 = D.obs_of[(size_t)i0 * D.C + ca];
        ob0 = diag ? oa0 : D.obs_of[(size_t)i0 * D.C + cb];
#pragma unroll
        for (int q = 0; q < 9; ++q) Vi0[q] = D.Vinv[9 * (size_t)i0 + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) g0[q] = diag ? D.gp[3 * (size_t)i0 + q] : 0.0;
    }
    int c0 = 0, c1 = 0;
    if (diag) {
        c0 = D.cam_ptr[ca];
        c1 = D.cam_ptr[ca + 1];
    }
    if (all_done || inner_done) return;
    // round trip 2: flags and W blocks of the first point (indices clamped: a missing measurement loads slot 0 and is
    // masked below), and the first entry of the camera's list
    const int perU = (c1 - c0 + D.nSlices - 1) / D.nSlices;
    const int loU = c0 + slice * perU, hiU = min(loU + perU, c1);
    const int sI0 = loU + lane;
    const int oU0 = (diag && sI0 < hiU) ? D.cam_obs[sI0] : -1;
    int outA0 = 1, outB0 = 1;
    double Wa0[18], Wb0[18];
    if (has0) {
        const int qa = oa0 < 0 ? 0 : oa0, qb = ob0 < 0 ? 0 : ob0;
        outA0 = D.outlier[qa];
        outB0 = D.outlier[qb];
#pragma unroll
        for (int q = 0; q < 18; ++q) Wa0[q] = D.W[18 * (size_t)qa + q];
#pragma unroll
        for (int q = 0; q < 18; ++q) Wb0[q] = diag ? Wa0[q] : D.W[18 * (size_t)qb + q];
    }
    double* out = D.schurPart + (size_t)blockIdx.x * 72;
    double acc[42];
#pragma unroll
    for (int q = 0; q < 42; ++q) acc[q] = 0;
    for (int i = i0; i < hi; i += 64) {
        const bool first = (i == i0);
        int oa, ob;
        double Wa[18], Wb[18], Vi[9], g[3];
        if (first) {
            if (oa0 < 0 || outA0 || ob0 < 0 || outB0) continue;
#pragma unroll
            for (int q = 0; q < 18; ++q) {
                Wa[q] = Wa0[q];
                Wb[q] = Wb0[q];
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) Vi[q] = Vi0[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) g[q] = g0[q];
        } else {
            oa = D.obs_of[(size_t)i * D.C + ca];
            if (oa < 0 || D.outlier[oa]) continue;
            ob = diag ? oa : D.obs_of[(size_t)i * D.C + cb];
            if (ob < 0 || D.outlier[ob]) continue;
#pragma unroll
            for (int q = 0; q < 18; ++q) {
                Wa[q] = D.W[18 * (size_t)oa + q];
                Wb[q] = D.W[18 * (size_t)ob + q];
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) Vi[q] = D.Vinv[9 * (size_t)i + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) g[q] = diag ? D.gp[3 * (size_t)i + q] : 0.0;
        }
        double Y[18];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Y[3 * r + c] = Wa[3 * r] * Vi[c] + Wa[3 * r + 1] * Vi[3 + c] + Wa[3 * r + 2] * Vi[6 + c];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c)
                acc[6 * r + c] += Y[3 * r] * Wb[3 * c] + Y[3 * r + 1] * Wb[3 * c + 1] + Y[3 * r + 2] * Wb[3 * c + 2];
        if (diag) {
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[36 + r] += Y[3 * r] * g[0] + Y[3 * r + 1] * g[1] + Y[3 * r + 2] * g[2];
        }
    }
    // round trip 3 (diagonal pairs; independent of the loop above, so it is in flight under it): Jc, e of the list entry
    int outU0 = 1;
    double J0[12], e00 = 0, e01 = 0;
    if (oU0 >= 0) {
        outU0 = D.outlier[oU0];
#pragma unroll
        for (int q = 0; q < 12; ++q) J0[q] = D.Jc[12 * (size_t)oU0 + q];
        e00 = D.e[2 * (size_t)oU0];
        e01 = D.e[2 * (size_t)oU0 + 1];
    }
    cs_reduce_many<42>(acc, lane);
    {
        const int q = cs_reduce_index<42>(lane);
        if (q >= 0) out[q] = acc[0];
    }
    if (diag) {  // U_j = sum Jc^T Jc and g_j = sum Jc^T e over this slice of the camera's own measurement list
        double u[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) u[q] = 0;
        for (int sI = sI0; sI < hiU; sI += 64) {
            double J[12], e0, e1;
            if (sI == sI0) {
                if (outU0) continue;
#pragma unroll
                for (int q = 0; q < 12; ++q) J[q] = J0[q];
                e0 = e00;
                e1 = e01;
            } else {
                const int o = D.cam_obs[sI];
                if (D.outlier[o]) continue;
#pragma unroll
                for (int q = 0; q < 12; ++q) J[q] = D.Jc[12 * (size_t)o + q];
                e0 = D.e[2 * (size_t)o];
                e1 = D.e[2 * (size_t)o + 1];
            }
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
        if (q >= 0) out[42 + q] = u[0];
    }
}

// ---- large reduced systems (order > 138: the LDS single-workgroup Cholesky no longer fits) -----------------
// Right-looking blocked Cholesky on S in HBM, block CB = 32, lower triangle, two launches per block column:
//   k_chol_panel   every workgroup re-factors the 32x32 diagonal block in LDS (cheap, saves a launch and a grid-wide
//                  dependency) and solves 64 rows of the panel below it against L_kk^T (thread per row);
//   k_chol_trail   C -= A_i A_j^T on the 64x64 tiles of the trailing lower triangle, both 64x32 panels staged in LDS,
//                  4x4 outputs per thread in registers;
//   k_chol_trsv    one workgroup: blocked forward and backward substitution of the right-hand side.
// binary64 VALU throughout: on MI355X the f64 MFMA peak equals the f64 vector peak (78.6 TFLOP/s), so the matrix
// cores buy nothing for this contraction; what matters is LDS tiling and enough workgroups (cfg5: C = 120, order 720,
// 58 trailing tiles in the first step).
constexpr int CB = 32;   // block column width
constexpr int CT = 64;   // tile edge of panel / trailing kernels

__global__ __launch_bounds__(256) void k_chol_panel(BaDev D, int k0) {
    if (!BA_ACTIVE(D)) return;
    __shared__ double Lkk[CB][CB + 1];
    __shared__ int okSh;
    const int n = D.n, tid = threadIdx.x;
    const int kb = min(CB, n - k0);
    double* S = D.S;
    for (int q = tid; q < CB * CB; q += 256) {
        const int r = q / CB, c = q - CB * r;
        Lkk[r][c] = (r < kb && c < kb) ? S[(size_t)(k0 + r) * n + k0 + c] : ((r == c) ? 1.0 : 0.0);
    }
    if (tid == 0) okSh = 1;
    __syncthreads();
    // factor the diagonal block: column by column, one wave (rows = lanes), LDS resident
    if (tid < 64) {
        const int i = tid;
        for (int j = 0; j < kb; ++j) {
            double d = Lkk[j][j];
            if (!(d > 0)) {
                if (i == 0) okSh = 0;
                d = 1.0;
            }
            d = sqrt(d);
            const double lij = (i > j && i < kb) ? Lkk[i][j] / d : 0.0;
            __builtin_amdgcn_wave_barrier();
            if (i == j) Lkk[j][j] = d;
            if (i > j && i < kb) Lkk[i][j] = lij;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (i > j && i < kb)
                for (int c = j + 1; c <= i; ++c) Lkk[i][c] -= lij * Lkk[c][j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int q = tid; q < kb * kb; q += 256) {
            const int r = q / kb, c = q - kb * r;
            if (c <= r) S[(size_t)(k0 + r) * n + k0 + c] = Lkk[r][c];
        }
        if (tid == 0 && !okSh) D.st->chol_ok = 0;
    }
    // panel rows below the diagonal block: X L^T = B, thread per row
    const int row = k0 + kb + blockIdx.x * 256 + tid;
    if (row < n) {
        double x[CB];
        double* src = S + (size_t)row * n + k0;
#pragma unroll
        for (int c = 0; c < CB; ++c) x[c] = (c < kb) ? src[c] : 0.0;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            if (j < kb) {
                double v = x[j];
#pragma unroll
                for (int m = 0; m < CB; ++m)
                    if (m < j) v -= x[m] * Lkk[j][m];
                x[j] = v / Lkk[j][j];
            }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (c < kb) src[c] = x[c];
    }
}

__global__ __launch_bounds__(256) void k_chol_trail(BaDev D, int k0) {
    if (!BA_ACTIVE(D)) return;
    __shared__ double Ai[CT][CB + 1];
    __shared__ double Aj[CT][CB + 1];
    const int n = D.n, tid = threadIdx.x;
    const int kb = min(CB, n - k0), t0 = k0 + kb;
    // decode the lower-triangular tile (ti >= tj) from the linear block index
    int tj = 0, rem = blockIdx.x;
    const int nt = (n - t0 + CT - 1) / CT;
    while (rem >= nt - tj) {
        rem -= nt - tj;
        ++tj;
    }
    const int ti = tj + rem;
    const int i0 = t0 + ti * CT, j0 = t0 + tj * CT;
    double* S = D.S;
    for (int q = tid; q < CT * CB; q += 256) {
        const int r = q / CB, c = q - CB * r;
        Ai[r][c] = (i0 + r < n && c < kb) ? S[(size_t)(i0 + r) * n + k0 + c] : 0.0;
        Aj[r][c] = (j0 + r < n && c < kb) ? S[(size_t)(j0 + r) * n + k0 + c] : 0.0;
    }
    __syncthreads();
    const int tr = (tid >> 4) * 4, tc = (tid & 15) * 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0;
#pragma unroll 8
    for (int c = 0; c < CB; ++c) {
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            av[a] = Ai[tr + a][c];
            bv[a] = Aj[tc + a][c];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = i0 + tr + a, c = j0 + tc + b;
            if (r < n && c < n && c <= r) S[(size_t)r * n + c] -= acc[a][b];
        }
}

// L y = b, then L^T x = y, blocked by CB: the diagonal block is solved by thread 0..kb-1 cooperatively through LDS,
// the remaining right-hand side is updated by all threads (one row / column entry each)
__global__ __launch_bounds__(1024) void k_chol_trsv(BaDev D) {
    if (!BA_ACTIVE(D)) return;
    extern __shared__ __attribute__((aligned(16))) double bsh[];  // [n]
    __shared__ double xk[CB];
    const int n = D.n, tid = threadIdx.x;
    const double* L = D.S;
    for (int q = tid; q < n; q += 1024) bsh[q] = D.rhs[q];
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += CB) {
        const int kb = min(CB, n - k0);
        if (tid < 64) {  // forward solve of the diagonal block by one wave (serial over its columns)
            for (int j = 0; j < kb; ++j) {
                const double yj = bsh[k0 + j] / L[(size_t)(k0 + j) * n + k0 + j];
                __builtin_amdgcn_wave_barrier();
                if (tid == j) {
                    bsh[k0 + j] = yj;
                    xk[j] = yj;
                }
                if (tid > j && tid < kb) bsh[k0 + tid] -= L[(size_t)(k0 + tid) * n + k0 + j] * yj;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();
        for (int r = k0 + kb + tid; r < n; r += 1024) {
            double v = bsh[r];
            const double* Lr = L + (size_t)r * n + k0;
            for (int c = 0; c < kb; ++c) v -= Lr[c] * xk[c];
            bsh[r] = v;
        }
        __syncthreads();
    }
    for (int k0 = ((n - 1) / CB) * CB; k0 >= 0; k0 -= CB) {
        const int kb = min(CB, n - k0);
        if (tid < 64) {
            for (int j = kb - 1; j >= 0; --j) {
                const double xj = bsh[k0 + j] / L[(size_t)(k0 + j) * n + k0 + j];
                __builtin_amdgcn_wave_barrier();
                if (tid == j) {
                    bsh[k0 + j] = xj;
                    xk[j] = xj;
                }
                if (tid < j) bsh[k0 + tid] -= L[(size_t)(k0 + j) * n + k0 + tid] * xj;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();
        for (int r = tid; r < k0; r += 1024) {  // x_r -= sum_c L[k0 + c][r] x_{k0 + c}
            double v = bsh[r];
            for (int c = 0; c < kb; ++c) v -= L[(size_t)(k0 + c) * n + r] * xk[c];
            bsh[r] = v;
        }
        __syncthreads();
    }
    for (int q = tid; q < n; q += 1024) D.rhs[q] = bsh[q];
}

__global__ void k_chol_begin(BaDev D) {
    if (!BA_ACTIVE(D)) return;
    D.st->chol_ok = 1;
}

// ---- one workgroup, blocked, LDS-resident: reduced camera systems of order 36 < n <= 176 ---------------------------
// The joint local BA of the 8-camera rig (24 free key-frame cameras, order 144) and the inter-camera solve (order 48)
// land here.  The unblocked kernels pay three workgroup barriers per COLUMN (k_solve<256>) or run on one wave
// (k_solve_wave); the HBM-blocked Cholesky was built for order 720 and costs ~350 us at order 144.  Here the lower
// triangle of S lives in LDS as packed 16 x 16 blocks (order 144: 45 blocks = 90 KB; order 192: 156 KB) and one
// 1024-thread workgroup runs a right-looking blocked factorisation with two barriers per BLOCK column:
//   1. wave 0 factors the diagonal block out of REGISTERS (lane i = row i, sixteen unrolled column steps, every
//      cross-lane read a v_readlane of a constant lane) and leaves L_kk and 1 / diag(L_kk) in LDS -- with look-ahead:
//      the block of step k + 1 is updated first and factored while the other fifteen waves do step k's trailing update;
//   2. panel  A[I][k] <- A[I][k] L_kk^-T by substitution, one row per thread;
//   3. trailing update A[I][J] -= P_I P_J^T, one 4 x 4 register tile per thread.
// The right-hand side rides along as one more row of the matrix (an augmented factorisation), so the forward
// substitution costs no extra phase; the back substitution is a wave-level sweep per diagonal block plus one update
// phase.  Rows n..16*NB-1 are identity padding.
constexpr int SB = 16;
// Row pitch of a block in LDS: 18 doubles = 36 banks, so the sixteen rows of a block start in sixteen different
// bank groups (pitch 16 put every other row on the same banks: 8- and 16-way conflicts in every phase -- 90 us per
// order-144 solve instead of the ~40 below).  Order limit: 66 blocks x 2304 B = 152 KB at NB = 11.
constexpr int SP = 18, SBLK = SB * SP;
constexpr int SB_MAX_ORDER = 176;
__device__ __forceinline__ int sb_off(int I, int J) { return (I * (I + 1) / 2 + J) * SBLK; }

__host__ __device__ constexpr size_t sb_lds_bytes(int n) {
    const int NB = (n + SB - 1) / SB;
    return sizeof(double) * ((size_t)(NB * (NB + 1) / 2) * SBLK + (size_t)NB * SB) + 16;
}

// f64 DPP (gfx90a+ "DP ALU DPP": VOP1 / VOP2 f64 ops take row_newbcast:k -- every lane of a 16-lane row reads lane k of
// its row).  v_fmac_f64_dpp folds the broadcast into the multiply-add: ONE instruction where v_readlane needs two scalar
// reads, their hazards and the fma.  The leading s_nop 1 covers the VALU-write -> DPP-read hazard (2 wait states); the
// compiler's hazard recogniser does not look inside inline assembly.
template <int K>
__device__ __forceinline__ double sb_bcast(double v) {  // lane K of the row, to every lane of the row
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
template <int K, bool NOP>
__device__ __forceinline__ void sb_fmac_bcast(double& acc, double bsrc, double own) {  // acc += bcast_K(bsrc) * own
    if (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(own), "n"(K));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(own), "n"(K));
}

template <int J, int K>
struct SbRank1 {  // a[k] -= l_ij l_kj for k = K .. 15
    static __device__ __forceinline__ void run(double (&a)[16], double lij, double nl) {
        sb_fmac_bcast<K, K == J + 1>(a[K], lij, nl);
        SbRank1<J, K + 1>::run(a, lij, nl);
    }
};
template <int J>
struct SbRank1<J, 16> {
    static __device__ __forceinline__ void run(double (&)[16], double, double) {}
};

template <int J>
struct SbColumn {
    static __device__ __forceinline__ void run(double (&a)[16], double& myr, bool& bad, int row) {
        const double d = sb_bcast<J>(a[J]);
        bad |= !(d > 0);  // (the same in every lane of a row; off the dependency chain: a bad pivot poisons the block with
                          // NaNs, chol_ok = 0 makes the LM control reject the step without reading the result)
        // 1 / sqrt(d): v_rsq_f64 (~2^-26) + one Newton step r0 (1.5 - (d / 2) r0^2): three dependent operations instead of
        // the library's eight (range checks + a third-order step); the pivots of a damped normal matrix are normal numbers
        const double r0 = __builtin_amdgcn_rsq(d);
        const double r = r0 * fma(-0.5 * d, r0 * r0, 1.5);
        const double lij = a[J] * r;  // lane J holds d itself: sqrt(d) on the diagonal, a_ij / sqrt(d) below
        a[J] = lij;
        myr = (row == J) ? r : myr;
        SbRank1<J, J + 1>::run(a, lij, -lij);  // meaningful for k <= row (the lower triangle); the rest is never read
        SbColumn<J + 1>::run(a, myr, bad, row);
    }
};
template <>
struct SbColumn<16> {
    static __device__ __forceinline__ void run(double (&)[16], double&, bool&, int) {}
};

template <int K>
struct SbInvStep {  // row K of X = L^-1 is final: X[i][c] -= (L_iK / L_ii) X[K][c] for the rows i > K (this lane's four columns)
    static __device__ __forceinline__ void run(double (&x)[4], const double (&nlp)[16]) {
        sb_fmac_bcast<K, true>(x[0], x[0], nlp[K]);
        sb_fmac_bcast<K, false>(x[1], x[1], nlp[K]);
        sb_fmac_bcast<K, false>(x[2], x[2], nlp[K]);
        sb_fmac_bcast<K, false>(x[3], x[3], nlp[K]);
        SbInvStep<K + 1>::run(x, nlp);
    }
};
template <>
struct SbInvStep<15> {
    static __device__ __forceinline__ void run(double (&)[4], const double (&)[16]) {}
};

// wave 0: Cholesky of one diagonal block out of registers (lane i holds row i & 15: the four 16-lane rows of the wave
// carry the same block), then its INVERSE, which replaces the block in LDS: with L_kk^-1 at hand the panel A L_kk^-T and
// both substitutions with L_kk are products (matrix cores / independent dot products) instead of sixteen-step recurrences
// in every panel row.  Factor: sixteen unrolled column steps; the pivot broadcast and the rank-1 updates are f64 DPP
// instructions (v_readlane pairs + fma: 4.6 k cycles per block; a 32-bit DPP mov pair per value, the first attempt: 11.5 k).
// Inverse: forward substitution on the identity, rows = lanes; each of the wave's four 16-lane rows takes four columns, so a
// step is four v_fmac_f64_dpp and the whole inverse ~100 instructions.
__device__ __forceinline__ bool sb_factor_diag(double* Dk, int lane) {
    const int row = lane & 15, g = lane >> 4;
    double a[SB];
#pragma unroll
    for (int k = 0; k < SB; ++k) a[k] = Dk[row * SP + k];
    bool bad = false;
    double myr = 1.0;  // 1 / L_jj of this lane's own column, picked up branch-free as the columns go by
    SbColumn<0>::run(a, myr, bad, row);
    // X = L^-1:  X[i][c] = delta_ic / L_ii - sum_{k < i} (L_ik / L_ii) X[k][c]; this lane holds X[row][4 g .. 4 g + 3]
    double nlp[SB];
#pragma unroll
    for (int k = 0; k < SB; ++k) nlp[k] = (k < row) ? -(a[k] * myr) : 0.0;
    double x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = (row == 4 * g + j) ? myr : 0.0;
    SbInvStep<0>::run(x, nlp);
    // (the block is read by nobody else until the workgroup barrier that follows)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) Dk[row * SP + 4 * g + j] = x[j];
    return !bad;
}

// One whole 16 x 16 block  A[I][J] -= P_I P_J^T  by ONE wave on the matrix cores: four v_mfma_f64_16x16x4f64.  The f64 matrix
// peak equals the f64 vector peak on this chip, so this buys no throughput -- it buys instructions: 4 MFMAs + 16 LDS accesses
// per lane instead of 16 lanes x (64 fma + 64 LDS reads) for the same block, and the factorisation is a latency chain.
// Operand layout (tools/micro/mfma_f64_layout.hip): A: lane l supplies A[l % 16][l / 16]; B: lane l supplies B[l / 16][l % 16];
// D: lane l, register q holds D[l / 16 + 4 q][l % 16].
typedef double sb_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sb_block_update_mfma(double* A, int I, int J, int kb, int lane) {
    const double* PI = A + sb_off(I, kb);
    const double* PJ = A + sb_off(J, kb);
    double* T = A + sb_off(I, J);
    const int lr = lane & 15, lg = lane >> 4;
    sb_d4 c;
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = T[(lg + 4 * q) * SP + lr];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double a = -PI[lr * SP + 4 * s + lg];
        const double bb = PJ[lr * SP + 4 * s + lg];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) T[(lg + 4 * q) * SP + lr] = c[q];
}

// panel block  A[I][kb] <- A[I][kb] L_kk^-T = A[I][kb] Linv^T  (Linv = the inverted diagonal block, in place of A[kb][kb])
__device__ __forceinline__ void sb_panel_mfma(double* A, int I, int kb, int lane) {
    double* P = A + sb_off(I, kb);
    const double* Li = A + sb_off(kb, kb);
    const int lr = lane & 15, lg = lane >> 4;
    double pa[4], pb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        pa[s] = P[lr * SP + 4 * s + lg];    // A operand: A[i = lr][c = 4 s + lg]
        pb[s] = Li[lr * SP + 4 * s + lg];   // B operand: B[c][j = lr] = Linv[j][c]
    }
    sb_d4 c = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[s], pb[s], c, 0, 0, 0);
    // (every lane has read its operands of this block before anyone overwrites it: one wave owns the block)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) P[(lg + 4 * q) * SP + lr] = c[q];
}

#ifndef CS_SOLVE_PRIO
#define CS_SOLVE_PRIO CS_BA_PRIO  // A/B: the solver is ONE workgroup on the LM step's critical path
#endif
#include "ba_cholflow_dev.h"

// memory flavour of the solver's (and the packed LM-step functions') traffic: plain when a kernel boundary separates producer
// and consumer, relaxed agent-scope atomics (sc1: never served from a stale L1 / non-coherent L2 line) when both run inside
// ONE launch (the dataflow Cholesky, ba_cholflow_dev.h)
template <bool COH>
__device__ __forceinline__ double ldm(const double* p) {
    if (COH) return cf_ld(p);
    return *p;
}
template <bool COH>
__device__ __forceinline__ void stm(double* p, double v) {
    if (COH)
        cf_st(p, v);
    else
        *p = v;
}

// NW waves (16: k_solve_blocked's own workgroup; 8: inside the persistent LM kernel); n > 0; `skip` = the LM state says there
// is nothing to do (evaluated by the caller, consumed behind the matrix loads); *okFlagP (shared) = 1 iff every pivot was
// positive.  Which wave takes which block does not enter the arithmetic: the result is the same for every NW.
template <int NW, bool COH>
__device__ __forceinline__ void sb_solve_body(const BaDev& D, double* sm, int* okFlagP, const bool skip) {
    const int n = D.n, tid = threadIdx.x;
    constexpr int NT = NW * 64;
    const int NB = (n + SB - 1) / SB, NBT = NB * (NB + 1) / 2;
#ifdef CS_SOLVE_PROBE
    unsigned long long pT0 = __builtin_amdgcn_s_memtime(), pLoad = 0, pDiag = 0, pPanel = 0, pTrail = 0, pBack = 0, pT = 0;
#define CS_PROBE(acc)                                      \
    do {                                                   \
        unsigned long long _t = __builtin_amdgcn_s_memtime(); \
        acc += _t - pT;                                    \
        pT = _t;                                           \
    } while (0)
#else
#define CS_PROBE(acc) ((void)0)
#endif
    double* A = sm;
    double* b = sm + (size_t)NBT * SBLK;  // the right-hand side: one more row of the matrix
    // ---- load the lower triangle (identity padding beyond n).  One wave per 16 x 16 block, four entries per lane (a lane
    // reads 32-byte row pieces, a wave 16 full rows of 128 bytes); the block -> (I, J) arithmetic is wave-uniform and runs
    // on the scalar unit, every load of a wave's blocks (up to 5 at order 176) is issued before the first LDS store.
    // (The first version derived (I, J) per element with a per-lane search loop: ~2 k instructions in front of the loads.)
    {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
        const int r = ln >> 2, c0 = (ln & 3) * 4;
        constexpr int MAXB = (66 + NW - 1) / NW;  // 66 blocks at order 176
        double v[MAXB][4];
#pragma unroll
        for (int q = 0; q < MAXB; ++q) {
            const int blk = wv + q * NW;
            if (blk < NBT) {
                int I = 0;
                while ((I + 1) * (I + 2) / 2 <= blk) ++I;
                const int J = blk - I * (I + 1) / 2;
                const int gi = I * SB + r, gj = J * SB + c0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[q][u] = (gi == gj + u) ? 1.0 : 0.0;
                    if (gi < n && gj + u < n) v[q][u] = ldm<COH>(&D.S[(size_t)gi * n + gj + u]);
                }
            }
        }
        if (skip) return;  // !BA_ACTIVE (uniform); tested here so that it is not one more dependent trip
#pragma unroll
        for (int q = 0; q < MAXB; ++q) {
            const int blk = wv + q * NW;
            if (blk < NBT) {
#pragma unroll
                for (int u = 0; u < 4; ++u) A[blk * SBLK + r * SP + c0 + u] = v[q][u];
            }
        }
    }
    for (int q = tid; q < NB * SB; q += NT) b[q] = (q < n) ? ldm<COH>(&D.rhs[q]) : 0.0;
    if (tid == 0) *okFlagP = 1;
    __syncthreads();
#ifdef CS_SOLVE_PROBE
    pT = __builtin_amdgcn_s_memtime();
    pLoad = pT - pT0;
#endif
    // the first diagonal block; every later one is factored by wave 0 NEXT TO the trailing update of the step before
    // (look-ahead), so the serial column chain -- the longest phase -- is off the critical path
    if (tid < 64) {
        if (!sb_factor_diag(A + sb_off(0, 0), tid) && tid == 0) *okFlagP = 0;
    }
    __syncthreads();
    CS_PROBE(pDiag);

    for (int kb = 0; kb < NB; ++kb) {
        // ---- panel: A[I][kb] <- A[I][kb] Linv^T, one block per wave on the matrix cores; the right-hand side's block kb
        // (one more row of the matrix) is sixteen dot products with the rows of Linv: y_c = sum_{j <= c} b_j Linv[c][j].
        const int m = NB - kb - 1;
        {
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
            for (int I = kb + 1 + wv; I < NB; I += NW) sb_panel_mfma(A, I, kb, ln);
            if (wv == NW - 1 && ln < SB) {
                const double* Li = A + sb_off(kb, kb) + ln * SP;
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < SB; ++j) acc = fma(b[kb * SB + j], Li[j], acc);  // Linv[c][j] = 0 for j > c
                b[kb * SB + ln] = acc;  // (in place: the sixteen lanes' reads are all issued before this store)
            }
        }
        __syncthreads();
        CS_PROBE(pPanel);
        if (m == 0) break;
        // ---- trailing update A[I][J] -= P_I P_J^T for kb < J <= I (one block per wave and turn, on the matrix cores) and
        // b_J -= b_kb P_J^T.  Wave 0 takes the next diagonal block and then factors it; waves 1..15 take everything else.
        if (tid < 64) {
            sb_block_update_mfma(A, kb + 1, kb + 1, kb, tid);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!sb_factor_diag(A + sb_off(kb + 1, kb + 1), tid) && tid == 0) *okFlagP = 0;
        } else {
            const int wv = __builtin_amdgcn_readfirstlane((tid >> 6) - 1), ln = tid & 63;  // 0..NW-2
            const int nBlk = m * (m + 1) / 2;
            for (int blk = 1 + wv; blk < nBlk; blk += NW - 1) {  // block 0 of the list is wave 0's (kb + 1, kb + 1)
                int bi = 0;
                while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;
                const int bj = blk - bi * (bi + 1) / 2;
                sb_block_update_mfma(A, kb + 1 + bi, kb + 1 + bj, kb, ln);
            }
            // the right-hand side row: one entry per thread of the waves that are done first
            for (int q = (tid - 64); q < m * SB; q += NT - 64) {
                const int J = kb + 1 + q / SB, c = q % SB;
                const double* PJ = A + sb_off(J, kb) + c * SP;
                const double* y = b + kb * SB;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < SB; ++k) acc = fma(y[k], PJ[k], acc);
                b[J * SB + c] -= acc;
            }
        }
        __syncthreads();
        CS_PROBE(pTrail);
    }
    // b now holds y = L^-1 rhs.  Back substitution L^T x = y, block by block from the bottom; the diagonal blocks hold
    // their inverses, so x_kb = Linv^T y_kb is sixteen independent dot products: x_c = sum_{j >= c} Linv[j][c] y_j.
    for (int kb = NB - 1; kb >= 0; --kb) {
        if (tid < SB) {
            const double* Li = A + sb_off(kb, kb);
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < SB; ++j) acc = fma(Li[j * SP + tid], b[kb * SB + j], acc);  // Linv[j][c] = 0 for j < c
            b[kb * SB + tid] = acc;  // (in place: see the panel)
        }
        __syncthreads();
        // b_J -= L_{kb,J}^T x_kb for every block column J < kb: one entry per thread
        for (int cc = tid; cc < kb * SB; cc += NT) {
            const int J = cc / SB, c = cc % SB;
            const double* blk = A + sb_off(kb, J);
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < SB; ++k) acc = fma(blk[k * SP + c], b[kb * SB + k], acc);
            b[J * SB + c] -= acc;
        }
        __syncthreads();
    }
    for (int q = tid; q < n; q += NT) stm<COH>(&D.rhs[q], b[q]);
#ifdef CS_SOLVE_PROBE
    CS_PROBE(pBack);
    if (tid == 0 && D.st->nIterTotal == 3)
        printf("k_solve_blocked n=%d NB=%d cycles: load %llu diag %llu panel %llu trail %llu back %llu total %llu\n", n, NB, pLoad,
               pDiag, pPanel, pTrail, pBack, __builtin_amdgcn_s_memtime() - pT0);
#endif
#undef CS_PROBE
}

// NW waves.  16 (1024 threads, 4 waves of 112 VGPRs per SIMD) is the fastest solve on an idle chip and needs a compute unit with
// NOTHING else on it -- next to the persistent tracker it waited 35-45 us per LM step for one to drain
// (profiles/r03_key_frame_interval_per_queue.txt); 8 (2 waves per SIMD) starts on any compute unit whose tracker workgroup
// leaves half the register file, 4 on nearly any.  Same arithmetic and bits for every NW (sb_solve_body).
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_solve_blocked(BaDev D) {
    if (CS_SOLVE_PRIO) __builtin_amdgcn_s_setprio(CS_SOLVE_PRIO);
    const int stAllDone = D.st->all_done, stInnerDone = D.st->inner_done;  // (consumed after the matrix loads are in flight)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int okFlag;
    if (D.n == 0) {
        if (threadIdx.x == 0) D.st->chol_ok = 1;
        return;
    }
    sb_solve_body<NW, false>(D, sm, &okFlag, stAllDone || stInnerDone);
    if (stAllDone || stInnerDone) return;
    if (threadIdx.x == 0) D.st->chol_ok = okFlag;  // (behind the body's last workgroup barrier)
}
static int sb_solve_waves(int n) { return n <= 64 ? 4 : 8; }
static void sb_launch_solve(hipStream_t stream, const BaDev& D) {
    const size_t lds = sb_lds_bytes(D.n);
    switch (sb_solve_waves(D.n)) {
        case 4: hipLaunchKernelGGL(k_solve_blocked<4>, dim3(1), dim3(256), lds, stream, D); break;
        case 8: hipLaunchKernelGGL(k_solve_blocked<8>, dim3(1), dim3(512), lds, stream, D); break;
        default: hipLaunchKernelGGL(k_solve_blocked<16>, dim3(1), dim3(1024), lds, stream, D); break;
    }
}

// ---- one WAVE, rows in REGISTERS: reduced camera systems of order n <= NMAX <= 36 -----------------------
// Lane i keeps row i of S in NMAX registers; both loops are fully unrolled, so every index is static and every
// cross-lane read is a v_readlane of a constant lane: no LDS, no barrier, one reciprocal square root per column.
// Rows/columns n..NMAX-1 are identity padding.
// solve_reg_combine then solve_reg_factor, both called by all 256 threads of a workgroup; on return
// Ssm[NMAX * NMAX + r] holds the solved camera step and *okSh the Cholesky status (both in LDS, visible to the whole
// workgroup).
template <int NMAX>
__device__ __forceinline__ void solve_reg_combine(const BaDev& D, double* Ssm) {
    // all four waves add the slice partials of k_schur_part (slice order) into the reduced system in LDS ...
    {
        const double lambda = D.st->lambda;
        const int nPairs = D.nc * (D.nc + 1) / 2;
        for (int w = threadIdx.x; w < nPairs * 42; w += 256) {
            const int pi = w / 42, q = w - 42 * pi;
            int a = 0, pr = pi;
            while (pr >= D.nc - a) {
                pr -= D.nc - a;
                ++a;
            }
            const int b = a + pr;
            const double* p = D.schurPart + (size_t)pi * D.nSlices * 72;
            int uq = -1;
            if (a == b) {
                if (q < 36) {
                    const int r = q / 6, c = q - 6 * r, rr = r < c ? r : c, cc = r < c ? c : r;
                    uq = rr * 6 - (rr * (rr - 1)) / 2 + (cc - rr);
                } else {
                    uq = 21 + (q - 36);
                }
            }
            // all slice loads of this entry are issued before the first is consumed (a loop with a run-time trip count
            // is not unrolled, and eight dependent round trips were 6.4 us of this 17 us kernel, s_memtime); the slices
            // are still added in slice order, the padding adds exact zeros
            double sv[16], uv[16];
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) {
                sv[sl] = (sl < D.nSlices) ? p[sl * 72 + q] : 0.0;
                uv[sl] = (sl < D.nSlices && uq >= 0) ? p[sl * 72 + 42 + uq] : 0.0;
            }
            double sSum = 0, uSum = 0;
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) {
                sSum += sv[sl];
                uSum += uv[sl];
            }
            if (q < 36) {
                const int r = q / 6, c = q - 6 * r;
                if (a == b) {
                    Ssm[(6 * a + r) * NMAX + 6 * a + c] = (uSum + ((r == c) ? lambda : 0.0)) - sSum;
                } else {
                    Ssm[(6 * a + r) * NMAX + 6 * b + c] = -sSum;
                    Ssm[(6 * b + c) * NMAX + 6 * a + r] = -sSum;
                }
            } else if (a == b) {
                Ssm[NMAX * NMAX + 6 * a + (q - 36)] = uSum - sSum;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ double rdlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int NMAX>
__device__ __forceinline__ void solve_reg_factor(const BaDev& D, double* Ssm, int* okSh) {
    const int n = D.n;
    if (threadIdx.x < 64) {
    // ... and wave 0 factorises it out of registers
    const int i = threadIdx.x;
    double a[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) a[k] = (i < n && k < n) ? Ssm[i * NMAX + k] : ((i == k) ? 1.0 : 0.0);
    double b = (i < n) ? Ssm[NMAX * NMAX + i] : 0.0;
    double rdiag[NMAX];  // 1 / L[j][j] (wave-uniform)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        const double cj = a[j];
        double d = rdlane_d(cj, j);
        if (!(d > 0)) {
            ok = false;
            d = 1.0;
        }
        // 1/sqrt(d): the hardware estimate + two Newton steps (full binary64 accuracy) is a ~12-instruction dependent
        // chain; IEEE sqrt followed by an IEEE division is ~50, and this chain is the critical path of every column
        double rd = __builtin_amdgcn_rsq(d);
        const double hd = 0.5 * d;
        rd = __builtin_fma(rd, __builtin_fma(-(hd * rd), rd, 0.5), rd);  // y += y (1/2 - (d/2) y^2)
        rd = __builtin_fma(rd, __builtin_fma(-(hd * rd), rd, 0.5), rd);
        rdiag[j] = rd;
        const double lij = (i > j) ? cj * rd : 0.0;
        a[j] = lij;
        const double yj = rdlane_d(b, j) * rd;
        b = (i == j) ? yj : __builtin_fma(-lij, yj, b);
        // (fused multiply-adds here and in the back substitution: this solve already differs from the oracle's in the
        // last bits -- reciprocal square roots instead of sqrt + division -- and the chain below is the longest
        // dependent instruction sequence of the LM step)
#pragma unroll
        for (int k = j + 1; k < NMAX; ++k) a[k] = __builtin_fma(-lij, rdlane_d(lij, k), a[k]);
    }
    // L^T x = y as a column sweep over the rows of L, which go back to LDS (row i by lane i) for the transposed reads
    if (i < NMAX) {
#pragma unroll
        for (int k = 0; k < NMAX; ++k) Ssm[i * NMAX + k] = a[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = NMAX - 1; j >= 0; --j) {
        const double xj = rdlane_d(b, j) * rdiag[j];
        if (i == j) b = xj;
        if (i < j) b = __builtin_fma(-Ssm[j * NMAX + i], xj, b);
    }
    if (i < n) Ssm[NMAX * NMAX + i] = b;
    if (i == 0) *okSh = ok ? 1 : 0;
    }
    __syncthreads();
}

// ---- tentative step + its cost ----------------------------------------------------------------------
// Wave per point: back-substituted point step, then (lane = measurement) the squared inlier residuals of that point
// at the tentative estimate.  The tentative pose of a measurement's camera is re-derived in the lane from the
// solved camera step (R exp(w), t + dt -- the same operations block 0 performs when it writes Rn / Tn for the
// commit), so the cost needs no second launch behind a grid-wide dependency.  Per-block partial costs go to
// costPart[blockIdx.x] (D.nUpdBlocks of them), summed in a fixed order by k_control.
template <int NMAX>  // > 0: the reduced system is combined and solved here, redundantly per workgroup (no solve launch)
__global__ __launch_bounds__(256) void k_update(BaDev D) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    __shared__ double red[4];
    __shared__ double Ssm[NMAX > 0 ? NMAX * NMAX + NMAX : 1];
    __shared__ int okSh;
    // The partials of the reduced system are the first thing on the critical path (combine -> factorisation) and the
    // last thing the previous launch wrote: their loads go out first, alone.
    if (NMAX > 0) solve_reg_combine<(NMAX > 0 ? NMAX : 1)>(D, Ssm);
    // Everything the point step and the tentative residuals read that does NOT depend on the solve is loaded now: the
    // loads fly under the factorisation below instead of starting a fresh miss chain behind it.
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool mine = gw < D.P && gw >= D.pLo && gw < D.pHi;
    int o0 = 0, o1 = 0, pj = -1, pout = 1;
    double pW[18], pR[9], pT[3], pK[9], pxy[2], pVi[9], pg[3], pM[3];
    if (mine) {
        o0 = D.obs_ptr[gw];
        o1 = D.obs_ptr[gw + 1];
#pragma unroll
        for (int q = 0; q < 9; ++q) pVi[q] = D.Vinv[9 * (size_t)gw + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            pg[q] = D.gp[3 * (size_t)gw + q];
            pM[q] = D.pts[3 * (size_t)gw + q];
        }
        const int o = o0 + lane;
        if (o < o1) {
            pj = D.obs_cam[o];
            pout = D.outlier[o];
#pragma unroll
            for (int q = 0; q < 18; ++q) pW[q] = D.W[18 * (size_t)o + q];
            pxy[0] = D.obs_xy[2 * (size_t)o];
            pxy[1] = D.obs_xy[2 * (size_t)o + 1];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                pR[q] = D.Rs[9 * pj + q];
                pK[q] = D.Ks[9 * pj + q];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) pT[q] = D.Ts[3 * pj + q];
        }
    }
    const double* rhs = D.rhs;
    if (NMAX > 0) {
        solve_reg_factor<(NMAX > 0 ? NMAX : 1)>(D, Ssm, &okSh);
        rhs = Ssm + NMAX * NMAX;
        if (blockIdx.x == 0) {
            if (threadIdx.x < D.n) D.rhs[threadIdx.x] = rhs[threadIdx.x];
            if (threadIdx.x == 0) D.st->chol_ok = okSh;
        }
    }
    double cost = 0;
    if (gw < D.P && !mine) {
        if (lane == 0) D.stepPart[gw] = 0;  // another rank's point
    } else if (mine) {
        const int i = gw;
        double b[3] = {0, 0, 0};
        if (i >= D.nPtsCon) {
            for (int o = o0 + lane; o < o1; o += 64) {
                const bool first = (o == o0 + lane);
                const int j = (first ? pj : D.obs_cam[o]) - D.nCamsCon;
                if (j < 0 || (first ? pout : D.outlier[o])) continue;
                const double* dc = rhs + 6 * j;
                if (first) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int r = 0; r < 6; ++r) b[c] -= pW[3 * r + c] * dc[r];
                } else {
                    const double* Wo = D.W + 18 * (size_t)o;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int r = 0; r < 6; ++r) b[c] -= Wo[3 * r + c] * dc[r];
                }
            }
        }
        b[0] = wsum(b[0]);
        b[1] = wsum(b[1]);
        b[2] = wsum(b[2]);
        double d[3] = {0, 0, 0};
        if (i >= D.nPtsCon) {
            const double g0 = pg[0] + b[0], g1 = pg[1] + b[1], g2 = pg[2] + b[2];
#pragma unroll
            for (int r = 0; r < 3; ++r) d[r] = pVi[3 * r] * g0 + pVi[3 * r + 1] * g1 + pVi[3 * r + 2] * g2;
        }
        double Mn[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) Mn[r] = pM[r] + d[r];
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) D.Mn[3 * (size_t)i + r] = Mn[r];
            D.stepPart[i] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        }
        for (int o = o0 + lane; o < o1; o += 64) {
            const bool first = (o == o0 + lane);
            if (first ? pout : D.outlier[o]) continue;
            const int j = first ? pj : D.obs_cam[o];
            double Rc[9], Tc[3], Kc[9], xy[2];
            if (first) {
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    Rc[q] = pR[q];
                    Kc[q] = pK[q];
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) Tc[q] = pT[q];
                xy[0] = pxy[0];
                xy[1] = pxy[1];
            } else {
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    Rc[q] = D.Rs[9 * j + q];
                    Kc[q] = D.Ks[9 * j + q];
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) Tc[q] = D.Ts[3 * j + q];
                xy[0] = D.obs_xy[2 * (size_t)o];
                xy[1] = D.obs_xy[2 * (size_t)o + 1];
            }
            double Rn[9], Tn[3];
            if (j >= D.nCamsCon) {
                const double* dc = rhs + 6 * (j - D.nCamsCon);
                double w[3] = {dc[0], dc[1], dc[2]}, dR[9];
                so3_exp(w, dR);
                mat33AB(Rc, dR, Rn);
#pragma unroll
                for (int q = 0; q < 3; ++q) Tn[q] = Tc[q] + dc[3 + q];
            } else {
#pragma unroll
                for (int q = 0; q < 9; ++q) Rn[q] = Rc[q];
#pragma unroll
                for (int q = 0; q < 3; ++q) Tn[q] = Tc[q];
            }
            double e[2];
            residual<false>(Kc, Rn, Tn, Mn, xy, e, nullptr, nullptr);
            cost += e[0] * e[0] + e[1] * e[1];
        }
    }
    cost = wsum(cost);
    if (lane == 0) red[threadIdx.x >> 6] = cost;
    __syncthreads();
    if (threadIdx.x == 0) D.costPart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < D.C) {
        const int j = t;
        double s2 = 0;
        if (j >= D.nCamsCon) {
            const double* dc = rhs + 6 * (j - D.nCamsCon);
            double w[3] = {dc[0], dc[1], dc[2]}, dR[9], Rn[9];
            so3_exp(w, dR);
            mat33AB(D.Rs + 9 * j, dR, Rn);
#pragma unroll
            for (int q = 0; q < 9; ++q) D.Rn[9 * j + q] = Rn[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) D.Tn[3 * j + q] = D.Ts[3 * j + q] + dc[3 + q];
#pragma unroll
            for (int q = 0; q < 6; ++q) s2 += dc[q] * dc[q];
        } else {
#pragma unroll
            for (int q = 0; q < 9; ++q) D.Rn[9 * j + q] = D.Rs[9 * j + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) D.Tn[3 * j + q] = D.Ts[3 * j + q];
        }
        D.stepPart[D.P + j] = s2;
    }
}

// ---- tentative step + its cost, eight lanes per point (see k_linearize_seg8): 32 points per workgroup, so the redundant
// combine + factorisation runs in 16 workgroups instead of 125 for the 500-point local BA, and the ~300 f64 instructions of
// a measurement's tentative residual run with 40 of 64 lanes busy instead of 5.
template <int NMAX>
__global__ __launch_bounds__(256) void k_update_seg8(BaDev D) {
    CS_BA_SETPRIO();
    if (!BA_ACTIVE(D)) return;
    __shared__ double red[4];
    __shared__ double Ssm[NMAX * NMAX + NMAX];
    __shared__ int okSh;
    solve_reg_combine<NMAX>(D, Ssm);
    const int lane = threadIdx.x & 63, k = threadIdx.x & 7;
    const int gw = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool mine = gw < D.P && gw >= D.pLo && gw < D.pHi;
    int o = -1, pj = 0, pout = 1;
    double pW[18], pR[9], pT[3], pK[9], pxy[2] = {0, 0}, pVi[9], pg[3], pM[3];
#pragma unroll
    for (int q = 0; q < 18; ++q) pW[q] = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) pR[q] = pK[q] = pVi[q] = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) pT[q] = pg[q] = pM[q] = 0;
    if (mine) {
        const int o0 = D.obs_ptr[gw], o1 = D.obs_ptr[gw + 1];
#pragma unroll
        for (int q = 0; q < 9; ++q) pVi[q] = D.Vinv[9 * (size_t)gw + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            pg[q] = D.gp[3 * (size_t)gw + q];
            pM[q] = D.pts[3 * (size_t)gw + q];
        }
        if (o0 + k < o1) {
            o = o0 + k;
            pj = D.obs_cam[o];
            pout = D.outlier[o];
#pragma unroll
            for (int q = 0; q < 18; ++q) pW[q] = D.W[18 * (size_t)o + q];
            pxy[0] = D.obs_xy[2 * (size_t)o];
            pxy[1] = D.obs_xy[2 * (size_t)o + 1];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                pR[q] = D.Rs[9 * pj + q];
                pK[q] = D.Ks[9 * pj + q];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) pT[q] = D.Ts[3 * pj + q];
        }
    }
    solve_reg_factor<NMAX>(D, Ssm, &okSh);
    const double* rhs = Ssm + NMAX * NMAX;
    if (blockIdx.x == 0) {
        if (threadIdx.x < D.n) D.rhs[threadIdx.x] = rhs[threadIdx.x];
        if (threadIdx.x == 0) D.st->chol_ok = okSh;
    }
    double cost = 0;
    if (gw < D.P && !mine) {
        if (k == 0) D.stepPart[gw] = 0;  // another rank's point
    } else if (mine) {
        const int i = gw;
        const bool inl = (o >= 0) && !pout;
        double b[3] = {0, 0, 0};
        if (i >= D.nPtsCon && inl && pj >= D.nCamsCon) {
            const double* dc = rhs + 6 * (pj - D.nCamsCon);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 6; ++r) b[c] -= pW[3 * r + c] * dc[r];
        }
        b[0] = seg8_sum(b[0]);
        b[1] = seg8_sum(b[1]);
        b[2] = seg8_sum(b[2]);
        double d[3] = {0, 0, 0};
        if (i >= D.nPtsCon) {
            const double g0 = pg[0] + b[0], g1 = pg[1] + b[1], g2 = pg[2] + b[2];
#pragma unroll
            for (int r = 0; r < 3; ++r) d[r] = pVi[3 * r] * g0 + pVi[3 * r + 1] * g1 + pVi[3 * r + 2] * g2;
        }
        double Mn[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) Mn[r] = pM[r] + d[r];
        if (k == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) D.Mn[3 * (size_t)i + r] = Mn[r];
            D.stepPart[i] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        }
        if (inl) {
            double Rn[9], Tn[3];
            if (pj >= D.nCamsCon) {
                const double* dc = rhs + 6 * (pj - D.nCamsCon);
                double w[3] = {dc[0], dc[1], dc[2]}, dR[9];
                so3_exp(w, dR);
                mat33AB(pR, dR, Rn);
#pragma unroll
                for (int q = 0; q < 3; ++q) Tn[q] = pT[q] + dc[3 + q];
            } else {
#pragma unroll
                for (int q = 0; q < 9; ++q) Rn[q] = pR[q];
#pragma unroll
                for (int q = 0; q < 3; ++q) Tn[q] = pT[q];
            }
            double e[2];
            residual<false>(pK, Rn, Tn, Mn, pxy, e, nullptr, nullptr);
            cost = e[0] * e[0] + e[1] * e[1];
        }
    }
    cost = wsum(cost);
    if (lane == 0) red[threadIdx.x >> 6] = cost;
    __syncthreads();
    if (threadIdx.x == 0) D.costPart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < D.C) {
        const int j = t;
        double s2 = 0;
        if (j >= D.nCamsCon) {
            const double* dc = rhs + 6 * (j - D.nCamsCon);
            double w[3] = {dc[0], dc[1], dc[2]}, dR[9], Rn[9];
            so3_exp(w, dR);
            mat33AB(D.Rs + 9 * j, dR, Rn);
#pragma unroll
            for (int q = 0; q < 9; ++q) D.Rn[9 * j + q] = Rn[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) D.Tn[3 * j + q] = D.Ts[3 * j + q] + dc[3 + q];
#pragma unroll
            for (int q = 0; q < 6; ++q) s2 += dc[q] * dc[q];
        } else {
#pragma unroll
            for (int q = 0; q < 9; ++q) D.Rn[9 * j + q] = D.Rs[9 * j + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) D.Tn[3 * j + q] = D.Ts[3 * j + q];
        }
        D.stepPart[D.P + j] = s2;
    }
}

// ---- cost at the tentative (which=1) or current (which=0) estimate ------------------------------------
__global__ __launch_bounds__(256) void k_cost(BaDev D, int which) {
    if (D.st->all_done) return;
    if (which == 1 && D.st->inner_done) return;
    __shared__ double red[4];
    const double* Rs = which ? D.Rn : D.Rs;
    const double* Ts = which ? D.Tn : D.Ts;
    const double* pts = which ? D.Mn : D.pts;
    double c = 0;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < D.nObs; o += gridDim.x * 256) {
        if (D.outlier[o]) continue;
        const int j = D.obs_cam[o], i = D.obs_pt[o];
        if (i < D.pLo || i >= D.pHi) continue;
        double e[2];
        residual<false>(D.Ks + 9 * j, Rs + 9 * j, Ts + 3 * j, pts + 3 * (size_t)i, D.obs_xy + 2 * (size_t)o, e, nullptr,
                        nullptr);
        c += e[0] * e[0] + e[1] * e[1];
    }
    c = wsum(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) D.costPart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- LM control: one workgroup -----------------------------------------------------------------------
// phase 0: start of an LM run (cost of the current estimate over the current inliers)
// phase 1: after a tentative step (accept / reject, commit, stop tests)
__global__ __launch_bounds__(256) void k_control(BaDev D) {  // phase 0
    __shared__ double red[4];
    BaState* st = D.st;
    if (st->all_done) return;
    const int tid = threadIdx.x;
    // fixed-order sum of k_cost's partials
    double c = 0;
    for (int q = tid; q < D.nCostBlocks; q += 256) c += D.costPart[q];
    c = wsum(c);
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    const double cost_sum = ((red[0] + red[1]) + red[2]) + red[3];
    if (tid == 0) {
        st->cost = cost_sum;
        st->lambda = 1e-3;
        st->inner_it = 0;
        st->inner_done = (D.innerMaxIter <= 0) ? 1 : 0;
        st->changed = 0;  // k_outer_begin's job, for the k_flag at the end of this round
        st->nOutliers = 0;
        st->pending = 0;
        st->cur = 0;
        if (st->first_cost) {
            st->cost0 = cost_sum;
            st->first_cost = 0;
        }
    }
}

// phase 1: after a tentative step (accept / reject, commit, stop tests).  Everything the launch reads -- the state word,
// both partial lists and the tentative values each thread would commit -- is requested up front, so the launch is one
// round trip to memory, the decision, and the stores; the sums keep k_control's order.
__global__ __launch_bounds__(256) void k_control_step(BaDev D) {
    CS_BA_SETPRIO();
    __shared__ double red[8];
    __shared__ int accept;
    BaState* st = D.st;
    const int tid = threadIdx.x;
    const int all_done = st->all_done, inner_done = st->inner_done, chol_ok = st->chol_ok, inner_it = st->inner_it;
    const double cost_old = st->cost, lambda = st->lambda;
    // both partial lists in one batch of loads per thread (a loop with a run-time trip count would make the second
    // trip wait for the first); same order of additions as before
    double cv[2], sv[8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = tid + 256 * k;
        cv[k] = (q < D.nUpdBlocks) ? D.costPart[q] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = tid + 256 * k;
        sv[k] = (q < D.P + D.C) ? D.stepPart[q] : 0.0;
    }
    constexpr int PRE = 8;
    const int nR = 9 * D.C, nT = 3 * D.C, total = nR + nT + 3 * D.P;
    double pre[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        const int q = tid + 256 * k;
        pre[k] = (q < nR) ? D.Rn[q] : (q < nR + nT) ? D.Tn[q - nR] : (q < total) ? D.Mn[q - nR - nT] : 0.0;
    }
    if (all_done || inner_done) return;
    double c = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) c += cv[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) s2 += sv[k];
    for (int q = tid + 512; q < D.nUpdBlocks; q += 256) c += D.costPart[q];
    for (int q = tid + 2048; q < D.P + D.C; q += 256) s2 += D.stepPart[q];
    c = wsum(c);
    s2 = wsum(s2);
    if ((tid & 63) == 0) {
        red[tid >> 6] = c;
        red[4 + (tid >> 6)] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        const double cost_sum = ((red[0] + red[1]) + red[2]) + red[3];
        const double step2 = ((red[4] + red[5]) + red[6]) + red[7];
        const double cost_new = chol_ok ? cost_sum : 1e300;
        int acc = (chol_ok && cost_new <= cost_old) ? 1 : 0;
        int done = 0;
        st->nIterTotal += 1;
        st->inner_it = inner_it + 1;
        if (!chol_ok) st->nCholFail += 1;
        if (acc) st->nAccepted += 1;
        if (acc) {
            const double dec = cost_old - cost_new;
            st->cost = cost_new;
            st->lambda = lambda / 10;
            if (dec < 1e-9 * cost_new + 1e-15 || step2 < 1e-20) done = 1;
        } else {
            st->lambda = lambda * 10;
            if (lambda * 10 > 1e12) done = 1;
        }
        if (inner_it + 1 >= D.innerMaxIter) done = 1;
        st->inner_done = done;
        accept = acc;
    }
    __syncthreads();
    if (accept) {
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int q = tid + 256 * k;
            if (q < nR)
                D.Rs[q] = pre[k];
            else if (q < nR + nT)
                D.Ts[q - nR] = pre[k];
            else if (q < total)
                D.pts[q - nR - nT] = pre[k];
        }
        for (int q = tid + 256 * PRE; q < total; q += 256) {  // (more than 170 cameras: the tail still holds R / t entries)
            if (q < nR)
                D.Rs[q] = D.Rn[q];
            else if (q < nR + nT)
                D.Ts[q - nR] = D.Tn[q - nR];
            else
                D.pts[q - nR - nT] = D.Mn[q - nR - nT];
        }
    }
}

#include "ba_packed_dev.h"
#include "ba_window_dev.h"
#include "small_ops.h"
#include "ba_output_dev.h"
#include "ba_intercam_dev.h"

// ---- outer loop: outlier flags ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flag(BaDev D) {
    BaState* st = D.st;
    if (st->all_done) return;
    const double thr2 = D.maxErr * D.maxErr;
    int changed = 0, nout = 0;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < D.nObs; o += gridDim.x * 256) {
        const int j = D.obs_cam[o], i = D.obs_pt[o];
        if (i < D.pLo || i >= D.pHi) continue;
        double e[2];
        residual<false>(D.Ks + 9 * j, D.Rs + 9 * j, D.Ts + 3 * j, D.pts + 3 * (size_t)i, D.obs_xy + 2 * (size_t)o, e,
                        nullptr, nullptr);
        const int out = (e[0] * e[0] + e[1] * e[1] > thr2) ? 1 : 0;
        if (out != D.outlier[o]) changed = 1;
        D.outlier[o] = out;
        nout += out;
    }
    if (changed) atomicOr(&st->changed, 1);
    if (nout) atomicAdd(&st->nOutliers, nout);
}

// start of a solve: LM state + every measurement an inlier
struct BaInitCopy {  // initial estimate to copy into the workspace (cs_ba_solve_dev); src null = already in place
    const double *R0, *T0, *M0;
    double *R, *T, *M;
    int nR, nT, nM;
};
__global__ __launch_bounds__(256) void k_init_state(BaState* st, int* outlier, int nObs, BaInitCopy I) {
    const int t0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int o = t0; o < nObs; o += stride) outlier[o] = 0;
    if (I.R0) {
        for (int q = t0; q < I.nR; q += stride) I.R[q] = I.R0[q];
        for (int q = t0; q < I.nT; q += stride) I.T[q] = I.T0[q];
        for (int q = t0; q < I.nM; q += stride) I.M[q] = I.M0[q];
    }
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    BaState z;
    z.lambda = 1e-3;
    z.cost = z.cost_new = z.step2 = z.cost0 = 0;
    z.inner_it = z.inner_done = z.all_done = z.chol_ok = z.changed = z.nIterTotal = z.nOuter = z.nOutliers = 0;
    z.nCholFail = z.nAccepted = z.solverTimeout = 0;
    z.pending = z.cur = z.seq = z.fuseEpoch = 0;
    z.first_cost = 1;
    *st = z;
}

__global__ void k_outer_begin(BaDev D) {
    BaState* st = D.st;
    if (st->all_done) return;
    st->changed = 0;
    st->nOutliers = 0;
}

__global__ void k_outer_end(BaDev D) {
    BaState* st = D.st;
    if (st->all_done) return;
    st->nOuter += 1;
    st->inner_done = 0;
    if (!st->changed) st->all_done = 1;
    ba_publish_state(D, 0, st->all_done);
}

__global__ void k_finish(BaDev D, cs_ba_stats_dev* out) {
    // final cost over the final inlier set was computed by k_cost(0) + costPart
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double c = 0;
    for (int q = tid; q < D.nCostBlocks; q += 256) c += D.costPart[q];
    c = wsum(c);
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        out->cost0 = D.st->cost0;
        out->cost = ((red[0] + red[1]) + red[2]) + red[3];
        out->nIterTotal = D.st->nIterTotal;
        out->nOuter = D.st->nOuter;
        out->nOutliers = D.st->nOutliers;
        int fl = 0;
        if (D.st->nCholFail > 0) fl |= 1;                               // CS_BA_FLAG_CHOL_FAILED
        if (D.st->nCholFail > 0 && D.st->nAccepted == 0) fl |= 2;       // CS_BA_FLAG_NO_PROGRESS
        if (D.st->solverTimeout) fl |= 4;                               // CS_BA_FLAG_SOLVER_TIMEOUT
        out->flags = fl;
    }
}

__global__ void k_cost_force(BaDev D) {  // k_cost(0) ignoring the stop flags (final report)
    __shared__ double red[4];
    double c = 0;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < D.nObs; o += gridDim.x * 256) {
        if (D.outlier[o]) continue;
        const int j = D.obs_cam[o], i = D.obs_pt[o];
        if (i < D.pLo || i >= D.pHi) continue;
        double e[2];
        residual<false>(D.Ks + 9 * j, D.Rs + 9 * j, D.Ts + 3 * j, D.pts + 3 * (size_t)i, D.obs_xy + 2 * (size_t)o, e,
                        nullptr, nullptr);
        c += e[0] * e[0] + e[1] * e[1];
    }
    c = wsum(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) D.costPart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- index building on the device -----------------------------------------------------------------
__global__ void k_build_obs_pt(int P, const int* obs_ptr, int* obs_pt) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= P) return;
    for (int o = obs_ptr[i] + lane; o < obs_ptr[i + 1]; o += 64) obs_pt[o] = i;
}

__global__ void k_build_obs_of(int nObs, int C, const int* obs_pt, const int* obs_cam, int* obs_of) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o < nObs) obs_of[(size_t)obs_pt[o] * C + obs_cam[o]] = o;  // duplicates: last writer wins (host rejects them)
}


// ---- distributed solve (collective 2): points sliced by rank -------------------------------------
// Every rank holds the whole problem (the per-frame all-gather replicated measurements and poses) and owns the points
// [pLo, pHi).  Per LM step a rank linearises its points, forms ITS part of the reduced camera system (the dense
// S || rhs, lambda added by rank 0 only), the host all-reduces S || rhs (RCCL over xGMI), every rank solves the same
// system, steps its own points and all cameras, and the partial tentative costs are all-reduced as four scalars.
// The LM / outlier control flow stays on the device exactly as in the single-process solve (every rank takes the
// same decisions from the same all-reduced numbers), so the host still never synchronises: it only interleaves the
// phase launches below with collectives on the same stream.
// scal[0] cost partial, scal[1] squared point step partial, scal[2] flags changed, scal[3] outlier count
__global__ __launch_bounds__(256) void k_dist_pack(BaDev D, int what) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    if (what == 2) {  // after k_flag
        if (tid == 0) {
            D.scal[2] = (double)D.st->changed;
            D.scal[3] = (double)D.st->nOutliers;
        }
        return;
    }
    double c = 0;
    const int nPart = (what == 1) ? D.nUpdBlocks : D.nCostBlocks;
    for (int q = tid; q < nPart; q += 256) c += D.costPart[q];
    c = wsum(c);
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    const double cost_sum = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    double s2 = 0;
    if (what == 1)
        for (int q = D.pLo + tid; q < D.pHi; q += 256) s2 += D.stepPart[q];
    s2 = wsum(s2);
    if ((tid & 63) == 0) red[tid >> 6] = s2;
    __syncthreads();
    if (tid == 0) {
        D.scal[0] = cost_sum;
        D.scal[1] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

// k_control with the all-reduced scalars; commits all cameras and this rank's points
__global__ __launch_bounds__(256) void k_dist_control(BaDev D, int phase) {
    __shared__ int accept;
    BaState* st = D.st;
    if (st->all_done) return;
    if (phase == 1 && st->inner_done) return;
    const int tid = threadIdx.x;
    const double cost_sum = D.scal[0];
    if (phase == 0) {
        if (tid == 0) {
            st->cost = cost_sum;
            st->lambda = 1e-3;
            st->inner_it = 0;
            st->inner_done = (D.innerMaxIter <= 0) ? 1 : 0;
            st->changed = 0;  // k_outer_begin's job, for the k_flag at the end of this round
            st->nOutliers = 0;
            if (st->first_cost) {
                st->cost0 = cost_sum;
                st->first_cost = 0;
            }
        }
        return;
    }
    if (tid == 0) {
        double step2 = D.scal[1];
        for (int j = 0; j < D.C; ++j) step2 += D.stepPart[D.P + j];  // camera steps: identical on every rank
        const double cost_new = st->chol_ok ? cost_sum : 1e300;
        int acc = (st->chol_ok && cost_new <= st->cost) ? 1 : 0;
        int done = 0;
        st->nIterTotal += 1;
        st->inner_it += 1;
        if (!st->chol_ok) st->nCholFail += 1;
        if (acc) st->nAccepted += 1;
        if (acc) {
            const double dec = st->cost - cost_new;
            st->cost = cost_new;
            st->lambda /= 10;
            if (dec < 1e-9 * cost_new + 1e-15 || step2 < 1e-20) done = 1;
        } else {
            st->lambda *= 10;
            if (st->lambda > 1e12) done = 1;
        }
        if (st->inner_it >= D.innerMaxIter) done = 1;
        st->inner_done = done;
        accept = acc;
    }
    __syncthreads();
    if (accept) {
        for (int q = tid; q < 9 * D.C; q += 256) D.Rs[q] = D.Rn[q];
        for (int q = tid; q < 3 * D.C; q += 256) D.Ts[q] = D.Tn[q];
        for (int q = 3 * D.pLo + tid; q < 3 * D.pHi; q += 256) D.pts[q] = D.Mn[q];
    }
}

__global__ void k_dist_outer_end(BaDev D) {
    BaState* st = D.st;
    if (st->all_done) return;
    st->changed = (D.scal[2] > 0.0) ? 1 : 0;
    st->nOutliers = (int)(D.scal[3] + 0.5);
    st->nOuter += 1;
    st->inner_done = 0;
    if (!st->changed) st->all_done = 1;
}

// before the final all-reduce (sum) of points and flags: entries owned by other ranks contribute zero
__global__ void k_dist_zero_foreign(BaDev D) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 3 * D.P) {
        const int i = t / 3;
        if (i < D.pLo || i >= D.pHi) D.pts[t] = 0.0;
    }
    if (t < D.nObs) {
        const int i = D.obs_pt[t];
        if (i < D.pLo || i >= D.pHi) D.outlier[t] = 0;
    }
}

}  // namespace

// =====================================================================================================
// launch geometry of one solve (host side)
struct BaPlan {
    bool seg8;  // eight lanes per point in the linearisation and the update (no point has more than 8 measurements)
    BaDev D;
    int cb, gPts, gUpd, nPairs;
    bool sliced;
    bool cholFlow;     // orders beyond the LDS solver: the one-launch dataflow Cholesky (ba_cholflow_dev.h)
    CholFlow F;
    bool packed;       // orders 37..176 with pair lists, no point with more than 64 measurements: ba_packed_dev.h
    int gPack;
    BaDev DB;          // packed path: the same launch arguments with the two LM state words swapped
    bool syrk;         // large orders without pair lists: the Schur sum as Z Z^T on the f64 matrix cores (ba_syrk_dev.h)
    SyrkDev Y;
    size_t syrkZtBytes;
};

struct cs_ba {
    int device;
    int capC, capP, capObs;
    hipStream_t own_stream;
    bool ownStreamIsOurs;  // false after cs_ba_set_stream: the caller's stream, not destroyed with the workspace
    // device buffers
    double *Ks, *Rs, *Ts, *pts, *Rn, *Tn, *Mn, *obs_xy, *Jc, *e, *W, *Vinv, *gp, *S, *rhs, *costPart, *stepPart, *schurPart, *scal, *Y;
    BaState* st2;
    int *obs_ptr, *obs_cam, *obs_pt, *cam_ptr, *cam_obs, *obs_of, *outlier;
    BaState* st;
    cs_ba_stats_dev* stats;
    BaPlan* dist;  // plan of the distributed solve in progress (cs_ba_dist_begin)
    // host-pointer entry: every input lives in ONE device block (io) mirrored by one pinned block (h_io), so a solve
    // is one H2D copy; Rs | Ts | pts sit next to each other in it (one D2H); statistics | outlier flags form the
    // second block (ob / h_ob, one D2H)
    unsigned char *io, *h_io, *ob, *h_ob;
    size_t ioBytes, obBytes;
    unsigned char* slab;  // ONE device allocation behind every workspace array (few TLB entries for the whole solve)
    int nCostBlocks;
    int maxObs;  // largest measurement count of a point of the uploaded problem
    int* pairPtr;     // camera-pair lists of the uploaded problem (own allocations: sized by the topology, not by C / P / nObs)
    int4* pairEnt;
    size_t pairPtrCap, pairEntCap;
    bool havePairs;
    int* waveStart;  // packed lane plan of the uploaded problem (own allocation), nPackWaves waves; 0 = none
    size_t waveStartCap;
    int nPackWaves;
    // cached executable graph of one full solve (cs_ba_solve_dev): ~150 launches become one
    struct GraphKey {
        int C, P, nObs, nCamsCon, nPtsCon, maxIter, innerMaxIter;
        double maxErr;
        const void *R0, *T0, *M0;
    } gkey;
    hipGraphExec_t gexec;
    struct BaWorker* worker;  // cs_ba_solve_async: the workspace's solver thread (the reference's BA thread)
    unsigned char* cholBuf;  // published columns | x | flags of the dataflow Cholesky (own allocation, grown on demand)
    size_t cholCap;
    unsigned char* syrkBuf;  // Zt | Tobs | Cpart | Udiag of the SYRK path (own allocation, grown on demand)
    size_t syrkCap;
    cs_ba_followup_fn followup;  // cs_ba_set_followup: enqueued on the solve's stream right behind every solve
    void* followupUser;
    struct cs_ba_output* output;  // cs_ba_output_attach: every window solve's result is packed into its next record
};

// solver breakdown is an error, not a silently unchanged estimate (the reference's callers catch what bundleAdjustRobust
// throws: src/app/SL_CoSLAMRobustBA.cpp:173-179)
static int ba_check_flags(int flags, const char* who) {
    if (flags & 4) {
        cs_set_error("%s: the dataflow Cholesky timed out waiting for a block column (a scheduling stall: another persistent kernel "
                     "holds the CUs its workgroups need); the step was rejected -- results are not to be trusted", who);
        return CS_ERR_NUMERIC;
    }
    if (flags & 2) {
        cs_set_error("%s: the reduced camera system could not be factorised in any LM step (not positive definite or not finite) "
                     "and no step was accepted: the estimate is unchanged", who);
        return CS_ERR_NUMERIC;
    }
    return CS_OK;
}

static int ba_run_followup(cs_ba* b, hipStream_t s) {
    if (!b->followup) return CS_OK;
    const int rc = b->followup((void*)s, b->followupUser);
    if (rc != CS_OK) cs_set_error("cs_ba: the follow-up of the solve failed (%d): %s", rc, cs_last_error());
    return rc;
}

static void ba_worker_drop_graphs(cs_ba* b);
static void ba_drop_graph(cs_ba* b) {
    // (callers on the API thread have drained the worker's queue -- cs_ba_wait -- before they get here; the worker itself gets
    // here only from ba_make_plan when a scratch allocation grew, with the API thread blocked out of the workspace by the same
    // rule: a queued job means every mutating entry point waits)
    if (b->gexec) (void)hipGraphExecDestroy(b->gexec);
    b->gexec = nullptr;
    ba_worker_drop_graphs(b);
}

static int ba_free(cs_ba* b) {
    ba_drop_graph(b);
    delete b->dist;
    b->dist = nullptr;
    if (b->pairPtr) (void)hipFree(b->pairPtr);
    if (b->pairEnt) (void)hipFree(b->pairEnt);
    if (b->waveStart) (void)hipFree(b->waveStart);
    b->waveStart = nullptr;
    b->waveStartCap = 0;
    b->nPackWaves = 0;
    b->pairPtr = nullptr;
    b->pairEnt = nullptr;
    b->pairPtrCap = b->pairEntCap = 0;
    b->havePairs = false;
    if (b->slab) (void)hipFree(b->slab);
    if (b->syrkBuf) (void)hipFree(b->syrkBuf);
    b->syrkBuf = nullptr;
    b->syrkCap = 0;
    if (b->cholBuf) (void)hipFree(b->cholBuf);
    b->cholBuf = nullptr;
    b->cholCap = 0;
    if (b->h_io) (void)hipHostFree(b->h_io);
    if (b->h_ob) (void)hipHostFree(b->h_ob);
    b->slab = nullptr;
    b->Rn = b->Tn = b->Mn = b->Jc = b->e = b->W = b->Vinv = b->gp = b->S = b->costPart = b->stepPart = b->schurPart = b->scal =
        b->Y = nullptr;
    b->st2 = nullptr;
    b->obs_pt = b->obs_of = nullptr;
    b->st = nullptr;
    b->io = b->ob = b->h_io = b->h_ob = nullptr;
    b->Ks = b->Rs = b->Ts = b->pts = b->obs_xy = nullptr;
    b->obs_ptr = b->obs_cam = b->cam_ptr = b->cam_obs = b->outlier = nullptr;
    b->stats = nullptr;
    return CS_OK;
}

struct BaIoLayout {
    size_t Ks, Rs, Ts, pts, xy, optr, ocam, cptr, cobs, total;
};
static BaIoLayout ba_io_layout(size_t C, size_t P, size_t O) {
    BaIoLayout L;
    size_t o = 0;
    auto take = [&o](size_t bytes) {
        const size_t at = o;
        o += (bytes + 7) & ~(size_t)7;
        return at;
    };
    L.Ks = take(72 * C);
    L.Rs = take(72 * C);
    L.Ts = take(24 * C);
    L.pts = take(24 * P);
    L.xy = take(16 * O);
    L.optr = take(4 * (P + 1));
    L.ocam = take(4 * O);
    L.cptr = take(4 * (C + 1));
    L.cobs = take(4 * O);
    L.total = o;
    return L;
}
// point the workspace's input / output arrays into the blocks for this problem size
static void ba_bind_io(cs_ba* b, int C, int P, int nObs) {
    const BaIoLayout L = ba_io_layout((size_t)C, (size_t)P, (size_t)nObs);
    b->Ks = (double*)(b->io + L.Ks);
    b->Rs = (double*)(b->io + L.Rs);
    b->Ts = (double*)(b->io + L.Ts);
    b->pts = (double*)(b->io + L.pts);
    b->obs_xy = (double*)(b->io + L.xy);
    b->obs_ptr = (int*)(b->io + L.optr);
    b->obs_cam = (int*)(b->io + L.ocam);
    b->cam_ptr = (int*)(b->io + L.cptr);
    b->cam_obs = (int*)(b->io + L.cobs);
    b->stats = (cs_ba_stats_dev*)b->ob;
    b->outlier = (int*)(b->ob + 64);
}

static int ba_reserve(cs_ba* b, int C, int P, int nObs) {
    if (C <= b->capC && P <= b->capP && nObs <= b->capObs) return CS_OK;
    ba_free(b);
    const size_t cC = (size_t)(C > b->capC ? C : b->capC), cP = (size_t)(P > b->capP ? P : b->capP),
                 cO = (size_t)(nObs > b->capObs ? nObs : b->capObs);
    const size_t n = 6 * cC;
    // every array of the workspace is carved out of ONE allocation (256-byte aligned pieces): the solve is a chain of
    // small dependent kernels, and two dozen separate allocations would mean two dozen translations to warm per kernel
    b->ioBytes = ba_io_layout(cC, cP, cO).total + 64;
    b->obBytes = 64 + 4 * (cO > 0 ? cO : 1);
    struct Piece {
        void** ptr;
        size_t bytes;
    };
    const Piece pieces[] = {
        {(void**)&b->st, sizeof(BaState)},
        {(void**)&b->st2, sizeof(BaState)},
        {(void**)&b->scal, 8 * sizeof(double)},
        {(void**)&b->costPart, (1024 + cP / 4 + cC / 256 + 2) * sizeof(double)},
        {(void**)&b->stepPart, (cP + cC + 1) * sizeof(double)},
        {(void**)&b->schurPart, (size_t)21 * 16 * 72 * sizeof(double)},  // <= 6 free cameras (21 pairs) x 16 slices
        {(void**)&b->S, (n * n + n + 8) * sizeof(double)},  // S || rhs contiguous: one all-reduce in the distributed solve
        {(void**)&b->Rn, (9 * cC + 1) * sizeof(double)},
        {(void**)&b->Tn, (3 * cC + 1) * sizeof(double)},
        {(void**)&b->Mn, (3 * cP + 1) * sizeof(double)},
        {(void**)&b->Vinv, (9 * cP + 1) * sizeof(double)},
        {(void**)&b->gp, (3 * cP + 1) * sizeof(double)},
        {(void**)&b->ob, b->obBytes},
        {(void**)&b->io, b->ioBytes},
        {(void**)&b->e, (2 * cO + 1) * sizeof(double)},
        {(void**)&b->Jc, (12 * cO + 1) * sizeof(double)},
        {(void**)&b->W, (18 * cO + 1) * sizeof(double)},
        {(void**)&b->Y, (18 * cO + 1) * sizeof(double)},
        {(void**)&b->obs_pt, (cO + 1) * sizeof(int)},
        {(void**)&b->obs_of, (cP * cC + 1) * sizeof(int)},
    };
    size_t total = 0;
    for (const Piece& q : pieces) total += (q.bytes + 255) & ~(size_t)255;
    CS_HIP(hipMalloc((void**)&b->slab, total));
    {
        size_t off = 0;
        for (const Piece& q : pieces) {
            *q.ptr = b->slab + off;
            off += (q.bytes + 255) & ~(size_t)255;
        }
    }
    b->rhs = nullptr;  // set per solve: S + n^2 of the actual order
    CS_HIP(hipHostMalloc((void**)&b->h_io, b->ioBytes, hipHostMallocDefault));
    CS_HIP(hipHostMalloc((void**)&b->h_ob, b->obBytes, hipHostMallocDefault));
    b->capC = (int)cC;
    b->capP = (int)cP;
    b->capObs = (int)cO;
    return CS_OK;
}

static int ba_make_plan(cs_ba* b, int C, int P, int nObs, int nCamsCon, int nPtsCon, double maxErr, int innerMaxIter,
                        bool distributed, BaPlan* out) {
    BaPlan& L = *out;
    BaDev& D = L.D;
    memset(&D, 0, sizeof(D));
    D.C = C;
    D.P = P;
    D.nObs = nObs;
    D.nCamsCon = nCamsCon > C ? C : nCamsCon;
    D.nPtsCon = nPtsCon > P ? P : nPtsCon;
    D.nc = C - D.nCamsCon;
    D.n = 6 * D.nc;
    D.Ks = b->Ks;
    D.Rs = b->Rs;
    D.Ts = b->Ts;
    D.pts = b->pts;
    D.Rn = b->Rn;
    D.Tn = b->Tn;
    D.Mn = b->Mn;
    D.obs_ptr = b->obs_ptr;
    D.obs_cam = b->obs_cam;
    D.obs_pt = b->obs_pt;
    D.cam_ptr = b->cam_ptr;
    D.cam_obs = b->cam_obs;
    D.obs_of = b->obs_of;
    D.pairPtr = b->havePairs ? b->pairPtr : nullptr;
    D.pairEnt = b->havePairs ? b->pairEnt : nullptr;
    D.obs_xy = b->obs_xy;
    D.outlier = b->outlier;
    D.Jc = b->Jc;
    D.e = b->e;
    D.W = b->W;
    D.Vinv = b->Vinv;
    D.gp = b->gp;
    D.S = b->S;
    b->rhs = b->S + (size_t)D.n * D.n;  // S || rhs contiguous
    D.rhs = b->rhs;
    D.costPart = b->costPart;
    D.stepPart = b->stepPart;
    D.schurPart = b->schurPart;
    D.scal = b->scal;
    D.st = b->st;
    D.maxErr = maxErr;
    D.innerMaxIter = innerMaxIter;
    D.pLo = 0;
    D.pHi = P;
    D.addLambda = 1;
    D.maxObsPerPoint = b->maxObs;
    D.Y = b->Y;
    D.stn = b->st2;
    D.waveStart = b->waveStart;
    D.nPackWaves = b->nPackWaves;
    int cb = (nObs + 255) / 256;
    if (cb < 1) cb = 1;
    if (cb > 1024) cb = 1024;
    D.nCostBlocks = cb;
    b->nCostBlocks = cb;
    L.cb = cb;
    L.nPairs = D.nc * (D.nc + 1) / 2;
    {
        if (D.n > 36 && D.n <= SB_MAX_ORDER && sb_lds_bytes(D.n) > 64 * 1024) {
            CS_HIP(hipFuncSetAttribute((const void*)k_solve_blocked<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_lds_bytes(D.n)));
            CS_HIP(hipFuncSetAttribute((const void*)k_solve_blocked<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_lds_bytes(D.n)));
            CS_HIP(hipFuncSetAttribute((const void*)k_solve_blocked<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_lds_bytes(D.n)));
        }
    }
    L.gPts = (P + 3) / 4 > 0 ? (P + 3) / 4 : 1;
    int gUpd = (P + 3) / 4;
    if (gUpd * 256 < C) gUpd = (C + 255) / 256;
    if (gUpd < 1) gUpd = 1;
    D.nUpdBlocks = gUpd;
    L.gUpd = gUpd;
    // orders <= 36: sliced Schur partials, combined by the register Cholesky (single-process solve only: the
    // distributed solve needs the dense S || rhs for its all-reduce)
    L.sliced = (!distributed && D.n > 0 && D.n <= 36);
    {
        L.seg8 = L.sliced && b->maxObs <= 8;
        if (L.seg8) {
            gUpd = (P + 31) / 32;
            if (gUpd * 256 < C) gUpd = (C + 255) / 256;
            if (gUpd < 1) gUpd = 1;
            D.nUpdBlocks = gUpd;
            L.gUpd = gUpd;
        }
    }
    D.nSlices = 1;
    if (L.sliced) {
        int sl = 96 / (L.nPairs > 0 ? L.nPairs : 1);
        if (sl > 16) sl = 16;
        if (sl < 1) sl = 1;
        const int nFree = P - D.nPtsCon;
        while (sl > 1 && (nFree + sl - 1) / sl < 32) sl /= 2;  // at least half a wave of points per slice
        D.nSlices = sl;
    }
    // orders beyond the LDS solver: block columns owned by workgroups, one launch (beyond its 66 block columns: the
    // launch-per-block kernels)
    L.cholFlow = false;
    {
        const int NB = (D.n + SB - 1) / SB;
        if (D.n > SB_MAX_ORDER && NB + 1 <= CF_MAX_BLOCKS) {
            auto pad = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t bPub = pad(sizeof(double) * (size_t)NB * (NB + 1) * 256), bX = pad(sizeof(double) * 16 * (size_t)NB),
                         bF = pad(sizeof(int) * 2 * (size_t)NB);
            const size_t need = bPub + bX + bF;
            if (need > b->cholCap) {
                if (b->cholBuf) (void)hipFree(b->cholBuf);
                b->cholBuf = nullptr;
                b->cholCap = 0;
                if (hipMalloc((void**)&b->cholBuf, need) != hipSuccess) {
                    cs_set_error("cs_ba: cannot allocate %zu KB for the Cholesky columns", need >> 10);
                    return CS_ERR_ALLOC;
                }
                b->cholCap = need;
                ba_drop_graph(b);
            }
            L.F.pub = (double*)b->cholBuf;
            L.F.xpub = (double*)(b->cholBuf + bPub);
            L.F.flagL = (int*)(b->cholBuf + bPub + bX);
            L.F.flagX = L.F.flagL + NB;
            L.F.NB = NB;
            CS_HIP(hipFuncSetAttribute((const void*)k_cholflow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cf_lds_bytes(NB + 1)));
            L.cholFlow = true;
        }
    }
    // orders beyond the LDS solver whose pair lists were too large to build: Z Z^T on the matrix cores
    L.syrk = false;
    {
        // cs_debug_set("ba_syrk", 0): never (k_schur); 2: always, also where the pair lists or the LDS solver would apply (tests)
        const int sw = cs_debug_get(CS_DBG_BA_SYRK);
        const bool noSyrk = sw == 0, force = sw == 2;
        if (force && D.n > 0) D.pairPtr = nullptr, D.pairEnt = nullptr;
        if (!distributed && !noSyrk && !D.pairPtr && (D.n > SB_MAX_ORDER || (force && D.n > 0)) && P > 0 && nObs > 0) {
            SyrkDev& Y = L.Y;
            Y.nT = (D.n + SY_TB - 1) / SY_TB;
            Y.nTiles = Y.nT * (Y.nT + 1) / 2;
            Y.ldz = Y.nT * SY_TB;
            int sl = 512 / Y.nTiles;  // two workgroups per CU: 189 us at cfg5 (one per CU: 204, 1.3 per CU: 259)
            if (sl > 32) sl = 32;
            if (sl < 1) sl = 1;
            const int K = 3 * P;
            int ks = (K + sl - 1) / sl;
            ks = (ks + SY_KC - 1) / SY_KC * SY_KC;
            Y.nSlices = sl;
            Y.Kslice = ks;
            Y.Kpad = ks * sl;
            auto pad = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t bZt = pad(sizeof(double) * (size_t)Y.Kpad * Y.ldz), bT = pad(sizeof(double) * 6 * (size_t)nObs),
                         bC = pad(sizeof(double) * (size_t)sl * Y.nTiles * SY_TB * SY_TB), bU = pad(sizeof(double) * 33 * SY_US * (size_t)D.nc);
            const size_t need = bZt + bT + bC + bU;
            if (need > b->syrkCap) {
                if (b->syrkBuf) (void)hipFree(b->syrkBuf);
                b->syrkBuf = nullptr;
                b->syrkCap = 0;
                if (hipMalloc((void**)&b->syrkBuf, need) != hipSuccess) {
                    cs_set_error("cs_ba: cannot allocate %zu MB for the Schur contraction", need >> 20);
                    return CS_ERR_ALLOC;
                }
                b->syrkCap = need;
                ba_drop_graph(b);  // captured graphs hold the old addresses
            }
            Y.Zt = (double*)b->syrkBuf;
            Y.Tobs = (double*)(b->syrkBuf + bZt);
            Y.Cpart = (double*)(b->syrkBuf + bZt + bT);
            Y.Udiag = (double*)(b->syrkBuf + bZt + bT + bC);
            L.syrkZtBytes = bZt;
            CS_HIP(hipFuncSetAttribute((const void*)k_syrk_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(double) * 4 * SY_KC * SY_LDP)));
            L.syrk = true;
        }
    }
    // orders 37..176 with pair lists and a lane plan: the kernels that fit a few compute units (cs_debug_set("ba_packed", 0): the
    // wave-per-point / workgroup-per-pair kernels, for A/B runs)
    {
        const bool noPacked = cs_debug_get(CS_DBG_BA_PACKED) == 0;
        L.packed = !noPacked && !distributed && !L.sliced && !L.syrk && D.pairPtr && b->nPackWaves > 0 &&
                   D.n > 36 && D.n <= SB_MAX_ORDER && P > 0 && nObs > 0;
        L.gPack = 0;
        if (L.packed) {
            int g = (b->nPackWaves + 3) / 4;
            if (g * 256 < C) g = (C + 255) / 256;
            L.gPack = g;
            D.nUpdBlocks = g;
            L.DB = D;
            L.DB.st = b->st2;
            L.DB.stn = b->st;
        }
    }
    return CS_OK;
}

// rebuildTopology = false: the measurement tables (obs_pt, obs_of) of the uploaded problem are already built
static void ba_enqueue_init(cs_ba* b, hipStream_t stream, const BaPlan& L, bool rebuildTopology = true,
                            const double* d_Rs0 = nullptr, const double* d_Ts0 = nullptr, const double* d_pts0 = nullptr) {
    const BaDev& D = L.D;
    int gi = ((D.nObs > 3 * D.P ? D.nObs : 3 * D.P) + 255) / 256;
    if (gi < 1) gi = 1;
    if (gi > 64) gi = 64;
    BaInitCopy I = {d_Rs0, d_Ts0, d_pts0, b->Rs, b->Ts, b->pts, 9 * D.C, 3 * D.C, 3 * D.P};
    hipLaunchKernelGGL(k_init_state, dim3(gi), dim3(256), 0, stream, b->st, b->outlier, D.nObs, I);
    // Z's entries of (point, camera) pairs without a measurement are never written: cleared once per solve (every present
    // entry is rewritten by every step)
    if (L.syrk) (void)hipMemsetAsync(L.Y.Zt, 0, L.syrkZtBytes, stream);
    if (L.packed) hipLaunchKernelGGL(k_mirror_estimate, dim3(gi > 16 ? 16 : gi), dim3(256), 0, stream, D);
    if (!rebuildTopology) return;
    (void)hipMemsetAsync(b->obs_of, 0xff, sizeof(int) * (size_t)D.P * D.C, stream);
    if (D.P > 0) hipLaunchKernelGGL(k_build_obs_pt, dim3((D.P + 3) / 4), dim3(256), 0, stream, D.P, b->obs_ptr, b->obs_pt);
    if (D.nObs > 0)
        hipLaunchKernelGGL(k_build_obs_of, dim3((D.nObs + 255) / 256), dim3(256), 0, stream, D.nObs, D.C, b->obs_pt,
                           b->obs_cam, b->obs_of);
}

// linearisation + reduced system of one LM step (this rank's point slice)
static void ba_enqueue_lin_schur(hipStream_t stream, const BaPlan& L) {
    const BaDev& D = L.D;
    const dim3 blk(256);
    if (L.packed) {  // (+ the LM control of the previous step in the head of the linearisation)
        hipLaunchKernelGGL(k_lin_packed, dim3(L.gPack), blk, 0, stream, D);
        if (D.nc > 0) hipLaunchKernelGGL(k_schur_wave, dim3((L.nPairs * CS_SCHUR_WPP + 3) / 4), blk, 0, stream, L.DB);
        return;
    }
    {
        if (D.maxObsPerPoint <= 8)
            hipLaunchKernelGGL(k_linearize_seg8, dim3((D.P + 31) / 32 > 0 ? (D.P + 31) / 32 : 1), blk, 0, stream, D);
        else
            hipLaunchKernelGGL(k_linearize, dim3(L.gPts), blk, 0, stream, D);
    }
    if (D.nc > 0) {
        if (L.sliced)
            hipLaunchKernelGGL(k_schur_part, dim3(L.nPairs * D.nSlices), dim3(64), 0, stream, D);
        else
            if (D.pairPtr) {
                hipLaunchKernelGGL(k_schur_pairs, dim3(L.nPairs), blk, 0, stream, D);
            } else if (L.syrk) {
                const SyrkDev& Y = L.Y;
                hipLaunchKernelGGL(k_syrk_pack, dim3(L.gPts), blk, 0, stream, D, Y);
                hipLaunchKernelGGL(k_syrk_mfma, dim3(Y.nTiles * Y.nSlices), blk, sizeof(double) * 4 * SY_KC * SY_LDP, stream, D, Y);
                hipLaunchKernelGGL(k_schur_diag_u, dim3(D.nc * SY_US), blk, 0, stream, D, Y);
                hipLaunchKernelGGL(k_syrk_reduce, dim3((unsigned)(((size_t)D.n * D.n + 255) / 256)), blk, 0, stream, D, Y);
            } else {
                hipLaunchKernelGGL(k_schur, dim3(L.nPairs), blk, 0, stream, D);
            }
    }
}

// solve of the reduced system + tentative step + its cost
static void ba_enqueue_solve_update(hipStream_t stream, const BaPlan& L) {
    const BaDev& D = L.D;
    const dim3 blk(256);
    const int gUpd = L.gUpd;
    if (L.packed) {
        sb_launch_solve(stream, L.DB);
        hipLaunchKernelGGL(k_update_packed, dim3(L.gPack), blk, 0, stream, L.DB);
        return;
    }
    if (L.seg8) {
        switch (D.n) {
            case 6: hipLaunchKernelGGL(k_update_seg8<6>, dim3(gUpd), blk, 0, stream, D); break;
            case 12: hipLaunchKernelGGL(k_update_seg8<12>, dim3(gUpd), blk, 0, stream, D); break;
            case 18: hipLaunchKernelGGL(k_update_seg8<18>, dim3(gUpd), blk, 0, stream, D); break;
            case 24: hipLaunchKernelGGL(k_update_seg8<24>, dim3(gUpd), blk, 0, stream, D); break;
            case 30: hipLaunchKernelGGL(k_update_seg8<30>, dim3(gUpd), blk, 0, stream, D); break;
            default: hipLaunchKernelGGL(k_update_seg8<36>, dim3(gUpd), blk, 0, stream, D); break;
        }
    } else if (L.sliced && D.n == 6) {
        hipLaunchKernelGGL(k_update<6>, dim3(gUpd), blk, 0, stream, D);  // + solve + tentative cost
    } else if (L.sliced && D.n == 12) {
        hipLaunchKernelGGL(k_update<12>, dim3(gUpd), blk, 0, stream, D);
    } else if (L.sliced && D.n == 18) {
        hipLaunchKernelGGL(k_update<18>, dim3(gUpd), blk, 0, stream, D);
    } else if (L.sliced && D.n == 24) {
        hipLaunchKernelGGL(k_update<24>, dim3(gUpd), blk, 0, stream, D);
    } else if (L.sliced && D.n == 30) {
        hipLaunchKernelGGL(k_update<30>, dim3(gUpd), blk, 0, stream, D);
    } else if (L.sliced && D.n == 36) {
        hipLaunchKernelGGL(k_update<36>, dim3(gUpd), blk, 0, stream, D);
    } else {
        if (D.n <= SB_MAX_ORDER) {
            sb_launch_solve(stream, D);
        } else if (L.cholFlow) {  // one launch: workgroup per block column, flags in HBM
            hipLaunchKernelGGL(k_cholflow_begin, dim3(1), dim3(128), 0, stream, D, L.F);
            hipLaunchKernelGGL(k_cholflow, dim3(L.F.NB), dim3(CF_NT), cf_lds_bytes(L.F.NB + 1), stream, D, L.F);
        } else {  // blocked Cholesky in HBM
            hipLaunchKernelGGL(k_chol_begin, dim3(1), dim3(1), 0, stream, D);
            for (int k0 = 0; k0 < D.n; k0 += CB) {
                const int kb = (D.n - k0 < CB) ? D.n - k0 : CB;
                const int below = D.n - k0 - kb;
                int gp = (below + 255) / 256;
                if (gp < 1) gp = 1;
                hipLaunchKernelGGL(k_chol_panel, dim3(gp), blk, 0, stream, D, k0);
                if (below > 0) {
                    const int nt = (below + CT - 1) / CT;
                    hipLaunchKernelGGL(k_chol_trail, dim3(nt * (nt + 1) / 2), blk, 0, stream, D, k0);
                }
            }
            hipLaunchKernelGGL(k_chol_trsv, dim3(1), dim3(1024), sizeof(double) * (size_t)D.n, stream, D);
        }
        hipLaunchKernelGGL(k_update<0>, dim3(gUpd), blk, 0, stream, D);  // + the tentative cost
    }
}

// LM control behind a tentative step: its own launch, or (packed path) nothing -- the next linearisation decides
static void ba_enqueue_control_step(hipStream_t stream, const BaPlan& L) {
    if (!L.packed) hipLaunchKernelGGL(k_control_step, dim3(1), dim3(256), 0, stream, L.D);
}
// ... and at the end of a run of LM steps (packed path): the pending decision, the estimate back in Rs / Ts / pts
static void ba_enqueue_control_final(hipStream_t stream, const BaPlan& L) {
    if (L.packed) hipLaunchKernelGGL(k_control_final, dim3(1), dim3(256), 0, stream, L.D);
}

// a run of up to `steps` LM steps (the state word says where it stops)
static void ba_enqueue_lm_run(hipStream_t stream, const BaPlan& L, int steps) {
    for (int it = 0; it < steps; ++it) {
        ba_enqueue_lin_schur(stream, L);
        ba_enqueue_solve_update(stream, L);
        ba_enqueue_control_step(stream, L);
    }
    ba_enqueue_control_final(stream, L);
}

// enqueue the whole solve on `stream`; every array already resident in b's device buffers
static int ba_enqueue(cs_ba* b, hipStream_t stream, int C, int P, int nObs, int nCamsCon, int nPtsCon, double maxErr,
                      int maxIter, int innerMaxIter, bool rebuildTopology = true, const double* d_Rs0 = nullptr,
                      const double* d_Ts0 = nullptr, const double* d_pts0 = nullptr) {
    BaPlan L;
    int rc = ba_make_plan(b, C, P, nObs, nCamsCon, nPtsCon, maxErr, innerMaxIter, false, &L);
    if (rc) return rc;
    const BaDev& D = L.D;
    const int cb = L.cb;
    const dim3 blk(256);
    ba_enqueue_init(b, stream, L, rebuildTopology, d_Rs0, d_Ts0, d_pts0);

    for (int outer = 0; outer < maxIter; ++outer) {
        hipLaunchKernelGGL(k_cost, dim3(cb), blk, 0, stream, D, 0);
        hipLaunchKernelGGL(k_control, dim3(1), blk, 0, stream, D);
        ba_enqueue_lm_run(stream, L, innerMaxIter);
        hipLaunchKernelGGL(k_flag, dim3(cb), blk, 0, stream, D);  // (its counters were zeroed by k_control phase 0)
        hipLaunchKernelGGL(k_outer_end, dim3(1), dim3(1), 0, stream, D);
    }
    hipLaunchKernelGGL(k_cost_force, dim3(cb), blk, 0, stream, D);
    hipLaunchKernelGGL(k_finish, dim3(1), blk, 0, stream, D, b->stats);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// ---- cs_ba_solve_async: the solve on the workspace's own thread, enqueued in chunks -----------------------------------
// cs_ba_solve_dev enqueues the whole maxIter x innerMaxIter schedule up front; a run that converges early leaves the rest
// behind as no-op launches (~2 us each: the inter-camera solve -- 3 x 40 steps scheduled, ~13 taken -- spent more time in
// them than in its LM steps).  The reference runs bundle adjustment on a worker thread next to tracking
// (src/app/SL_CoSLAM.cpp:1702-1784); so does this entry: a thread per workspace replays captured graphs of CHUNK LM
// steps on the workspace's stream and reads the 8-byte {inner_done, all_done} word back between chunks, so at most one
// chunk of no-ops is ever issued.  The caller's thread only records an event and queues the request.
struct BaAsyncJob {
    int C, P, nObs, nCamsCon, nPtsCon, maxIter, innerMaxIter;
    double maxErr;
    const double *R0, *T0, *M0;
    hipEvent_t ready;
    struct cs_ba_window* win = nullptr;  // cs_ba_solve_window_async: the problem is built on the device from this window
    const double* d_map = nullptr;
    const unsigned char* d_mapStatic = nullptr;
    const double *winR = nullptr, *winT = nullptr;   // the ring's key poses as they stood when the solve was requested
    struct cs_ba_intercam* ic = nullptr;             // cs_ba_solve_intercam_async: the problem sits in staging record icSlot
    int icSlot = 0;
    int winCount = 0, winSlotOf[16], winFrames[16];  // the window as it stood when the solve was requested (oldest first)
    long long winNewest = -1;                        // the push count of its newest key frame (cs_ba_window::pendingNewest)
};

struct BaWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv, cvDone;
    std::deque<BaAsyncJob> q;
    bool stop = false;
    int inflight = 0;
    int lastRc = CS_OK;
    char err[256] = "";
    // captured pieces of the solve, keyed like the whole-solve graph
    cs_ba::GraphKey key;
    bool haveGraphs = false;
    int chunk = 0;
    hipGraphExec_t gHead = nullptr, gChunk = nullptr, gTail = nullptr, gRound = nullptr, gFinish = nullptr;
    int* h_state = nullptr;  // pinned {inner_done, all_done}
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t tmEv[3] = {nullptr, nullptr, nullptr};  // timing: the job's first and last moment on the workspace's stream, end of the window parse
    int jobsDone = 0;
    long long jobsCompleted = 0;  // since the workspace was created (cs_ba_completed)
    double gpuMsTotal = 0, gpuMsLast = 0, gpuMsMax = 0, gpuMsParse = 0;
    bool parseStamped = false;
    std::thread::id tid;     // the worker thread: the only one that destroys the graph handles above
    bool stale = false;      // (under mu) another thread invalidated what the graphs bake in: re-capture before the next job
};

// The worker's graph handles are worker-private: any other thread only marks them stale (under the mutex) and the worker
// destroys and re-captures them before its next job; every entry point that rewrites what they bake in (upload, solve_dev,
// dist_begin, set_stream) waits for the queue to drain first (cs_ba_wait), so no job can be replaying them meanwhile.
static void ba_worker_destroy_graphs(BaWorker* w) {
    for (hipGraphExec_t* g : {&w->gHead, &w->gChunk, &w->gTail, &w->gRound, &w->gFinish}) {
        if (*g) (void)hipGraphExecDestroy(*g);
        *g = nullptr;
    }
    w->haveGraphs = false;
}
static void ba_worker_drop_graphs(cs_ba* b) {
    BaWorker* w = b->worker;
    if (!w) return;
    if (!w->th.joinable() || std::this_thread::get_id() == w->tid) {
        ba_worker_destroy_graphs(w);
        return;
    }
    std::lock_guard<std::mutex> lk(w->mu);
    w->stale = true;
}

template <class F>
static int ba_capture(hipStream_t s, hipGraphExec_t* out, F&& body) {
    CS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    body();
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        cs_set_error("cs_ba_solve_async: hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return CS_ERR_HIP;
    }
    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        *out = nullptr;
        cs_set_error("cs_ba_solve_async: hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return CS_ERR_HIP;
    }
    return CS_OK;
}

// Packed launch-per-phase schedule: the last launch of a chunk (k_control_final) and of a tail (k_outer_end) store the state word
// straight into the worker's pinned host word -- no copy node on the chain.  The schedules of large systems keep the copy.
static bool ba_state_in_kernel(BaWorker* w, BaPlan& L) {
    L.D.hostState = L.DB.hostState = nullptr;
    if (!L.packed || !w->h_state) return false;
    int* dp = nullptr;
    if (hipHostGetDevicePointer((void**)&dp, w->h_state, 0) != hipSuccess || !dp) return false;
    L.D.hostState = L.DB.hostState = dp;
    return true;
}

// ---- the sliding window of key frames (ba_window_dev.h) ------------------------------------------------------------------
// The schedule of a robust solve as a list of segments -- head, then per round [round start,] chunks of w->chunk LM steps, tail,
// then finish -- issued with ONE segment of look-ahead: segment i + 1 is on the stream before the host waits for segment i's
// state word, so the stream never idles for a host round trip (~40 us per chunk next to a busy tracker).  What the state word
// says only prunes segments that are not launched yet; a speculative segment behind a converged one runs as no-ops (every
// kernel tests the state first).  `launch(kind)` puts one segment on the stream: a captured graph (cs_ba_solve_async, sizes
// fixed) or the kernels themselves (cs_ba_solve_window_async, sizes new with every key frame); chunk and tail segments end
// with the copy of {inner_done, all_done} into w->h_state.
template <class F>
static int ba_run_segments(BaWorker* w, hipStream_t s, int maxIter, int innerMaxIter, F&& launch) {
    struct Seg {
        char kind;  // 'H'ead, 'R'ound start, 'C'hunk, 'T'ail, 'U' = tail + next round's start, 'F'inish
        int outer;
    };
    std::vector<Seg> seg;
    // (a round's tail and the next round's start are ONE segment, 'U' = T + R: two launches each -- as separate segments the second
    // one waited for the host thread to wake up from the first one's event, ~15 us of idle stream per round; behind a converged
    // tail the round start runs as no-ops like any speculative segment)
    seg.push_back({'H', 0});
    for (int outer = 0; outer < maxIter; ++outer) {
        for (int done = 0; done < innerMaxIter; done += w->chunk) seg.push_back({'C', outer});
        seg.push_back({outer + 1 < maxIter ? 'U' : 'T', outer});
    }
    seg.push_back({'F', maxIter});
    if (!w->ev[0]) {
        CS_HIP(hipEventCreateWithFlags(&w->ev[0], hipEventDisableTiming));
        CS_HIP(hipEventCreateWithFlags(&w->ev[1], hipEventDisableTiming));
    }
    int skipChunksOfOuter = -1;  // chunks of this round that are not launched yet are dropped
    bool allDone = false;        // ... and everything up to the finish segment
    auto next_of = [&](size_t i) {
        size_t n = i + 1;
        while (n < seg.size()) {
            const Seg& q = seg[n];
            if (q.kind != 'F' && allDone) {
                ++n;
                continue;
            }
            if (q.kind == 'C' && q.outer == skipChunksOfOuter) {
                ++n;
                continue;
            }
            break;
        }
        return n;
    };
    auto launch_seg = [&](char kind) {
        if (kind != 'U') return launch(kind);
        const hipError_t e = launch('T');
        return e != hipSuccess ? e : launch('R');
    };
    size_t i = 0;
    CS_HIP(launch_seg(seg[0].kind));
    CS_HIP(hipEventRecord(w->ev[0], s));
    int slot = 0;
    while (i < seg.size()) {
        const size_t n = next_of(i);
        if (n < seg.size()) {
            CS_HIP(launch_seg(seg[n].kind));
            CS_HIP(hipEventRecord(w->ev[slot ^ 1], s));
        }
        CS_HIP(hipEventSynchronize(w->ev[slot]));
        if (seg[i].kind == 'C' && (w->h_state[0] || w->h_state[1])) skipChunksOfOuter = seg[i].outer;  // inner_done / all_done
        if ((seg[i].kind == 'T' || seg[i].kind == 'U') && w->h_state[1]) allDone = true;
        i = n;
        slot ^= 1;
    }
    return CS_OK;
}

// The ring holds WIN_SLACK key frames more than a window: a request's window (its slots are fixed when the solve is REQUESTED)
// stays intact while up to WIN_SLACK newer key frames are pushed -- the frame loop's thread may run that far ahead of the
// worker's parse; one further push waits (on the host) for the oldest outstanding parse.
constexpr int WIN_SLACK = 2, WIN_MAX_RING = 16 + WIN_SLACK;
struct cs_ba_window {
    int device, nCams, nKf, N, nMap;
    int ring;            // slots in the ring: nKf + WIN_SLACK
    int head, count;     // ring: the next slot to write; key frames held (<= nKf)
    int frameOf[WIN_MAX_RING];
    std::mutex mu;       // parsesPending / lastFrames: the requesting thread and the worker
    std::condition_variable cv;
    int parsesPending;   // solves requested whose parse has not read the ring yet
    // ... and WHICH: the push count of each pending request's newest key frame.  Push n rewrites the slot of push n - ring, which a request whose
    // newest key frame is push m still reads when m - nKf + 1 <= n - ring, i.e. m < n - WIN_SLACK.  (Counting the pending requests alone is the
    // same thing only when every push is followed by a request: with the windows dealt round the ranks a rank requests every N-th one, and a
    // loop that places a key frame per frame then rewrote slots under a parse -- found by the two-rank run of the key-frame decision, round 6.)
    long long pushCount;
    std::deque<long long> pendingNewest;
    int lastFrames[16], lastCount;
    unsigned char* slab;
    double *xy, *K, *R, *t;
    int *pf, *cnt, *ptIndex, *obsStart, *totals, *pointMap, *pairCnt, *pairTotal;
    double* mapSnap[WIN_SLACK + 1];   // the map as it stood when a solve was requested: one per request that can be outstanding
    unsigned char* staticSnap[WIN_SLACK + 1];
    double *poseSnapR[WIN_SLACK + 1], *poseSnapT[WIN_SLACK + 1];   // ... and the ring's key poses (cs_ba_output_apply_dev rewrites them)
    int snapNext;
    int* h_plan;         // pinned [nMap + 2]: the lane plan on its way to the device
    int* h_totals;       // pinned [8], followed by nMap + 1 ints: obs_ptr of the parsed problem (for the lane plan)
    int lastC, lastP, lastObs;
};

// ---- RobustBundleRTS::output() (ba_output_dev.h): a ring of result records, packed by the worker behind every window solve ----
struct cs_ba_output {
    int device, nCams, nKf, nMap, nSlots;
    BoLayout L;
    unsigned char* slab;  // nSlots records of L.bytes
    int* d_err;           // [2]: cs_ba_output_wait_dev's waits that gave up; records cs_ba_output_apply_seq_dev refused (not the one expected)
    int applyMask;        // cs_ba_output_set_apply_mask (diagnostic): which parts of output() an apply performs
    const void* featRef = nullptr;            // cs_ba_output_set_feat_refs: [nMap][nCams] cs_feat_ref or null
    const unsigned char* refStatic = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    long long issued;     // records the worker has started to pack (its slot = issued % nSlots)
    long long packed;     // records complete on the device (the worker has synchronised with the pack)
    // the applying side's scratch (cs_ba_output_apply_dev: one caller thread): camera graphs of nCams chains of graphNodes nodes
    cs_posegraph* graph;
    int graphNodes, graphKeyNode[16], maxNodes;   // graphKeyNode[j]: key frame j's node in a chain (its frame number - the first key frame's)
    double *nodeR, *nodeT, *newR, *newT, *edgeR, *edgeT;
    unsigned char* scratch;
};

static int ba_output_pack(cs_ba* b, cs_ba_window* win, const BaAsyncJob& J, hipStream_t s, int C, int P, int nObs, bool ok) {
    cs_ba_output* o = b->output;
    if (C > o->L.maxC || P > o->L.maxP) ok = false;   // (cannot happen for the window the record was sized for)
    BoPackArgs A;
    memset(&A, 0, sizeof(A));
    A.C = ok ? C : 0, A.P = ok ? P : 0, A.nObs = ok ? nObs : 0, A.nKf = J.winCount, A.nCams = win->nCams, A.ok = ok ? 1 : 0;
    for (int j = 0; j < 16; ++j) A.kfFrame[j] = j < J.winCount ? J.winFrames[j] : -1;
    long long seq;
    {
        std::lock_guard<std::mutex> lk(o->mu);
        seq = o->issued++;
    }
    A.seq = (int)seq;
    A.Rs = b->Rs, A.Ts = b->Ts, A.pts = b->pts, A.obs_ptr = b->obs_ptr, A.outlier = b->outlier, A.pointMap = win->pointMap;
    A.rec = o->slab + (size_t)(seq % o->nSlots) * o->L.bytes;
    A.L = o->L;
    int n = 9 * A.C > 3 * A.P ? 9 * A.C : 3 * A.P;
    if (n < BO_HDR_INTS) n = BO_HDR_INTS;
    hipLaunchKernelGGL(k_ba_output_pack, dim3((n + 255) / 256), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_ba_output_publish, dim3(1), dim3(1), 0, s, (int*)A.rec, A.seq);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
static void ba_output_publish(cs_ba_output* o) {
    {
        std::lock_guard<std::mutex> lk(o->mu);
        o->packed += 1;
    }
    o->cv.notify_all();
}

static int ba_worker_run_window_inner(cs_ba* b, BaWorker* w, const BaAsyncJob& J, bool* packed);
static int ba_worker_run_window(cs_ba* b, BaWorker* w, const BaAsyncJob& J) {
    bool packed = false;
    const int rc = ba_worker_run_window_inner(b, w, J, &packed);
    if (b->output && !packed) {   // the solve failed before it had a result: an empty record keeps the sequence of records = requests
        (void)hipSetDevice(b->device);
        char keep[256];
        snprintf(keep, sizeof(keep), "%s", cs_last_error());
        (void)ba_output_pack(b, J.win, J, b->own_stream, 0, 0, 0, false);
        (void)hipStreamSynchronize(b->own_stream);
        ba_output_publish(b->output);
        cs_set_error("%s", keep);
    }
    return rc;
}
static int ba_worker_run_window_inner(cs_ba* b, BaWorker* w, const BaAsyncJob& J, bool* packed) {
    cs_ba_window* win = J.win;
    CS_HIP(hipSetDevice(b->device));
    hipStream_t s = b->own_stream;
    CS_HIP(hipStreamWaitEvent(s, J.ready, 0));
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->stale) {
            ba_worker_destroy_graphs(w);
            w->stale = false;
        }
    }
    struct ParseDone {   // whatever way this function is left: the ring is free for the pushes that wait for this parse
        cs_ba_window* w;
        long long newest;
        bool done = false;
        void release() {
            if (done) return;
            done = true;
            {
                std::lock_guard<std::mutex> lk(w->mu);
                w->parsesPending -= 1;
                for (auto it = w->pendingNewest.begin(); it != w->pendingNewest.end(); ++it)
                    if (*it == newest) {
                        w->pendingNewest.erase(it);
                        break;
                    }
            }
            w->cv.notify_all();
        }
        ~ParseDone() { release(); }
    } parseDone{win, J.winNewest};
    if (J.winCount < 1) {
        cs_set_error("cs_ba_solve_window_async: the window holds no key frame");
        return CS_ERR_INVALID;
    }
    const int C = J.winCount * win->nCams;
    // the workspace for the largest problem this window can produce: every slot of every key camera a measurement
    int rc = ba_reserve(b, win->nKf * win->nCams, win->nMap, win->nKf * win->nCams * win->N);
    if (rc) return rc;
    ba_bind_io(b, b->capC, b->capP, b->capObs);
    ba_drop_graph(b);
    WinDev Wd;
    memset(&Wd, 0, sizeof(Wd));
    Wd.nCams = win->nCams, Wd.nKf = win->nKf, Wd.N = win->N, Wd.nMap = win->nMap, Wd.count = J.winCount;
    for (int j = 0; j < J.winCount; ++j) Wd.slotOf[j] = J.winSlotOf[j];  // oldest first
    Wd.xy = win->xy, Wd.pf = win->pf, Wd.K = win->K, Wd.R = J.winR ? J.winR : win->R, Wd.t = J.winT ? J.winT : win->t;
    Wd.mapStatic = J.d_mapStatic, Wd.mapPts = J.d_map;
    Wd.cnt = win->cnt, Wd.ptIndex = win->ptIndex, Wd.obsStart = win->obsStart, Wd.totals = win->totals;
    const int gM = (win->nMap + 3) / 4 > 0 ? (win->nMap + 3) / 4 : 1;   // a wave per map point
    hipLaunchKernelGGL(k_win_count, dim3(gM), dim3(256), 0, s, Wd);
    hipLaunchKernelGGL(k_win_scan, dim3(1), dim3(1024), 0, s, Wd);
    WinFillOut O = {b->Ks, b->Rs, b->Ts, b->pts, b->obs_xy, b->obs_ptr, b->obs_cam, win->pointMap, b->obs_pt, b->obs_of};
    const int gF = (win->nMap + 3) / 4 > (C + 255) / 256 ? (win->nMap + 3) / 4 : (C + 255) / 256;   // a wave per map point; a thread per key camera
    hipLaunchKernelGGL(k_win_fill, dim3(gF), dim3(256), 0, s, Wd, O);
    if (6 * (C - J.nCamsCon) <= 36 && C <= 256)   // (the small-order solver's Schur kernels walk camera-indexed lists)
        hipLaunchKernelGGL(k_cam_lists, dim3(1), dim3(1024), 0, s, C, win->totals, b->obs_cam, b->cam_ptr, b->cam_obs);
    // the camera-pair lists' sizes, still without the host knowing P (one wave per pair; P read on the device)
    const int nPairsAll = C * (C + 1) / 2;
    if ((size_t)nPairsAll + 1 > b->pairPtrCap) {
        if (b->pairPtr) (void)hipFree(b->pairPtr);
        b->pairPtr = nullptr;
        CS_HIP(hipMalloc((void**)&b->pairPtr, sizeof(int) * ((size_t)nPairsAll + 1)));
        b->pairPtrCap = (size_t)nPairsAll + 1;
    }
    hipLaunchKernelGGL(k_pairs_count, dim3(nPairsAll), dim3(64), 0, s, C, win->totals, b->obs_of, win->pairCnt);
    hipLaunchKernelGGL(k_pairs_scan, dim3(1), dim3(1024), 0, s, nPairsAll, win->pairCnt, b->pairPtr, win->pairTotal);
    // ONE round trip: the sizes (launch dimensions of everything that follows), the pair total, obs_ptr for the lane plan
    CS_HIP(hipMemcpyAsync(win->h_totals, win->totals, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(win->h_totals + 4, win->pairTotal, sizeof(int), hipMemcpyDeviceToHost, s));
    int* h_optr = win->h_totals + 8;
    CS_HIP(hipMemcpyAsync(h_optr, b->obs_ptr, sizeof(int) * ((size_t)win->nMap + 1), hipMemcpyDeviceToHost, s));
    if (w->tmEv[2]) {
        (void)hipEventRecord(w->tmEv[2], s);
        w->parseStamped = true;
    }
    CS_HIP(hipStreamSynchronize(s));  // (this is the worker thread: the frame loop does not wait)
    parseDone.release();              // the ring has been read
    const int P = win->h_totals[0], nObs = win->h_totals[1];
    {
        std::lock_guard<std::mutex> lk(win->mu);
        win->lastC = C, win->lastP = P, win->lastObs = nObs;
        win->lastCount = J.winCount;
        for (int j = 0; j < J.winCount; ++j) win->lastFrames[j] = J.winFrames[j];
    }
    b->maxObs = win->h_totals[2];
    b->havePairs = false;
    if (P == 0 || nObs == 0) {
        cs_set_error("cs_ba_solve_window_async: no map point has two feature points in the window");
        return CS_ERR_INVALID;
    }
    // lane plan of the packed kernels (as cs_ba_upload builds it): whole points back to back, at most 64 measurements per wave
    b->nPackWaves = 0;
    if (b->maxObs <= 64) {
        if ((size_t)win->nMap + 2 > b->waveStartCap) {
            if (b->waveStart) (void)hipFree(b->waveStart);
            b->waveStart = nullptr;
            b->waveStartCap = 0;
            CS_HIP(hipMalloc((void**)&b->waveStart, sizeof(int) * ((size_t)win->nMap + 2)));
            b->waveStartCap = (size_t)win->nMap + 2;
        }
        int* ws = win->h_plan;
        int nw = 0, fill = 0;
        ws[nw++] = 0;
        for (int i = 0; i < P; ++i) {
            const int k = h_optr[i + 1] - h_optr[i];
            if (k == 0) continue;
            if (fill + k > 64) {
                ws[nw++] = h_optr[i];
                fill = 0;
            }
            fill += k;
        }
        ws[nw++] = nObs;
        CS_HIP(hipMemcpyAsync(b->waveStart, ws, sizeof(int) * nw, hipMemcpyHostToDevice, s));  // (h_plan is pinned and outlives the copy)
        b->nPackWaves = nw - 1;
    }
    const size_t nEnt = (size_t)win->h_totals[4];
    if (nEnt > 0 && nEnt <= ((size_t)8 << 20)) {
        if (nEnt > b->pairEntCap) {
            // sized ONCE for the largest list this path accepts (128 MB of 288 GB): a later, larger window never frees -- hipFree
            // synchronises the device, which would wait for a pose stream that is itself waiting for this solve (k_ba_output_wait)
            if (b->pairEnt) (void)hipFree(b->pairEnt);
            b->pairEnt = nullptr;
            const size_t cap = (size_t)8 << 20;
            CS_HIP(hipMalloc((void**)&b->pairEnt, sizeof(int4) * cap));
            b->pairEntCap = cap;
        }
        hipLaunchKernelGGL(k_pairs_fill, dim3(nPairsAll), dim3(64), 0, s, C, P, b->obs_of, b->pairPtr, b->pairEnt);
        b->havePairs = true;
    }
    if (!b->havePairs) {  // (the camera-indexed lists k_schur would need are not built on the device)
        cs_set_error("cs_ba_solve_window_async: %zu camera-pair entries exceed what the window path supports", nEnt);
        return CS_ERR_INVALID;
    }
    // the robust solve in the same segments as cs_ba_solve_async's, the kernels enqueued directly (the sizes change with every
    // key frame: no graph to replay); the estimate the fill kernel wrote into Rs / Ts / pts is the start (the key poses as
    // tracked, the map as it stands)
    BaPlan L;
    rc = ba_make_plan(b, C, P, nObs, J.nCamsCon, J.nPtsCon, J.maxErr, J.innerMaxIter, false, &L);
    if (rc) return rc;
    {
        w->chunk = 2;   // LM steps per segment (measured in the frame loop: 1 -> 1.85, 2 -> 1.81, 3 -> 1.89 ms per joint solve)
        if (w->chunk > J.innerMaxIter && J.innerMaxIter > 0) w->chunk = J.innerMaxIter;
        if (w->chunk < 1) w->chunk = 1;
    }
    const bool stateInKernel = ba_state_in_kernel(w, L);
    const BaDev& D = L.D;
    const dim3 blk(256);
    rc = ba_run_segments(w, s, J.maxIter, J.innerMaxIter, [&](char kind) {
        switch (kind) {
            case 'H':
                ba_enqueue_init(b, s, L, false, nullptr, nullptr, nullptr);
                hipLaunchKernelGGL(k_cost, dim3(L.cb), blk, 0, s, D, 0);
                hipLaunchKernelGGL(k_control, dim3(1), blk, 0, s, D);
                break;
            case 'R':
                hipLaunchKernelGGL(k_cost, dim3(L.cb), blk, 0, s, D, 0);
                hipLaunchKernelGGL(k_control, dim3(1), blk, 0, s, D);
                break;
            case 'C':
                ba_enqueue_lm_run(s, L, w->chunk);
                if (!stateInKernel) (void)hipMemcpyAsync(w->h_state, &b->st->inner_done, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
                break;
            case 'T':
                hipLaunchKernelGGL(k_flag, dim3(L.cb), blk, 0, s, D);
                hipLaunchKernelGGL(k_outer_end, dim3(1), dim3(1), 0, s, D);
                if (!stateInKernel) (void)hipMemcpyAsync(w->h_state, &b->st->inner_done, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
                break;
            default:
                hipLaunchKernelGGL(k_cost_force, dim3(L.cb), blk, 0, s, D);
                hipLaunchKernelGGL(k_finish, dim3(1), blk, 0, s, D, b->stats);
                break;
        }
        return hipGetLastError();
    });
    if (rc) return rc;
    rc = ba_run_followup(b, s);
    if (b->output) {   // RobustBundleRTS::output(), first half: the result into the next record, behind the solve's last kernel
        const int prc = ba_output_pack(b, win, J, s, C, P, nObs, rc == CS_OK);
        if (prc != CS_OK) return rc != CS_OK ? rc : prc;
        *packed = true;
        const hipError_t e = hipStreamSynchronize(s);
        ba_output_publish(b->output);
        if (e != hipSuccess) {
            cs_set_error("cs_ba_solve_window_async: hipStreamSynchronize failed: %s", hipGetErrorString(e));
            return CS_ERR_HIP;
        }
        return rc;
    }
    CS_HIP(hipStreamSynchronize(s));
    return rc;
}

// ---- InterCamPoseEstimator::addMapPoints built on the device (ba_intercam_dev.h) --------------------------------------------------
constexpr int IC_STAGES = 3;
struct cs_ba_intercam {
    int device, nCams, N, ptsStride, nMap, maxDyn;
    int maxP, maxObs;
    IcStage st[IC_STAGES];
    unsigned char* slab;
    int *pairCnt, *pairTotal;
    int *h_totals, *h_plan;   // pinned: [8 + maxP + 1], [maxP + 2]
    std::mutex mu;
    std::condition_variable cv;
    long long issued, consumed;   // staging records handed to requests / copied into a workspace by a worker
    int lastC, lastP, lastObs, lastStatic;
    int* lastPointMap;            // device, maxP: the map index of every point of the last problem solved
};

static int ba_worker_run_intercam(cs_ba* b, BaWorker* w, const BaAsyncJob& J) {
    cs_ba_intercam* ic = J.ic;
    CS_HIP(hipSetDevice(b->device));
    hipStream_t s = b->own_stream;
    CS_HIP(hipStreamWaitEvent(s, J.ready, 0));
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->stale) {
            ba_worker_destroy_graphs(w);
            w->stale = false;
        }
    }
    struct Consumed {   // whatever way this function is left: the staging record is free again
        cs_ba_intercam* ic;
        bool done = false;
        void release() {
            if (done) return;
            done = true;
            {
                std::lock_guard<std::mutex> lk(ic->mu);
                ic->consumed += 1;
            }
            ic->cv.notify_all();
        }
        ~Consumed() { release(); }
    } consumed{ic};
    const int C = ic->nCams;
    int rc = ba_reserve(b, C, ic->maxP, ic->maxObs);
    if (rc) return rc;
    ba_bind_io(b, b->capC, b->capP, b->capObs);
    ba_drop_graph(b);
    const IcStage& st = ic->st[J.icSlot];
    const size_t P1 = (size_t)ic->maxP, O1 = (size_t)ic->maxObs;
    {   // the staged problem into the workspace: one launch (small_ops.h)
        cs_small::List ops;
        ops.copy(b->Ks, st.Ks, 72 * (size_t)C), ops.copy(b->Rs, st.Rs, 72 * (size_t)C), ops.copy(b->Ts, st.Ts, 24 * (size_t)C);
        ops.copy(b->pts, st.pts, 24 * P1), ops.copy(b->obs_xy, st.obs_xy, 16 * O1), ops.copy(b->obs_ptr, st.obs_ptr, 4 * (P1 + 1));
        ops.copy(b->obs_cam, st.obs_cam, 4 * O1), ops.copy(b->obs_pt, st.obs_pt, 4 * O1), ops.copy(b->obs_of, st.obs_of, 4 * P1 * C);
        ops.copy(ic->lastPointMap, st.pointMap, 4 * P1);
        CS_HIP(ops.run(s));
    }
    if (6 * C <= 36)   // (the small-order solver's Schur kernels walk camera-indexed lists)
        hipLaunchKernelGGL(k_cam_lists, dim3(1), dim3(1024), 0, s, C, st.totals, b->obs_cam, b->cam_ptr, b->cam_obs);
    const int nPairsAll = C * (C + 1) / 2;
    if ((size_t)nPairsAll + 1 > b->pairPtrCap) {
        if (b->pairPtr) (void)hipFree(b->pairPtr);
        b->pairPtr = nullptr;
        CS_HIP(hipMalloc((void**)&b->pairPtr, sizeof(int) * ((size_t)nPairsAll + 1)));
        b->pairPtrCap = (size_t)nPairsAll + 1;
    }
    hipLaunchKernelGGL(k_pairs_count, dim3(nPairsAll), dim3(64), 0, s, C, st.totals, b->obs_of, ic->pairCnt);
    hipLaunchKernelGGL(k_pairs_scan, dim3(1), dim3(1024), 0, s, nPairsAll, ic->pairCnt, b->pairPtr, ic->pairTotal);
    CS_HIP(hipMemcpyAsync(ic->h_totals, st.totals, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(ic->h_totals + 4, ic->pairTotal, sizeof(int), hipMemcpyDeviceToHost, s));
    int* h_optr = ic->h_totals + 8;
    CS_HIP(hipMemcpyAsync(h_optr, b->obs_ptr, sizeof(int) * (P1 + 1), hipMemcpyDeviceToHost, s));
    if (w->tmEv[2]) {
        (void)hipEventRecord(w->tmEv[2], s);
        w->parseStamped = true;
    }
    CS_HIP(hipStreamSynchronize(s));
    consumed.release();
    const int P = ic->h_totals[0], nObs = ic->h_totals[1], nStatic = ic->h_totals[3];
    {
        std::lock_guard<std::mutex> lk(ic->mu);
        ic->lastC = C, ic->lastP = P, ic->lastObs = nObs, ic->lastStatic = nStatic;
    }
    b->maxObs = ic->h_totals[2];
    b->havePairs = false;
    if (P == 0 || nObs == 0 || nStatic == 0) {   // (the reference asserts m_numStatic > 0, :93)
        cs_set_error("cs_ba_solve_intercam_async: no static feature point carries a map point");
        return CS_ERR_INVALID;
    }
    b->nPackWaves = 0;
    if (b->maxObs <= 64) {
        if ((size_t)ic->maxP + 2 > b->waveStartCap) {
            if (b->waveStart) (void)hipFree(b->waveStart);
            b->waveStart = nullptr;
            b->waveStartCap = 0;
            CS_HIP(hipMalloc((void**)&b->waveStart, sizeof(int) * ((size_t)ic->maxP + 2)));
            b->waveStartCap = (size_t)ic->maxP + 2;
        }
        int* ws = ic->h_plan;
        int nw = 0, fill = 0;
        ws[nw++] = 0;
        for (int i = 0; i < P; ++i) {
            const int k = h_optr[i + 1] - h_optr[i];
            if (k == 0) continue;
            if (fill + k > 64) {
                ws[nw++] = h_optr[i];
                fill = 0;
            }
            fill += k;
        }
        ws[nw++] = nObs;
        CS_HIP(hipMemcpyAsync(b->waveStart, ws, sizeof(int) * nw, hipMemcpyHostToDevice, s));
        b->nPackWaves = nw - 1;
    }
    const size_t nEnt = (size_t)ic->h_totals[4];
    if (nEnt == 0 || nEnt > ((size_t)8 << 20)) {
        cs_set_error("cs_ba_solve_intercam_async: %zu camera-pair entries", nEnt);
        return CS_ERR_INVALID;
    }
    if (nEnt > b->pairEntCap) {
        // sized once for the largest problem addMapPoints can build (a static point: one entry; a dynamic one: every camera pair),
        // so that no later solve frees (hipFree synchronises the device: see the window path)
        if (b->pairEnt) (void)hipFree(b->pairEnt);
        b->pairEnt = nullptr;
        const size_t bound = (size_t)ic->nCams * ic->ptsStride + (size_t)(ic->maxDyn + 1) * ic->nCams * (ic->nCams + 1) / 2 + 1024;
        const size_t cap = nEnt + nEnt / 4 + 1024 > bound ? nEnt + nEnt / 4 + 1024 : bound;
        CS_HIP(hipMalloc((void**)&b->pairEnt, sizeof(int4) * cap));
        b->pairEntCap = cap;
    }
    hipLaunchKernelGGL(k_pairs_fill, dim3(nPairsAll), dim3(64), 0, s, C, P, b->obs_of, b->pairPtr, b->pairEnt);
    b->havePairs = true;
    BaPlan L;
    rc = ba_make_plan(b, C, P, nObs, 0, nStatic, J.maxErr, J.innerMaxIter, false, &L);   // nCamsCon 0, nPtsCon = m_numStatic (:95)
    if (rc) return rc;
    w->chunk = 5;
    if (w->chunk > J.innerMaxIter && J.innerMaxIter > 0) w->chunk = J.innerMaxIter;
    if (w->chunk < 1) w->chunk = 1;
    const bool stateInKernel = ba_state_in_kernel(w, L);
    const BaDev& D = L.D;
    const dim3 blk(256);
    rc = ba_run_segments(w, s, J.maxIter, J.innerMaxIter, [&](char kind) {
        switch (kind) {
            case 'H':
                ba_enqueue_init(b, s, L, false, nullptr, nullptr, nullptr);
                hipLaunchKernelGGL(k_cost, dim3(L.cb), blk, 0, s, D, 0);
                hipLaunchKernelGGL(k_control, dim3(1), blk, 0, s, D);
                break;
            case 'R':
                hipLaunchKernelGGL(k_cost, dim3(L.cb), blk, 0, s, D, 0);
                hipLaunchKernelGGL(k_control, dim3(1), blk, 0, s, D);
                break;
            case 'C':
                ba_enqueue_lm_run(s, L, w->chunk);
                if (!stateInKernel) (void)hipMemcpyAsync(w->h_state, &b->st->inner_done, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
                break;
            case 'T':
                hipLaunchKernelGGL(k_flag, dim3(L.cb), blk, 0, s, D);
                hipLaunchKernelGGL(k_outer_end, dim3(1), dim3(1), 0, s, D);
                if (!stateInKernel) (void)hipMemcpyAsync(w->h_state, &b->st->inner_done, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
                break;
            default:
                hipLaunchKernelGGL(k_cost_force, dim3(L.cb), blk, 0, s, D);
                hipLaunchKernelGGL(k_finish, dim3(1), blk, 0, s, D, b->stats);
                break;
        }
        return hipGetLastError();
    });
    if (rc) return rc;
    rc = ba_run_followup(b, s);
    CS_HIP(hipStreamSynchronize(s));
    return rc;
}

static int ba_worker_run_inner(cs_ba* b, BaWorker* w, const BaAsyncJob& J);
// (cs_ba_worker_stats: how long the job held the workspace's stream, from the moment the work it waits for was done)
static int ba_worker_run(cs_ba* b, BaWorker* w, const BaAsyncJob& J) {
    (void)hipSetDevice(b->device);
    if (!w->tmEv[0]) {
        (void)hipEventCreate(&w->tmEv[0]);
        (void)hipEventCreate(&w->tmEv[1]);
        (void)hipEventCreate(&w->tmEv[2]);
    }
    w->parseStamped = false;
    (void)hipStreamWaitEvent(b->own_stream, J.ready, 0);
    (void)hipEventRecord(w->tmEv[0], b->own_stream);
    const int rc = ba_worker_run_inner(b, w, J);
    (void)hipEventRecord(w->tmEv[1], b->own_stream);
    if (hipEventSynchronize(w->tmEv[1]) == hipSuccess) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, w->tmEv[0], w->tmEv[1]) == hipSuccess) {
            float pms = 0;
            if (w->parseStamped && hipEventElapsedTime(&pms, w->tmEv[0], w->tmEv[2]) != hipSuccess) pms = 0;
            std::lock_guard<std::mutex> lk(w->mu);
            w->gpuMsParse += pms;
            w->jobsDone += 1;
            w->gpuMsTotal += ms;
            w->gpuMsLast = ms;
            if (ms > w->gpuMsMax) w->gpuMsMax = ms;
        }
    }
    return rc;
}
static int ba_worker_run_inner(cs_ba* b, BaWorker* w, const BaAsyncJob& J) {
    if (J.win) return ba_worker_run_window(b, w, J);
    if (J.ic) return ba_worker_run_intercam(b, w, J);
    CS_HIP(hipSetDevice(b->device));
    hipStream_t s = b->own_stream;
    CS_HIP(hipStreamWaitEvent(s, J.ready, 0));
    cs_ba::GraphKey key = {J.C, J.P, J.nObs, J.nCamsCon, J.nPtsCon, J.maxIter, J.innerMaxIter, J.maxErr, J.R0, J.T0, J.M0};
    {
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->stale) {
            ba_worker_destroy_graphs(w);
            w->stale = false;
        }
    }
    BaPlan L;
    int rc = ba_make_plan(b, J.C, J.P, J.nObs, J.nCamsCon, J.nPtsCon, J.maxErr, J.innerMaxIter, false, &L);
    if (rc) return rc;
    const bool stateInKernel = ba_state_in_kernel(w, L);
    const BaDev& D = L.D;
    const dim3 blk(256);
    if (!w->haveGraphs || memcmp(&key, &w->key, sizeof(key)) != 0) {
        ba_worker_destroy_graphs(w);
        w->chunk = 5;   // LM steps per captured segment
        if (w->chunk > J.innerMaxIter && J.innerMaxIter > 0) w->chunk = J.innerMaxIter;
        // head: initial estimate into the workspace, cost and LM state of the first round
        rc = ba_capture(s, &w->gHead, [&] {
            ba_enqueue_init(b, s, L, false, J.R0, J.T0, J.M0);
            hipLaunchKernelGGL(k_cost, dim3(L.cb), blk, 0, s, D, 0);
            hipLaunchKernelGGL(k_control, dim3(1), blk, 0, s, D);
        });
        if (rc) return rc;
        rc = ba_capture(s, &w->gChunk, [&] {
            ba_enqueue_lm_run(s, L, w->chunk);  // (the state word read back below is exact at every chunk boundary)
            if (!stateInKernel) (void)hipMemcpyAsync(w->h_state, &b->st->inner_done, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
        });
        if (rc) return rc;
        // tail of a round: outlier flags, round bookkeeping; start of the next round: cost + LM state
        rc = ba_capture(s, &w->gTail, [&] {
            hipLaunchKernelGGL(k_flag, dim3(L.cb), blk, 0, s, D);
            hipLaunchKernelGGL(k_outer_end, dim3(1), dim3(1), 0, s, D);
            if (!stateInKernel) (void)hipMemcpyAsync(w->h_state, &b->st->inner_done, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
        });
        if (rc) return rc;
        rc = ba_capture(s, &w->gRound, [&] {
            hipLaunchKernelGGL(k_cost, dim3(L.cb), blk, 0, s, D, 0);
            hipLaunchKernelGGL(k_control, dim3(1), blk, 0, s, D);
        });
        if (rc) return rc;
        rc = ba_capture(s, &w->gFinish, [&] {
            hipLaunchKernelGGL(k_cost_force, dim3(L.cb), blk, 0, s, D);
            hipLaunchKernelGGL(k_finish, dim3(1), blk, 0, s, D, b->stats);
        });
        if (rc) return rc;
        w->key = key;
        w->haveGraphs = true;
    }
    rc = ba_run_segments(w, s, J.maxIter, J.innerMaxIter, [&](char kind) {
        hipGraphExec_t g = kind == 'H' ? w->gHead : kind == 'R' ? w->gRound : kind == 'C' ? w->gChunk : kind == 'T' ? w->gTail : w->gFinish;
        return hipGraphLaunch(g, s);
    });
    if (rc) return rc;
    rc = ba_run_followup(b, s);  // the finish segment is on the stream: RobustBundleRTS::output()'s non-key-frame update goes here
    CS_HIP(hipStreamSynchronize(s));
    return rc;
}

static void ba_worker_main(cs_ba* b, BaWorker* w) {
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->tid = std::this_thread::get_id();
    }
    for (;;) {
        BaAsyncJob J;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->stop || !w->q.empty(); });
            if (w->q.empty()) return;  // stop requested and nothing left
            J = w->q.front();
            w->q.pop_front();
        }
        const int rc = ba_worker_run(b, w, J);
        (void)hipEventDestroy(J.ready);
        {
            std::lock_guard<std::mutex> lk(w->mu);
            if (rc != CS_OK) {
                w->lastRc = rc;
                snprintf(w->err, sizeof(w->err), "%s", cs_last_error());
            }
            w->inflight -= 1;
            w->jobsCompleted += 1;
        }
        w->cvDone.notify_all();
    }
}

// diagnostics: jobs completed by the workspace's worker and the time they held its stream (GPU clock, ms); resets the sums
extern "C" int cs_ba_worker_stats(cs_ba* b, int* jobs, double* gpu_ms_total, double* gpu_ms_last, double* gpu_ms_max, double* gpu_ms_parse) {
    if (!b) return CS_ERR_INVALID;
    BaWorker* w = b->worker;
    if (jobs) *jobs = 0;
    if (gpu_ms_total) *gpu_ms_total = 0;
    if (gpu_ms_last) *gpu_ms_last = 0;
    if (gpu_ms_max) *gpu_ms_max = 0;
    if (gpu_ms_parse) *gpu_ms_parse = 0;
    if (!w) return CS_OK;
    std::lock_guard<std::mutex> lk(w->mu);
    if (jobs) *jobs = w->jobsDone;
    if (gpu_ms_total) *gpu_ms_total = w->gpuMsTotal;
    if (gpu_ms_last) *gpu_ms_last = w->gpuMsLast;
    if (gpu_ms_max) *gpu_ms_max = w->gpuMsMax;
    if (gpu_ms_parse) *gpu_ms_parse = w->gpuMsParse;
    w->jobsDone = 0;
    w->gpuMsTotal = w->gpuMsMax = w->gpuMsParse = 0;
    return CS_OK;
}

static void ba_worker_stop(cs_ba* b) {
    BaWorker* w = b->worker;
    if (!w) return;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    ba_worker_destroy_graphs(w);
    if (w->h_state) (void)hipHostFree(w->h_state);
    for (hipEvent_t e : w->ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : w->tmEv)
        if (e) (void)hipEventDestroy(e);
    delete w;
    b->worker = nullptr;
}

extern "C" {

cs_ba* cs_ba_create(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        cs_set_error("cs_ba_create: no usable HIP device %d; there is no CPU fallback", device);
        return nullptr;
    }
    cs_ba* b = new cs_ba();
    memset(b, 0, sizeof(*b));
    b->device = device;
    hipError_t se = hipSetDevice(device);
    if (se == hipSuccess) se = hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        cs_set_error("cs_ba_create: cannot create a stream");
        delete b;
        return nullptr;
    }
    b->ownStreamIsOurs = true;
    return b;
}

void cs_ba_destroy(cs_ba* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    ba_worker_stop(b);
    (void)hipStreamSynchronize(b->own_stream);
    ba_free(b);
    if (b->ownStreamIsOurs) (void)hipStreamDestroy(b->own_stream);
    delete b;
}

// The workspace's own stream: where the host-pointer entry points, cs_ba_solve_dev(NULL stream) and the asynchronous worker
// enqueue.
void* cs_ba_stream(cs_ba* b) { return b ? (void*)b->own_stream : nullptr; }

// Replace it by the caller's stream -- e.g. one confined to a CU range (cs_stream_create_cu_range), so that the solve's short
// dependent kernels never queue behind the per-frame streams' workgroups.  The caller keeps ownership of `hip_stream` (it must
// outlive the workspace or the next cs_ba_set_stream); NULL restores a plain stream of the workspace's own.
int cs_ba_set_stream(cs_ba* b, void* hip_stream) {
    if (!b) {
        cs_set_error("cs_ba_set_stream: null workspace");
        return CS_ERR_INVALID;
    }
    const int wrc = cs_ba_wait(b);
    if (wrc) return wrc;
    CS_HIP(hipSetDevice(b->device));
    CS_HIP(hipStreamSynchronize(b->own_stream));
    ba_drop_graph(b);  // (graphs replay on whatever stream they are launched on, but the worker's were captured on the old one)
    if (b->ownStreamIsOurs) (void)hipStreamDestroy(b->own_stream);
    if (hip_stream) {
        b->own_stream = (hipStream_t)hip_stream;
        b->ownStreamIsOurs = false;
    } else {
        CS_HIP(hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking));
        b->ownStreamIsOurs = true;
    }
    return CS_OK;
}

int cs_ba_robust_h(cs_ba* b, int C, int P, int nObs, const double* Ks, double* Rs, double* Ts, double* pts,
                   const int* obs_ptr, const int* obs_cam, const double* obs_xy, int nCamsCon, int nPtsCon,
                   double maxErr, int maxIter, int innerMaxIter, int* out_outlier, cs_ba_stats* stats) {
    if (!b || C <= 0 || P < 0 || nObs < 0 || !Ks || !Rs || !Ts || (P > 0 && (!pts || !obs_ptr)) ||
        (nObs > 0 && (!obs_cam || !obs_xy)) || maxIter < 0 || innerMaxIter < 0) {
        cs_set_error("cs_ba_robust: bad arguments");
        return CS_ERR_INVALID;
    }
    if (P > 0 && (obs_ptr[0] != 0 || obs_ptr[P] != nObs)) {
        cs_set_error("cs_ba_robust: obs_ptr must start at 0 and end at nObs");
        return CS_ERR_INVALID;
    }
    if (nCamsCon < 0 || nPtsCon < 0) {
        cs_set_error("cs_ba_robust: nCamsCon / nPtsCon must not be negative");
        return CS_ERR_INVALID;
    }
    {   // an asynchronous solve of this workspace may still be queued or running: it reads what this call rewrites
        const int wrc = cs_ba_wait(b);
        if (wrc) return wrc;
    }
    {
        // obs_ptr monotone; a point must not carry two measurements of the same view (the (point, view) table keeps one
        // entry per pair: U_j would count both while the Schur terms count one -- an inconsistent reduced system)
        std::vector<int> seen((size_t)C, 0);
        for (int i = 0; i < P; ++i) {
            if (obs_ptr[i + 1] < obs_ptr[i]) {
                cs_set_error("cs_ba_robust: obs_ptr is not monotone at point %d", i);
                return CS_ERR_INVALID;
            }
            for (int o = obs_ptr[i]; o < obs_ptr[i + 1]; ++o) {
                const int v = obs_cam[o];
                if (v < 0 || v >= C) continue;  // reported below
                if (seen[v] == i + 1) {
                    cs_set_error("cs_ba_robust: point %d has two measurements with viewId %d", i, v);
                    return CS_ERR_INVALID;
                }
                seen[v] = i + 1;
            }
        }
    }
    CS_HIP(hipSetDevice(b->device));
    ba_drop_graph(b);  // cached graphs bake in the plan derived from the uploaded topology (e.g. the seg8 kernels)
    int rc = ba_reserve(b, C, P, nObs);
    if (rc) return rc;
    ba_bind_io(b, C, P, nObs);
    const BaIoLayout L = ba_io_layout((size_t)C, (size_t)P, (size_t)nObs);
    // index by camera (counting sort; also validates the view ids), straight into the pinned block
    int* h_cam_ptr = (int*)(b->h_io + L.cptr);
    int* h_cam_obs = (int*)(b->h_io + L.cobs);
    for (int j = 0; j <= C; ++j) h_cam_ptr[j] = 0;
    for (int o = 0; o < nObs; ++o) {
        if (obs_cam[o] < 0 || obs_cam[o] >= C) {
            cs_set_error("cs_ba_robust: measurement %d has viewId %d outside [0,%d)", o, obs_cam[o], C);
            return CS_ERR_INVALID;
        }
        h_cam_ptr[obs_cam[o] + 1]++;
    }
    for (int j = 0; j < C; ++j) h_cam_ptr[j + 1] += h_cam_ptr[j];
    {
        int* fill = new int[C > 0 ? C : 1];
        for (int j = 0; j < C; ++j) fill[j] = h_cam_ptr[j];
        for (int o = 0; o < nObs; ++o) h_cam_obs[fill[obs_cam[o]]++] = o;
        delete[] fill;
    }
    memcpy(b->h_io + L.Ks, Ks, sizeof(double) * 9 * C);
    memcpy(b->h_io + L.Rs, Rs, sizeof(double) * 9 * C);
    memcpy(b->h_io + L.Ts, Ts, sizeof(double) * 3 * C);
    if (P > 0) {
        memcpy(b->h_io + L.pts, pts, sizeof(double) * 3 * P);
        memcpy(b->h_io + L.optr, obs_ptr, sizeof(int) * (P + 1));
    }
    b->maxObs = 0;
    for (int i = 0; i < P; ++i)
        if (obs_ptr[i + 1] - obs_ptr[i] > b->maxObs) b->maxObs = obs_ptr[i + 1] - obs_ptr[i];
    if (nObs > 0) {
        memcpy(b->h_io + L.xy, obs_xy, sizeof(double) * 2 * nObs);
        memcpy(b->h_io + L.ocam, obs_cam, sizeof(int) * nObs);
    }
    hipStream_t s = b->own_stream;
    // Lane plan of the packed kernels (ba_packed_dev.h): whole points back to back, at most 64 measurements per wave, in point
    // order -- a wave is then a contiguous range of the measurement arrays.  Not for a point with more than 64 measurements.
    b->nPackWaves = 0;
    if (maxIter == 0 && nObs > 0 && b->maxObs <= 64) {
        std::vector<int> ws;
        ws.push_back(0);
        int fill = 0;
        for (int i = 0; i < P; ++i) {
            const int k = obs_ptr[i + 1] - obs_ptr[i];
            if (k == 0) continue;
            if (fill + k > 64) {
                ws.push_back(obs_ptr[i]);
                fill = 0;
            }
            fill += k;
        }
        ws.push_back(nObs);
        if (ws.size() > b->waveStartCap) {
            if (b->waveStart) (void)hipFree(b->waveStart);
            b->waveStart = nullptr;
            b->waveStartCap = 0;
            CS_HIP(hipMalloc((void**)&b->waveStart, sizeof(int) * ws.size()));
            b->waveStartCap = ws.size();
        }
        CS_HIP(hipMemcpyAsync(b->waveStart, ws.data(), sizeof(int) * ws.size(), hipMemcpyHostToDevice, s));
        CS_HIP(hipStreamSynchronize(s));  // (the vector goes out of scope)
        b->nPackWaves = (int)ws.size() - 1;
    }
    // Camera-pair lists for k_schur_pairs: which measurements meet in which block of the reduced system is fixed by the
    // topology, so the kernel's index chain (camera list -> point -> partner measurement, three dependent loads per entry,
    // four in five of them misses) is walked once here instead of once per LM step.  sum_i k_i (k_i + 1) / 2 entries: 152 k
    // for the 8-camera rig's joint local BA; not built beyond 4 M entries (cfg5: 36 M), where k_schur's walk is used.
    b->havePairs = false;
    {
        size_t total = 0;
        for (int i = 0; i < P; ++i) {
            const size_t k = (size_t)(obs_ptr[i + 1] - obs_ptr[i]);
            total += k * (k + 1) / 2;
        }
        // Only on the upload path (cs_ba_upload: maxIter == 0), i.e. for callers that solve the same topology repeatedly
        // from device memory; a one-shot host call would pay ~1 ms of list building to save ~0.25 ms of LM steps.
        if (maxIter == 0 && total > 0 && total <= ((size_t)4 << 20) && C <= 1024) {
            const size_t nPairsAll = (size_t)C * (C + 1) / 2;
            auto pid = [C](int ca, int cb) { return (size_t)ca * C - (size_t)ca * (ca - 1) / 2 + (size_t)(cb - ca); };
            std::vector<int> ptr(nPairsAll + 1, 0);
            for (int i = 0; i < P; ++i)
                for (int o1 = obs_ptr[i]; o1 < obs_ptr[i + 1]; ++o1)
                    for (int o2 = obs_ptr[i]; o2 < obs_ptr[i + 1]; ++o2)
                        if (obs_cam[o1] <= obs_cam[o2] && (obs_cam[o1] != obs_cam[o2] || o1 == o2)) ptr[pid(obs_cam[o1], obs_cam[o2]) + 1]++;
            for (size_t q = 0; q < nPairsAll; ++q) ptr[q + 1] += ptr[q];
            std::vector<int4> ent((size_t)ptr[nPairsAll]);
            std::vector<int> fill(ptr.begin(), ptr.end() - 1);
            for (int i = 0; i < P; ++i)  // ascending point index inside every list, like the walk it replaces
                for (int o1 = obs_ptr[i]; o1 < obs_ptr[i + 1]; ++o1)
                    for (int o2 = obs_ptr[i]; o2 < obs_ptr[i + 1]; ++o2)
                        if (obs_cam[o1] <= obs_cam[o2] && (obs_cam[o1] != obs_cam[o2] || o1 == o2))
                            ent[(size_t)fill[pid(obs_cam[o1], obs_cam[o2])]++] = make_int4(o1, o2, i, 0);
            if (ptr.size() > b->pairPtrCap) {
                if (b->pairPtr) (void)hipFree(b->pairPtr);
                b->pairPtr = nullptr;
                CS_HIP(hipMalloc((void**)&b->pairPtr, sizeof(int) * ptr.size()));
                b->pairPtrCap = ptr.size();
            }
            if (ent.size() > b->pairEntCap) {
                if (b->pairEnt) (void)hipFree(b->pairEnt);
                b->pairEnt = nullptr;
                CS_HIP(hipMalloc((void**)&b->pairEnt, sizeof(int4) * ent.size()));
                b->pairEntCap = ent.size();
            }
            CS_HIP(hipMemcpyAsync(b->pairPtr, ptr.data(), sizeof(int) * ptr.size(), hipMemcpyHostToDevice, s));
            CS_HIP(hipMemcpyAsync(b->pairEnt, ent.data(), sizeof(int4) * ent.size(), hipMemcpyHostToDevice, s));
            CS_HIP(hipStreamSynchronize(s));  // (the vectors go out of scope)
            b->havePairs = true;
        }
    }
    CS_HIP(hipMemcpyAsync(b->io, b->h_io, L.total, hipMemcpyHostToDevice, s));  // the whole problem in one copy
    rc = ba_enqueue(b, s, C, P, nObs, nCamsCon, nPtsCon, maxErr, maxIter, innerMaxIter);
    if (rc) return rc;
    // Rs | Ts | pts are adjacent: one copy; statistics | outlier flags: one copy
    const size_t outBytes = (L.pts + sizeof(double) * 3 * (size_t)P) - L.Rs;
    CS_HIP(hipMemcpyAsync(b->h_io + L.Rs, b->io + L.Rs, outBytes, hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(b->h_ob, b->ob, 64 + sizeof(int) * (size_t)(nObs > 0 ? nObs : 0), hipMemcpyDeviceToHost, s));
    CS_HIP(hipStreamSynchronize(s));
    memcpy(Rs, b->h_io + L.Rs, sizeof(double) * 9 * C);
    memcpy(Ts, b->h_io + L.Ts, sizeof(double) * 3 * C);
    if (P > 0) memcpy(pts, b->h_io + L.pts, sizeof(double) * 3 * P);
    if (nObs > 0 && out_outlier) memcpy(out_outlier, b->h_ob + 64, sizeof(int) * nObs);
    if (stats) memcpy(stats, b->h_ob, sizeof(cs_ba_stats_dev));
    if (maxIter > 0) return ba_check_flags(((const cs_ba_stats_dev*)b->h_ob)->flags, "cs_ba_robust");
    return CS_OK;
}

// one-shot convenience with a per-thread cached workspace
int cs_ba_robust(int C, int P, int nObs, const double* Ks, double* Rs, double* Ts, double* pts, const int* obs_ptr,
                 const int* obs_cam, const double* obs_xy, int nCamsCon, int nPtsCon, double maxErr, int maxIter,
                 int innerMaxIter, int* out_outlier, cs_ba_stats* stats, int device) {
    static thread_local cs_ba* cached = nullptr;
    if (cached && cached->device != device) {
        cs_ba_destroy(cached);
        cached = nullptr;
    }
    if (!cached) cached = cs_ba_create(device);
    if (!cached) return CS_ERR_NO_DEVICE;
    return cs_ba_robust_h(cached, C, P, nObs, Ks, Rs, Ts, pts, obs_ptr, obs_cam, obs_xy, nCamsCon, nPtsCon, maxErr,
                          maxIter, innerMaxIter, out_outlier, stats);
}

// device-resident: upload once (cs_ba_upload), then re-solve from the stored initial state on any stream
int cs_ba_upload(cs_ba* b, int C, int P, int nObs, const double* Ks, const double* Rs, const double* Ts,
                 const double* pts, const int* obs_ptr, const int* obs_cam, const double* obs_xy) {
    double* R2 = new double[9 * (size_t)C];
    double* T2 = new double[3 * (size_t)C];
    double* M2 = new double[3 * (size_t)(P > 0 ? P : 1)];
    memcpy(R2, Rs, sizeof(double) * 9 * C);
    memcpy(T2, Ts, sizeof(double) * 3 * C);
    if (P > 0) memcpy(M2, pts, sizeof(double) * 3 * P);
    // maxIter = 0: uploads everything and runs no iteration
    int rc = cs_ba_robust_h(b, C, P, nObs, Ks, R2, T2, M2, obs_ptr, obs_cam, obs_xy, 0, 0, 1.0, 0, 0, nullptr, nullptr);
    delete[] R2;
    delete[] T2;
    delete[] M2;
    return rc;
}

/* d_Rs0/d_Ts0/d_pts0: device pointers to the initial estimate (copied into the workspace on the stream);
 * results are left in the workspace and can be fetched with cs_ba_download. */
int cs_ba_solve_dev(cs_ba* b, void* hip_stream, int C, int P, int nObs, const double* d_Rs0, const double* d_Ts0,
                    const double* d_pts0, int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter) {
    if (!b || C > b->capC || P > b->capP || nObs > b->capObs) {
        cs_set_error("cs_ba_solve_dev: workspace not uploaded for this size");
        return CS_ERR_INVALID;
    }
    {   // queued asynchronous solves of this workspace use the same buffers and graph handles
        const int wrc = cs_ba_wait(b);
        if (wrc) return wrc;
    }
    CS_HIP(hipSetDevice(b->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->own_stream;
    const bool useGraph = cs_debug_get(CS_DBG_BA_GRAPHS) != 0;   // (cs_debug_set("ba_graphs", 0): eager launches keep their names under a profiler)
    cs_ba::GraphKey key = {C, P, nObs, nCamsCon, nPtsCon, maxIter, innerMaxIter, maxErr, d_Rs0, d_Ts0, d_pts0};
    if (useGraph && b->gexec && memcmp(&key, &b->gkey, sizeof(key)) == 0) {
        CS_HIP(hipGraphLaunch(b->gexec, s));
        return ba_run_followup(b, s);
    }
    if (useGraph) {
        ba_drop_graph(b);
        CS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    }
    // the measurement tables were built when the problem was uploaded (cs_ba_upload -> cs_ba_robust_h); the initial
    // estimate is copied into the workspace by the solve's first kernel
    int rc = ba_enqueue(b, s, C, P, nObs, nCamsCon, nPtsCon, maxErr, maxIter, innerMaxIter, false, d_Rs0, d_Ts0, d_pts0);
    if (!useGraph) return rc ? rc : ba_run_followup(b, s);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc || e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        if (!rc) cs_set_error("cs_ba_solve_dev: hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return rc ? rc : CS_ERR_HIP;
    }
    e = hipGraphInstantiate(&b->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        b->gexec = nullptr;
        cs_set_error("cs_ba_solve_dev: hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return CS_ERR_HIP;
    }
    memset(&b->gkey, 0, sizeof(b->gkey));
    b->gkey = key;
    CS_HIP(hipGraphLaunch(b->gexec, s));
    return ba_run_followup(b, s);
}

int cs_ba_set_followup(cs_ba* b, cs_ba_followup_fn fn, void* user) {
    if (!b) {
        cs_set_error("cs_ba_set_followup: null workspace");
        return CS_ERR_INVALID;
    }
    const int rc = cs_ba_wait(b);  // no solve of the worker may be between its finish segment and the call
    b->followup = fn;
    b->followupUser = user;
    return rc;
}

int cs_ba_result_buffers(cs_ba* b, double** d_Rs, double** d_Ts, double** d_pts) {
    if (!b || !b->Rs) {
        cs_set_error("cs_ba_result_buffers: workspace not uploaded");
        return CS_ERR_INVALID;
    }
    if (d_Rs) *d_Rs = b->Rs;
    if (d_Ts) *d_Ts = b->Ts;
    if (d_pts) *d_pts = b->pts;
    return CS_OK;
}

// ---- distributed solve: phase API (see k_dist_pack above) --------------------------------------------------------
// The caller (coslam_amd/multicam.py) uploads the replicated problem with cs_ba_upload, calls cs_ba_dist_begin with
// its point slice, then per outer round
//     phase COST0 | all-reduce scal | phase CONTROL0
//     innerMaxIter x ( phase LIN_SCHUR | all-reduce S||rhs | phase SOLVE_UPDATE | all-reduce scal | phase CONTROL1 )
//     phase FLAG | all-reduce scal | phase OUTER_END
// and finally phase FINAL_PREP | all-reduce pts, outlier | phase FINISH.  Everything is enqueued on `stream`.
int cs_ba_dist_begin(cs_ba* b, void* hip_stream, int C, int P, int nObs, const double* d_Rs0, const double* d_Ts0,
                     const double* d_pts0, int nCamsCon, int nPtsCon, double maxErr, int innerMaxIter, int pLo, int pHi,
                     int addLambda) {
    if (!b || C > b->capC || P > b->capP || nObs > b->capObs || pLo < 0 || pHi > P || pLo > pHi) {
        cs_set_error("cs_ba_dist_begin: workspace not uploaded for this size, or bad slice");
        return CS_ERR_INVALID;
    }
    {
        const int wrc = cs_ba_wait(b);
        if (wrc) return wrc;
    }
    CS_HIP(hipSetDevice(b->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->own_stream;
    if (!b->dist) b->dist = new BaPlan();
    int rc = ba_make_plan(b, C, P, nObs, nCamsCon, nPtsCon, maxErr, innerMaxIter, true, b->dist);
    if (rc) return rc;
    b->dist->D.pLo = pLo;
    b->dist->D.pHi = pHi;
    b->dist->D.addLambda = addLambda ? 1 : 0;
    if (d_Rs0) CS_HIP(hipMemcpyAsync(b->Rs, d_Rs0, sizeof(double) * 9 * C, hipMemcpyDeviceToDevice, s));
    if (d_Ts0) CS_HIP(hipMemcpyAsync(b->Ts, d_Ts0, sizeof(double) * 3 * C, hipMemcpyDeviceToDevice, s));
    if (d_pts0 && P > 0) CS_HIP(hipMemcpyAsync(b->pts, d_pts0, sizeof(double) * 3 * P, hipMemcpyDeviceToDevice, s));
    CS_HIP(hipMemsetAsync(b->scal, 0, sizeof(double) * 8, s));
    ba_enqueue_init(b, s, *b->dist);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int cs_ba_dist_phase(cs_ba* b, void* hip_stream, int phase) {
    if (!b || !b->dist) {
        cs_set_error("cs_ba_dist_phase: cs_ba_dist_begin has not been called");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(b->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->own_stream;
    const BaPlan& L = *b->dist;
    const BaDev& D = L.D;
    const dim3 blk(256);
    switch (phase) {
        case CS_BA_PH_COST0:
            hipLaunchKernelGGL(k_cost, dim3(L.cb), blk, 0, s, D, 0);
            hipLaunchKernelGGL(k_dist_pack, dim3(1), blk, 0, s, D, 0);
            break;
        case CS_BA_PH_CONTROL0:
            hipLaunchKernelGGL(k_dist_control, dim3(1), blk, 0, s, D, 0);
            break;
        case CS_BA_PH_LIN_SCHUR:
            if (D.n > 0) (void)hipMemsetAsync(b->S, 0, sizeof(double) * ((size_t)D.n * D.n + D.n), s);
            ba_enqueue_lin_schur(s, L);
            break;
        case CS_BA_PH_SOLVE_UPDATE:
            ba_enqueue_solve_update(s, L);
            hipLaunchKernelGGL(k_dist_pack, dim3(1), blk, 0, s, D, 1);
            break;
        case CS_BA_PH_CONTROL1:
            hipLaunchKernelGGL(k_dist_control, dim3(1), blk, 0, s, D, 1);
            break;
        case CS_BA_PH_FLAG:
            hipLaunchKernelGGL(k_outer_begin, dim3(1), dim3(1), 0, s, D);
            hipLaunchKernelGGL(k_flag, dim3(L.cb), blk, 0, s, D);
            hipLaunchKernelGGL(k_dist_pack, dim3(1), blk, 0, s, D, 2);
            break;
        case CS_BA_PH_OUTER_END:
            hipLaunchKernelGGL(k_dist_outer_end, dim3(1), dim3(1), 0, s, D);
            break;
        case CS_BA_PH_FINAL_PREP: {
            int nn = 3 * D.P > D.nObs ? 3 * D.P : D.nObs;
            if (nn < 1) nn = 1;
            hipLaunchKernelGGL(k_dist_zero_foreign, dim3((nn + 255) / 256), blk, 0, s, D);
            break;
        }
        case CS_BA_PH_FINISH: {  // pts / outlier are complete on every rank now: report the full cost
            BaDev F = D;
            F.pLo = 0;
            F.pHi = D.P;
            hipLaunchKernelGGL(k_cost_force, dim3(L.cb), blk, 0, s, F);
            hipLaunchKernelGGL(k_finish, dim3(1), blk, 0, s, F, b->stats);
            break;
        }
        default:
            cs_set_error("cs_ba_dist_phase: unknown phase %d", phase);
            return CS_ERR_INVALID;
    }
    CS_CHECK_LAUNCH();
    return CS_OK;
}

/* device pointers of the buffers the collectives run on: S || rhs (n_red doubles), scal (4 doubles), pts (3 P doubles),
 * outlier (nObs int32) */
int cs_ba_dist_buffers(cs_ba* b, void** d_S_rhs, int* n_red, void** d_scal, void** d_pts, void** d_outlier) {
    if (!b || !b->dist) {
        cs_set_error("cs_ba_dist_buffers: cs_ba_dist_begin has not been called");
        return CS_ERR_INVALID;
    }
    const BaDev& D = b->dist->D;
    if (d_S_rhs) *d_S_rhs = b->S;
    if (n_red) *n_red = D.n * D.n + D.n;
    if (d_scal) *d_scal = b->scal;
    if (d_pts) *d_pts = b->pts;
    if (d_outlier) *d_outlier = b->outlier;
    return CS_OK;
}

int cs_ba_download(cs_ba* b, int C, int P, int nObs, double* Rs, double* Ts, double* pts, int* out_outlier,
                   cs_ba_stats* stats) {
    if (!b) return CS_ERR_INVALID;
    {
        const int wrc = cs_ba_wait(b);
        if (wrc) return wrc;
    }
    CS_HIP(hipSetDevice(b->device));
    CS_HIP(hipDeviceSynchronize());
    if (Rs) CS_HIP(hipMemcpy(Rs, b->Rs, sizeof(double) * 9 * C, hipMemcpyDeviceToHost));
    if (Ts) CS_HIP(hipMemcpy(Ts, b->Ts, sizeof(double) * 3 * C, hipMemcpyDeviceToHost));
    if (pts && P > 0) CS_HIP(hipMemcpy(pts, b->pts, sizeof(double) * 3 * P, hipMemcpyDeviceToHost));
    if (out_outlier && nObs > 0) CS_HIP(hipMemcpy(out_outlier, b->outlier, sizeof(int) * nObs, hipMemcpyDeviceToHost));
    cs_ba_stats_dev sd;
    CS_HIP(hipMemcpy(&sd, b->stats, sizeof(sd), hipMemcpyDeviceToHost));
    if (stats) memcpy(stats, &sd, sizeof(cs_ba_stats));
    return ba_check_flags(sd.flags, "cs_ba_download");
}

// bundleAdjustRobust as the reference runs it: on a worker thread next to tracking (src/app/SL_CoSLAM.cpp:1702-1784).
// The call records an event on `after_stream` (the solve starts once everything enqueued there so far -- e.g. the
// kernels producing d_Rs0 / d_Ts0 / d_pts0 -- has finished), queues the request for the workspace's thread and returns.
// Requests of one workspace run in order.  cs_ba_wait blocks until all of them are done; cs_ba_download then reads
// the result.  Same arithmetic and results as cs_ba_solve_dev; only the schedule differs (chunks of LM steps with an
// early exit between them instead of the whole schedule up front).
int cs_ba_solve_async(cs_ba* b, void* after_stream, int C, int P, int nObs, const double* d_Rs0, const double* d_Ts0,
                      const double* d_pts0, int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter) {
    if (!b || C > b->capC || P > b->capP || nObs > b->capObs) {
        cs_set_error("cs_ba_solve_async: workspace not uploaded for this size");
        return CS_ERR_INVALID;
    }
    if (nCamsCon < 0 || nPtsCon < 0 || maxIter < 0 || innerMaxIter < 0) {
        cs_set_error("cs_ba_solve_async: negative count");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(b->device));
    if (!b->worker) {
        BaWorker* w = new BaWorker();
        if (hipHostMalloc((void**)&w->h_state, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
            delete w;
            cs_set_error("cs_ba_solve_async: cannot allocate the pinned state word");
            return CS_ERR_ALLOC;
        }
        w->h_state[0] = w->h_state[1] = 0;
        b->worker = w;
        w->th = std::thread(ba_worker_main, b, w);
    }
    BaAsyncJob J;
    J.C = C, J.P = P, J.nObs = nObs, J.nCamsCon = nCamsCon, J.nPtsCon = nPtsCon, J.maxIter = maxIter, J.innerMaxIter = innerMaxIter;
    J.maxErr = maxErr, J.R0 = d_Rs0, J.T0 = d_Ts0, J.M0 = d_pts0, J.ready = nullptr;
    CS_HIP(hipEventCreateWithFlags(&J.ready, hipEventDisableTiming));
    CS_HIP(hipEventRecord(J.ready, (hipStream_t)after_stream));
    {
        std::lock_guard<std::mutex> lk(b->worker->mu);
        b->worker->q.push_back(J);
        b->worker->inflight += 1;
    }
    b->worker->cv.notify_one();
    return CS_OK;
}

// ---- sliding window of key frames: the BA's inputs from the tracker's own records (ba_window_dev.h) ---------------------------
cs_ba_window* cs_ba_window_create(int device, int nCams, int nKeyFrames, int N, int nMapPts) {
    if (nCams < 1 || nCams > 16 || nKeyFrames < 1 || nKeyFrames > 16 || N < 1 || nMapPts < 1) {
        cs_set_error("cs_ba_window_create: need 1..16 cameras, 1..16 key frames, N >= 1, nMapPts >= 1");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        cs_set_error("cs_ba_window_create: no usable HIP device %d", device);
        return nullptr;
    }
    cs_ba_window* w = new cs_ba_window();
    w->device = device, w->nCams = nCams, w->nKf = nKeyFrames, w->N = N, w->nMap = nMapPts;
    w->ring = nKeyFrames + WIN_SLACK;
    w->head = w->count = w->parsesPending = w->lastCount = w->snapNext = 0;
    w->pushCount = 0;
    w->lastC = w->lastP = w->lastObs = 0;
    w->slab = nullptr, w->h_totals = w->h_plan = nullptr;
    const size_t KC = (size_t)w->ring * nCams, KW = (size_t)nKeyFrames * nCams, nPairs = KW * (KW + 1) / 2;
    struct Piece {
        void** ptr;
        size_t bytes;
    };
    const Piece pieces[] = {
        {(void**)&w->xy, sizeof(double) * KC * 2 * N}, {(void**)&w->K, sizeof(double) * KC * 9},
        {(void**)&w->R, sizeof(double) * KC * 9},      {(void**)&w->t, sizeof(double) * KC * 3},
        {(void**)&w->pf, sizeof(int) * KC * nMapPts},  {(void**)&w->cnt, sizeof(int) * nMapPts},
        {(void**)&w->ptIndex, sizeof(int) * nMapPts},  {(void**)&w->obsStart, sizeof(int) * nMapPts},
        {(void**)&w->totals, sizeof(int) * 8},         {(void**)&w->pointMap, sizeof(int) * nMapPts},
        {(void**)&w->pairCnt, sizeof(int) * (nPairs + 1)}, {(void**)&w->pairTotal, sizeof(int) * 8},
        {(void**)&w->mapSnap[0], sizeof(double) * 3 * nMapPts}, {(void**)&w->mapSnap[1], sizeof(double) * 3 * nMapPts},
        {(void**)&w->mapSnap[2], sizeof(double) * 3 * nMapPts}, {(void**)&w->staticSnap[0], (size_t)nMapPts},
        {(void**)&w->staticSnap[1], (size_t)nMapPts},           {(void**)&w->staticSnap[2], (size_t)nMapPts},
        {(void**)&w->poseSnapR[0], sizeof(double) * KC * 9},    {(void**)&w->poseSnapT[0], sizeof(double) * KC * 3},
        {(void**)&w->poseSnapR[1], sizeof(double) * KC * 9},    {(void**)&w->poseSnapT[1], sizeof(double) * KC * 3},
        {(void**)&w->poseSnapR[2], sizeof(double) * KC * 9},    {(void**)&w->poseSnapT[2], sizeof(double) * KC * 3},
    };
    size_t total = 0;
    for (const Piece& q : pieces) total += (q.bytes + 255) & ~(size_t)255;
    if (hipMalloc((void**)&w->slab, total) != hipSuccess || hipHostMalloc((void**)&w->h_totals, (8 + (size_t)nMapPts + 1) * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&w->h_plan, ((size_t)nMapPts + 2) * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        cs_set_error("cs_ba_window_create: cannot allocate %zu MB", total >> 20);
        if (w->slab) (void)hipFree(w->slab);
        delete w;
        return nullptr;
    }
    size_t off = 0;
    for (const Piece& q : pieces) {
        *q.ptr = w->slab + off;
        off += (q.bytes + 255) & ~(size_t)255;
    }
    return w;
}

void cs_ba_window_destroy(cs_ba_window* w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(w->slab);
    (void)hipHostFree(w->h_totals);
    (void)hipHostFree(w->h_plan);
    delete w;
}

// A key frame into the ring (asynchronous on hip_stream): of every camera the hand-back's records of THIS frame -- xy, state,
// slot2map of cams[c] (the cs_handback_cam the hand-back was called with) -- and K, R, t (d_K: nCams x 9 or ONE 9 shared,
// d_R: nCams x 9, d_t: nCams x 3: the poses just solved).  The oldest key frame is overwritten when the ring is full.
int cs_ba_window_push_dev(cs_ba_window* w, void* hip_stream, const cs_handback_cam* cams, const double* d_K, int kShared,
                          const double* d_R, const double* d_t, int frame) {
    if (!w || !cams || !d_K || !d_R || !d_t) {
        cs_set_error("cs_ba_window_push_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(w->device));
    hipStream_t s = (hipStream_t)hip_stream;
    const int slot = w->head;
    {   // the slot about to be rewritten belongs to the window of the request WIN_SLACK + 1 pushes back: wait for its parse
        std::unique_lock<std::mutex> lk(w->mu);
        const long long n = w->pushCount;
        w->cv.wait(lk, [&] { return w->parsesPending <= WIN_SLACK && (w->pendingNewest.empty() || w->pendingNewest.front() >= n - WIN_SLACK); });
        w->pushCount = n + 1;
    }
    const size_t base = (size_t)slot * w->nCams;
    WinSnapArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = w->nCams, A.N = w->N, A.nMap = w->nMap;
    for (int c = 0; c < w->nCams; ++c) {
        if (!cams[c].xy || !cams[c].state || !cams[c].slot2map) {
            cs_set_error("cs_ba_window_push_dev: camera %d has no hand-back records", c);
            return CS_ERR_INVALID;
        }
        A.xy[c] = cams[c].xy, A.state[c] = cams[c].state, A.slot2map[c] = cams[c].slot2map;
    }
    A.xyOut = w->xy + base * 2 * w->N;
    A.pfOut = w->pf + base * w->nMap;
    {   // the slot's feature table cleared, its intrinsics and poses: one launch (small_ops.h)
        cs_small::List ops;
        auto put = [&](void* dst, const void* src, size_t bytes, unsigned v) -> hipError_t {
            if (ops.add(dst, src, bytes, v)) return hipSuccess;
            const hipError_t e = ops.run(s);
            ops.add(dst, src, bytes, v);
            return e;
        };
        CS_HIP(put(A.pfOut, nullptr, sizeof(int) * (size_t)w->nCams * w->nMap, 0xff));
        if (kShared) {
            for (int c = 0; c < w->nCams; ++c) CS_HIP(put(w->K + (base + c) * 9, d_K, 72, 0));
        } else {
            CS_HIP(put(w->K + base * 9, d_K, 72 * (size_t)w->nCams, 0));
        }
        CS_HIP(put(w->R + base * 9, d_R, 72 * (size_t)w->nCams, 0));
        CS_HIP(put(w->t + base * 3, d_t, 24 * (size_t)w->nCams, 0));
        CS_HIP(ops.run(s));
    }
    hipLaunchKernelGGL(k_win_snapshot, dim3((w->N + 255) / 256, w->nCams), dim3(256), 0, s, A);
    CS_CHECK_LAUNCH();
    w->frameOf[slot] = frame;
    w->head = (slot + 1) % w->ring;
    if (w->count < w->nKf) w->count += 1;
    return CS_OK;
}

// bundleAdjustRobust on the window the way RobustBundleRTS::run does it (reference src/app/SL_CoSLAMRobustBA.cpp:170-180): the
// flat problem is parsed on the device from the ring and the map (d_mapPts: nMapPts x 3; d_mapStatic: nMapPts flags or NULL =
// every mapped point is static), then solved in workspace b on its worker thread, started when the work enqueued on
// after_stream so far (the push of the newest key frame) is done.  The result stays in the workspace (cs_ba_result_buffers /
// cs_ba_download with cs_ba_window_last_problem's sizes); cs_ba_wait / cs_ba_download report errors.
static int ba_solve_window_async(cs_ba* b, cs_ba_window* w, void* after_stream, const double* d_mapPts, const unsigned char* d_mapStatic,
                                 int staticIsFlags, int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter);
int cs_ba_solve_window_async(cs_ba* b, cs_ba_window* w, void* after_stream, const double* d_mapPts, const unsigned char* d_mapStatic,
                             int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter) {
    return ba_solve_window_async(b, w, after_stream, d_mapPts, d_mapStatic, 0, nCamsCon, nPtsCon, maxErr, maxIter, innerMaxIter);
}
// the same with the map's CS_MAP_* flag bytes instead of a 0 / 1 table: a point takes part when it isLocalStatic() -- neither
// CS_MAP_DYNAMIC nor CS_MAP_FALSE (RobustBundleRTS::addPoints takes the static points only, src/app/SL_CoSLAMRobustBA.cpp:56-66)
int cs_ba_solve_window_flags_async(cs_ba* b, cs_ba_window* w, void* after_stream, const double* d_mapPts, const unsigned char* d_mapFlags,
                                   int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter) {
    if (!d_mapFlags) {
        cs_set_error("cs_ba_solve_window_flags_async: null flags");
        return CS_ERR_INVALID;
    }
    return ba_solve_window_async(b, w, after_stream, d_mapPts, d_mapFlags, 1, nCamsCon, nPtsCon, maxErr, maxIter, innerMaxIter);
}
static int ba_solve_window_async(cs_ba* b, cs_ba_window* w, void* after_stream, const double* d_mapPts, const unsigned char* d_mapStatic,
                                 int staticIsFlags, int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter) {
    if (!b || !w || !d_mapPts || nCamsCon < 0 || nPtsCon < 0 || maxIter < 0 || innerMaxIter < 0 || b->device != w->device) {
        cs_set_error("cs_ba_solve_window_async: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(b->device));
    if (!b->worker) {
        BaWorker* wk = new BaWorker();
        if (hipHostMalloc((void**)&wk->h_state, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
            delete wk;
            cs_set_error("cs_ba_solve_window_async: cannot allocate the pinned state word");
            return CS_ERR_ALLOC;
        }
        wk->h_state[0] = wk->h_state[1] = 0;
        b->worker = wk;
        wk->th = std::thread(ba_worker_main, b, wk);
    }
    BaAsyncJob J;
    J.C = J.P = J.nObs = 0;
    J.nCamsCon = nCamsCon, J.nPtsCon = nPtsCon, J.maxIter = maxIter, J.innerMaxIter = innerMaxIter, J.maxErr = maxErr;
    J.R0 = J.T0 = J.M0 = nullptr;
    // the map as it stands NOW, in after_stream's order (RobustBundleRTS::addPoints copies the points when the BA is requested,
    // under the BA mutex: src/app/SL_CoSLAMRobustBA.cpp:56-78, SL_CoSLAM.cpp:1731-1784): the frame loop goes on refining the map
    // (cs_pose_update_frame_dev) while the worker parses
    const int sn = w->snapNext;
    w->snapNext = (sn + 1) % (WIN_SLACK + 1);
    // ... and the ring's key poses: cs_ba_output_apply_dev writes a finished solve's key poses back into the ring (the next
    // window starts from them), possibly while this request is still waiting for its parse.  All four in one launch (small_ops.h).
    const size_t KCs = (size_t)w->ring * w->nCams;
    {
        cs_small::List ops;
        ops.copy(w->mapSnap[sn], d_mapPts, sizeof(double) * 3 * (size_t)w->nMap);
        if (d_mapStatic) ops.copy(w->staticSnap[sn], d_mapStatic, (size_t)w->nMap);
        ops.copy(w->poseSnapR[sn], w->R, sizeof(double) * 9 * KCs);
        ops.copy(w->poseSnapT[sn], w->t, sizeof(double) * 3 * KCs);
        CS_HIP(ops.run((hipStream_t)after_stream));
    }
    if (d_mapStatic) {
        if (staticIsFlags) {
            hipLaunchKernelGGL(k_win_static_from_flags, dim3((w->nMap + 255) / 256), dim3(256), 0, (hipStream_t)after_stream, w->nMap, w->staticSnap[sn]);
            CS_CHECK_LAUNCH();
        }
    }
    J.winR = w->poseSnapR[sn], J.winT = w->poseSnapT[sn];
    J.win = w, J.d_map = w->mapSnap[sn], J.d_mapStatic = d_mapStatic ? w->staticSnap[sn] : nullptr;
    J.winCount = w->count;
    for (int j = 0; j < w->count; ++j) {
        J.winSlotOf[j] = (w->head - w->count + j + 2 * w->ring) % w->ring;
        J.winFrames[j] = w->frameOf[J.winSlotOf[j]];
    }
    CS_HIP(hipEventCreateWithFlags(&J.ready, hipEventDisableTiming));
    CS_HIP(hipEventRecord(J.ready, (hipStream_t)after_stream));
    {   // (behind everything that can fail: a request that is counted is a request the worker will release)
        std::lock_guard<std::mutex> lk(w->mu);
        w->parsesPending += 1;
        w->pendingNewest.push_back(w->pushCount - 1);
        J.winNewest = w->pushCount - 1;
    }
    {
        std::lock_guard<std::mutex> lk(b->worker->mu);
        b->worker->q.push_back(J);
        b->worker->inflight += 1;
    }
    b->worker->cv.notify_one();
    return CS_OK;
}

// Size and bind the workspace for the largest problem the window can produce NOW (cs_ba_solve_window_async does it on first
// use): afterwards cs_ba_result_buffers' addresses stay valid across the window's solves -- what a follow-up record
// (cs_posegraph_after_ba_rec) needs before the first solve has run.
int cs_ba_reserve_for_window(cs_ba* b, cs_ba_window* w) {
    if (!b || !w || b->device != w->device) {
        cs_set_error("cs_ba_reserve_for_window: bad arguments");
        return CS_ERR_INVALID;
    }
    const int wrc = cs_ba_wait(b);
    if (wrc) return wrc;
    CS_HIP(hipSetDevice(b->device));
    ba_drop_graph(b);
    const int rc = ba_reserve(b, w->nKf * w->nCams, w->nMap, w->nKf * w->nCams * w->N);
    if (rc) return rc;
    ba_bind_io(b, b->capC, b->capP, b->capObs);
    return CS_OK;
}

// sizes of the problem the last cs_ba_solve_window_async of this window built, the map index of every point (device, P ints:
// RobustBundleRTS::int2MapPt) and the frame number of every key frame (host, oldest first).  Call after cs_ba_wait.
int cs_ba_window_last_problem(cs_ba_window* w, int* C, int* P, int* nObs, const int** d_pointMap, int* keyFrames /* [nKeyFrames] or NULL */) {
    if (!w) return CS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(w->mu);
    if (C) *C = w->lastC;
    if (P) *P = w->lastP;
    if (nObs) *nObs = w->lastObs;
    if (d_pointMap) *d_pointMap = w->pointMap;
    if (keyFrames)
        for (int j = 0; j < w->nKf; ++j) keyFrames[j] = j < w->lastCount ? w->lastFrames[j] : -1;
    return CS_OK;
}

// device addresses of the flat problem in the workspace (what parseInputs produced / cs_ba_upload stored): Ks [C][9],
// obs_ptr [P + 1], obs_cam [nObs], obs_xy [nObs][2]
int cs_ba_problem_buffers(cs_ba* b, const double** d_Ks, const int** d_obs_ptr, const int** d_obs_cam, const double** d_obs_xy) {
    if (!b || !b->Ks) {
        cs_set_error("cs_ba_problem_buffers: workspace holds no problem");
        return CS_ERR_INVALID;
    }
    if (d_Ks) *d_Ks = b->Ks;
    if (d_obs_ptr) *d_obs_ptr = b->obs_ptr;
    if (d_obs_cam) *d_obs_cam = b->obs_cam;
    if (d_obs_xy) *d_obs_xy = b->obs_xy;
    return CS_OK;
}

// Where the workspace's asynchronous solves stand, without blocking: cs_ba_pending = queued or running (0 = the last result is
// final), cs_ba_completed = finished since the workspace was created (the worker has synchronised with the solve's last kernel and
// its follow-up before it counts one).  A frame loop whose host runs ahead of the device always has the NEXT solve queued, so it
// watches the completed count to learn that a result is there (the reference's BA thread calls output() itself, under the lock it
// shares with the tracking thread: src/app/SL_CoSLAM.cpp:1713-1720).
int cs_ba_pending(cs_ba* b) {
    if (!b) return CS_ERR_INVALID;
    BaWorker* w = b->worker;
    if (!w) return 0;
    std::lock_guard<std::mutex> lk(w->mu);
    return w->inflight;
}
long long cs_ba_completed(cs_ba* b) {
    if (!b) return -1;
    BaWorker* w = b->worker;
    if (!w) return 0;
    std::lock_guard<std::mutex> lk(w->mu);
    return w->jobsCompleted;
}

int cs_ba_wait(cs_ba* b) {
    if (!b) return CS_ERR_INVALID;
    BaWorker* w = b->worker;
    if (!w) return CS_OK;
    std::unique_lock<std::mutex> lk(w->mu);
    w->cvDone.wait(lk, [&] { return w->inflight == 0; });
    const int rc = w->lastRc;
    if (rc != CS_OK) cs_set_error("cs_ba_solve_async: %s", w->err);
    w->lastRc = CS_OK;
    return rc;
}



// ---- RobustBundleRTS::output() on the device (ba_output_dev.h) ------------------------------------------------------------------------
cs_ba_output* cs_ba_output_create(int device, int nCams, int nKeyFrames, int nMapPts, int nSlots) {
    if (nCams < 1 || nCams > 16 || nKeyFrames < 1 || nKeyFrames > 16 || nMapPts < 1 || nSlots < 1 || nSlots > 64) {
        cs_set_error("cs_ba_output_create: need 1..16 cameras, 1..16 key frames, nMapPts >= 1, 1..64 slots");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        cs_set_error("cs_ba_output_create: no usable HIP device %d", device);
        return nullptr;
    }
    cs_ba_output* o = new cs_ba_output();
    o->device = device, o->nCams = nCams, o->nKf = nKeyFrames, o->nMap = nMapPts, o->nSlots = nSlots;
    o->L = bo_layout(nCams * nKeyFrames, nMapPts);
    o->issued = o->packed = 0;
    o->graph = nullptr, o->graphNodes = o->maxNodes = 0;
    for (int j = 0; j < 16; ++j) o->graphKeyNode[j] = -1;
    o->nodeR = o->nodeT = o->newR = o->newT = o->edgeR = o->edgeT = nullptr;
    o->scratch = nullptr, o->slab = nullptr, o->d_err = nullptr;
    if (hipMalloc((void**)&o->slab, o->L.bytes * nSlots) != hipSuccess || hipMalloc((void**)&o->d_err, 2 * sizeof(int)) != hipSuccess) {
        cs_set_error("cs_ba_output_create: cannot allocate %zu KB", (o->L.bytes * nSlots) >> 10);
        delete o;
        return nullptr;
    }
    (void)hipMemset(o->slab, 0, o->L.bytes * nSlots);   // (hdr[6] = 0: an unwritten record applies nothing)
    (void)hipMemset(o->d_err, 0, 2 * sizeof(int));
    o->applyMask = CS_BA_APPLY_ALL;
    return o;
}

void cs_ba_output_destroy(cs_ba_output* o) {
    if (!o) return;
    (void)hipSetDevice(o->device);
    (void)hipDeviceSynchronize();
    if (o->graph) cs_posegraph_destroy(o->graph);
    (void)hipFree(o->scratch);
    (void)hipFree(o->slab);
    (void)hipFree(o->d_err);
    delete o;
}

// every window solve of workspace b packs its result into o's next record from now on (NULL detaches); waits for queued solves
int cs_ba_output_attach(cs_ba_output* o, cs_ba* b) {
    if (!b || (o && o->device != b->device)) {
        cs_set_error("cs_ba_output_attach: bad arguments");
        return CS_ERR_INVALID;
    }
    const int rc = cs_ba_wait(b);
    if (rc) return rc;
    b->output = o;
    return CS_OK;
}

size_t cs_ba_output_record_bytes(const cs_ba_output* o) { return o ? o->L.bytes : 0; }

long long cs_ba_output_packed(cs_ba_output* o) {
    if (!o) return -1;
    std::lock_guard<std::mutex> lk(o->mu);
    return o->packed;
}

// blocks the calling thread until record number `seq` (0-based, in request order over every workspace attached to o) is complete
// on the device; returns its address.  A record stays valid until nSlots further solves have been packed.
int cs_ba_output_wait(cs_ba_output* o, long long seq, void** d_record) {
    if (!o || seq < 0) {
        cs_set_error("cs_ba_output_wait: bad arguments");
        return CS_ERR_INVALID;
    }
    {
        std::unique_lock<std::mutex> lk(o->mu);
        o->cv.wait(lk, [&] { return o->packed > seq; });
        if (o->issued - seq > o->nSlots) {
            cs_set_error("cs_ba_output_wait: record %lld has been overwritten (%lld issued, %d slots)", seq, o->issued, o->nSlots);
            return CS_ERR_INVALID;
        }
    }
    if (d_record) *d_record = o->slab + (size_t)(seq % o->nSlots) * o->L.bytes;
    return CS_OK;
}

// The same wait ON THE DEVICE: one polling lane is enqueued on hip_stream and the call returns the record's address at once -- work
// enqueued on hip_stream afterwards runs when record `seq` is complete, while the caller's thread goes on enqueueing frames.  The
// solve must be running on a DIFFERENT stream (a workspace's worker).  Gives up after timeoutMs of GPU wall-clock time
// (cs_ba_output_wait_errors counts that; the stream then goes on with whatever the slot holds).  seq must already be REQUESTED.
int cs_ba_output_wait_dev(cs_ba_output* o, long long seq, void* hip_stream, int timeoutMs, void** d_record) {
    if (!o || seq < 0 || !d_record) {
        cs_set_error("cs_ba_output_wait_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(o->device));
    unsigned char* rec = o->slab + (size_t)(seq % o->nSlots) * o->L.bytes;
    hipLaunchKernelGGL(k_ba_output_wait, dim3(1), dim3(1), 0, (hipStream_t)hip_stream, (const int*)rec, (int)seq,
                       (long long)(timeoutMs > 0 ? timeoutMs : 2000) * 100000LL, o->d_err);
    CS_CHECK_LAUNCH();
    *d_record = rec;
    return CS_OK;
}
int cs_ba_output_wait_errors(cs_ba_output* o) {   // synchronises the device; waits that gave up + records refused
    if (!o) return -1;
    int v[2] = {0, 0};
    if (hipSetDevice(o->device) != hipSuccess || hipMemcpy(v, o->d_err, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v[0] + v[1];
}
int cs_ba_output_set_apply_mask(cs_ba_output* o, int mask) {
    if (!o) return CS_ERR_INVALID;
    o->applyMask = mask & CS_BA_APPLY_ALL;
    return CS_OK;
}

// slot `seq` would use, without waiting (the receive buffer of a broadcast on the ranks that did not solve this window)
int cs_ba_output_slot(cs_ba_output* o, long long seq, void** d_record) {
    if (!o || seq < 0 || !d_record) return CS_ERR_INVALID;
    *d_record = o->slab + (size_t)(seq % o->nSlots) * o->L.bytes;
    return CS_OK;
}

// host copy of a record's header (synchronises hip_stream): C, P, nObs, nKf, nCams, seq, ok, and the key frames' numbers
int cs_ba_output_header(cs_ba_output* o, const void* d_record, void* hip_stream, int hdr8[8], int keyFrames[16]) {
    if (!o || !d_record) return CS_ERR_INVALID;
    int h[BO_HDR_INTS];
    CS_HIP(hipSetDevice(o->device));
    CS_HIP(hipMemcpyAsync(h, d_record, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    CS_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    if (hdr8) memcpy(hdr8, h, 8 * sizeof(int));
    if (keyFrames) memcpy(keyFrames, h + 8, 16 * sizeof(int));
    return CS_OK;
}

// device addresses of a record's arrays (tests, a caller that wants the numbers): Rs [C][9], Ts [C][3], pts [P][3], pointMap [P], ptOutlier [P]
int cs_ba_output_arrays(cs_ba_output* o, const void* d_record, const double** d_Rs, const double** d_Ts, const double** d_pts,
                        const int** d_pointMap, const unsigned char** d_ptOutlier) {
    if (!o || !d_record) return CS_ERR_INVALID;
    const unsigned char* r = (const unsigned char*)d_record;
    if (d_Rs) *d_Rs = (const double*)(r + o->L.offRs);
    if (d_Ts) *d_Ts = (const double*)(r + o->L.offTs);
    if (d_pts) *d_pts = (const double*)(r + o->L.offPts);
    if (d_pointMap) *d_pointMap = (const int*)(r + o->L.offMap);
    if (d_ptOutlier) *d_ptOutlier = r + o->L.offOut;
    return CS_OK;
}

static int bo_ensure_graph(cs_ba_output* o, int nNodes, const int* keyNode) {
    bool same = o->graph && o->graphNodes == nNodes;
    for (int j = 0; same && j < o->nKf; ++j) same = o->graphKeyNode[j] == keyNode[j];
    if (same) return CS_OK;
    if (o->graph) cs_posegraph_destroy(o->graph);
    o->graph = nullptr;
    if (nNodes > o->maxNodes) {
        (void)hipFree(o->scratch);
        o->scratch = nullptr;
        const int cap = nNodes + 16;
        const size_t nN = (size_t)o->nCams * cap;
        const size_t bytes = sizeof(double) * nN * (9 + 3) * 3;   // node | new | edge, R and t each
        CS_HIP(hipMalloc((void**)&o->scratch, bytes));
        double* p = (double*)o->scratch;
        o->nodeR = p, p += nN * 9;
        o->nodeT = p, p += nN * 3;
        o->newR = p, p += nN * 9;
        o->newT = p, p += nN * 3;
        o->edgeR = p, p += nN * 9;
        o->edgeT = p;
        o->maxNodes = cap;
    }
    // nCams chains: node i of camera c = frame firstKeyFrame + i; fixed: the window's key frames (constructCameraGraphs, :206-213)
    std::vector<int> nodePtr(o->nCams + 1), edgePtr(o->nCams + 1), id1, id2;
    std::vector<unsigned char> fixed((size_t)o->nCams * nNodes, 0);
    for (int c = 0; c <= o->nCams; ++c) nodePtr[c] = c * nNodes, edgePtr[c] = c * (nNodes - 1);
    for (int c = 0; c < o->nCams; ++c) {
        for (int j = 0; j < o->nKf; ++j)
            if (keyNode[j] < nNodes) fixed[(size_t)c * nNodes + keyNode[j]] = 1;
        for (int i = 0; i + 1 < nNodes; ++i) id1.push_back(i), id2.push_back(i + 1);
    }
    static const int none = 0;
    const int rc = cs_posegraph_create(o->device, o->nCams, nodePtr.data(), edgePtr.data(), fixed.data(), id1.empty() ? &none : id1.data(),
                                       id2.empty() ? &none : id2.data(), &o->graph);
    if (rc != CS_OK) return rc;
    o->graphNodes = nNodes;
    for (int j = 0; j < o->nKf; ++j) o->graphKeyNode[j] = keyNode[j];
    return CS_OK;
}

// RobustBundleRTS::output(), second half, on the stream that owns the map (asynchronous on hip_stream; call it BETWEEN two frames:
// the history's newest entry is the last frame whose pose update ran, d_pointFeat is that frame's hand-back table):
//   constructCameraGraphs over the history's frames firstKeyFrame .. newest (edges from the poses as they stand),
//   the record's key poses into the graphs' fixed nodes (and into window w's ring copies of those key frames, w may be NULL),
//   the record's points into the map, points with an outlier measurement set false,
//   updateNonKeyCameraPoses: relaxed poses back into the history, the newest frame's into d_Rcur / d_tcur (nCams x 9 / 3: the
//   poses the next frame's pose solve starts from),
//   updateNewPosesPoints over the live map (cs_update_new_poses_points_dev).
// The key frames of the record must be firstKeyFrame + j * keyEvery, j < nKeyFrames (the caller knows its own schedule; the
// record's header is not read back) -- or, cs_ba_output_apply_frames_dev, ANY ascending list of frames: the reference's key frames
// fall where CoSLAM::genNewMapPoints' decision puts them (cs_keyframe_ready_dev), not on a fixed cadence.  A record whose solve failed (ok = 0) moves nothing but still relaxes (a no-op up to rounding).
// d_counts [3] or NULL: static / dynamic points re-triangulated, points that became false.
// updateNewPosesPoints of every later apply over feature references (cs_feat_ref_advance_dev keeps the table; NULL: this frame's features)
int cs_ba_output_set_feat_refs(cs_ba_output* o, const void* d_featRef, const unsigned char* d_refStatic) {
    if (!o) {
        cs_set_error("cs_ba_output_set_feat_refs: null handle");
        return CS_ERR_INVALID;
    }
    o->featRef = d_featRef, o->refStatic = d_refStatic;
    return CS_OK;
}
int cs_ba_output_apply_dev(cs_ba_output* o, const void* d_record, void* hip_stream, cs_track_history* h, cs_ba_window* w,
                           const cs_poseupdate_cam* cams, const int* d_pointFeat, int nMap, double* d_mapPts, double* d_mapCov,
                           unsigned char* d_mapFlags, double pixelErrVar, int firstKeyFrame, int keyEvery, double* d_Rcur, double* d_tcur,
                           int* d_counts) {
    return cs_ba_output_apply_seq_dev(o, d_record, -1, hip_stream, h, w, cams, d_pointFeat, nMap, d_mapPts, d_mapCov, d_mapFlags, pixelErrVar,
                                      firstKeyFrame, keyEvery, d_Rcur, d_tcur, d_counts);
}
// ... and with the record's sequence number stated (seq >= 0: the number cs_ba_output_wait(_dev) was asked for, on the rank that
// solved the window; the ranks that received the record by broadcast pass the solving rank's number): a slot that holds ANOTHER
// window's record -- the wait gave up, the broadcast did not arrive -- moves nothing and is counted (cs_ba_output_wait_errors).
int cs_ba_output_apply_seq_dev(cs_ba_output* o, const void* d_record, long long seq, void* hip_stream, cs_track_history* h, cs_ba_window* w,
                               const cs_poseupdate_cam* cams, const int* d_pointFeat, int nMap, double* d_mapPts, double* d_mapCov,
                               unsigned char* d_mapFlags, double pixelErrVar, int firstKeyFrame, int keyEvery, double* d_Rcur,
                               double* d_tcur, int* d_counts) {
    if (!o || keyEvery < 1) {
        cs_set_error("cs_ba_output_apply_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    int frames[16];
    for (int j = 0; j < 16; ++j) frames[j] = firstKeyFrame + j * keyEvery;
    return cs_ba_output_apply_frames_dev(o, d_record, seq, hip_stream, h, w, cams, d_pointFeat, nMap, d_mapPts, d_mapCov, d_mapFlags, pixelErrVar,
                                         frames, o->nKf, d_Rcur, d_tcur, d_counts);
}
// ... and with the window's key frames as a list (keyFrames: HOST array of nKeyFrames = the output's key-frame count, strictly ascending):
// node keyFrames[j] - keyFrames[0] of every camera's chain is fixed at the record's pose j, the frames between and behind them relax.
// The camera graphs are rebuilt (host side: cs_posegraph_create) whenever the spacing or the span differs from the previous apply's.
int cs_ba_output_apply_frames_dev(cs_ba_output* o, const void* d_record, long long seq, void* hip_stream, cs_track_history* h, cs_ba_window* w,
                                  const cs_poseupdate_cam* cams, const int* d_pointFeat, int nMap, double* d_mapPts, double* d_mapCov,
                                  unsigned char* d_mapFlags, double pixelErrVar, const int* keyFrames, int nKeyFrames, double* d_Rcur,
                                  double* d_tcur, int* d_counts) {
    if (!o || !d_record || !h || !cams || !d_pointFeat || nMap != o->nMap || !d_mapPts || !d_mapCov || !d_mapFlags || !keyFrames ||
        nKeyFrames != o->nKf || !d_Rcur || !d_tcur || cs_track_history_cams(h) != o->nCams ||
        (w && (w->nCams != o->nCams || w->device != o->device))) {
        cs_set_error("cs_ba_output_apply_dev: bad arguments (the key frames: a host list of the output's %d)", o ? o->nKf : 0);
        return CS_ERR_INVALID;
    }
    int keyNode[16];
    for (int j = 0; j < o->nKf; ++j) {
        keyNode[j] = keyFrames[j] - keyFrames[0];
        if (j && keyFrames[j] <= keyFrames[j - 1]) {
            cs_set_error("cs_ba_output_apply_dev: key frames must ascend (%d after %d)", keyFrames[j], keyFrames[j - 1]);
            return CS_ERR_INVALID;
        }
    }
    const int firstKeyFrame = keyFrames[0];
    const int newest = cs_track_history_newest_frame(h), nNodes = newest - firstKeyFrame + 1;
    if (nNodes < keyNode[o->nKf - 1] + 1) {
        cs_set_error("cs_ba_output_apply_dev: the history's newest frame %d lies before the window's last key frame %d", newest,
                     keyFrames[o->nKf - 1]);
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(o->device));
    int rc = bo_ensure_graph(o, nNodes, keyNode);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    if ((rc = cs_track_history_get_span_dev(h, hip_stream, firstKeyFrame, nNodes, o->nodeR, o->nodeT))) return rc;
    if (nNodes > 1 && (rc = cs_posegraph_edges_dev(o->graph, hip_stream, o->nodeR, o->nodeT, o->edgeR, o->edgeT))) return rc;
    BoPosesArgs A;
    memset(&A, 0, sizeof(A));
    A.nKf = o->nKf, A.nCams = o->nCams, A.nNodes = nNodes;
    A.win.seq = (int)seq, A.win.nKf = o->nKf;
    for (int j = 0; j < 16; ++j) A.win.kf[j] = j < o->nKf ? keyFrames[j] : -1, A.nodeOf[j] = j < o->nKf ? keyNode[j] : 0;
    A.nodeR = o->nodeR, A.nodeT = o->nodeT;
    for (int j = 0; j < 16; ++j) A.slotOf[j] = -1;
    if (w) {
        A.winR = w->R, A.winT = w->t;
        for (int j = 0; j < o->nKf; ++j)
            for (int k = 0; k < w->count; ++k) {
                const int slot = (w->head - 1 - k + 2 * w->ring) % w->ring;
                if (w->frameOf[slot] == keyFrames[j]) A.slotOf[j] = slot;
            }
    }
    const int mask = o->applyMask;
    if (seq >= 0) hipLaunchKernelGGL(k_ba_output_check, dim3(1), dim3(1), 0, s, (const int*)d_record, A.win, o->d_err);
    if (mask & CS_BA_APPLY_POSES)
        hipLaunchKernelGGL(k_ba_output_poses, dim3((o->nKf * o->nCams * 12 + 255) / 256), dim3(256), 0, s, (const unsigned char*)d_record, o->L, A);
    {
        cs_small::List ops;
        if (d_counts) ops.fill(d_counts + 2, 0, sizeof(int));
        if (nNodes <= 1) ops.copy(o->newR, o->nodeR, sizeof(double) * 9 * o->nCams), ops.copy(o->newT, o->nodeT, sizeof(double) * 3 * o->nCams);
        CS_HIP(ops.run(s));
    }
    if (mask & (CS_BA_APPLY_POINTS | CS_BA_APPLY_FALSE))
        hipLaunchKernelGGL(k_ba_output_points, dim3((o->L.maxP + 255) / 256), dim3(256), 0, s, (const unsigned char*)d_record, o->L, nMap, d_mapPts,
                           d_mapFlags, d_counts ? d_counts + 2 : nullptr, A.win, (mask & CS_BA_APPLY_POINTS) ? 1 : 0,
                           (mask & CS_BA_APPLY_FALSE) ? 1 : 0);
    CS_CHECK_LAUNCH();
    if (nNodes > 1) {
        if ((rc = cs_posegraph_relax_dev(o->graph, hip_stream, o->nodeR, o->nodeT, o->edgeR, o->edgeT, o->newR, o->newT))) return rc;
    }
    if ((rc = cs_track_history_set_span_dev(h, hip_stream, firstKeyFrame, nNodes, o->newR, o->newT))) return rc;
    hipLaunchKernelGGL(k_ba_output_tail, dim3((o->nCams * 12 + 255) / 256), dim3(256), 0, s, o->nCams, nNodes, o->newR, o->newT, d_Rcur, d_tcur);
    CS_CHECK_LAUNCH();
    if (!(mask & CS_BA_APPLY_UPDATE)) return CS_OK;
    if (o->featRef)   // MapPoint::pFeatures as references: stale features are views, the walks follow re-linked chains, the frame test from them
        return cs_update_new_poses_points_ref_dev(h, hip_stream, cams, (const cs_feat_ref*)o->featRef, o->refStatic, nMap, nullptr, nullptr,
                                                  firstKeyFrame, d_mapPts, d_mapCov, d_mapFlags, pixelErrVar, d_counts);
    return cs_update_new_poses_points_dev(h, hip_stream, cams, d_pointFeat, nMap, nullptr, nullptr, firstKeyFrame, d_mapPts, d_mapCov,
                                          d_mapFlags, pixelErrVar, d_counts);
}


// ---- InterCamPoseEstimator::addMapPoints + apply's solve, the problem built on the device (ba_intercam_dev.h) ------------------------
cs_ba_intercam* cs_ba_intercam_create(int device, int nCams, int N, int ptsStride, int nMapPts, int maxDyn) {
    if (nCams < 1 || nCams > 16 || N < 1 || N >= (1 << 24) || ptsStride < 1 || nMapPts < 1 || maxDyn < 0 || maxDyn > 62) {
        cs_set_error("cs_ba_intercam_create: need 1..16 cameras, N in 1..2^24, ptsStride >= 1, nMapPts >= 1, maxDyn in 0..62");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        cs_set_error("cs_ba_intercam_create: no usable HIP device %d", device);
        return nullptr;
    }
    cs_ba_intercam* ic = new cs_ba_intercam();
    ic->device = device, ic->nCams = nCams, ic->N = N, ic->ptsStride = ptsStride, ic->nMap = nMapPts, ic->maxDyn = maxDyn;
    ic->maxP = nCams * ptsStride + maxDyn + 1;
    ic->maxObs = nCams * ptsStride + (maxDyn + 1) * nCams;
    ic->issued = ic->consumed = 0;
    ic->lastC = ic->lastP = ic->lastObs = ic->lastStatic = 0;
    ic->slab = nullptr, ic->h_totals = ic->h_plan = nullptr;
    const size_t P1 = ic->maxP, O1 = ic->maxObs, nPairs = (size_t)nCams * (nCams + 1) / 2;
    struct Piece {
        void** ptr;
        size_t bytes;
    };
    std::vector<Piece> pieces;
    for (int k = 0; k < IC_STAGES; ++k) {
        IcStage& t = ic->st[k];
        pieces.push_back({(void**)&t.Ks, 72 * (size_t)nCams}), pieces.push_back({(void**)&t.Rs, 72 * (size_t)nCams});
        pieces.push_back({(void**)&t.Ts, 24 * (size_t)nCams}), pieces.push_back({(void**)&t.pts, 24 * P1});
        pieces.push_back({(void**)&t.obs_xy, 16 * O1}), pieces.push_back({(void**)&t.obs_ptr, 4 * (P1 + 1)});
        pieces.push_back({(void**)&t.obs_cam, 4 * O1}), pieces.push_back({(void**)&t.pointMap, 4 * P1});
        pieces.push_back({(void**)&t.obs_pt, 4 * O1}), pieces.push_back({(void**)&t.obs_of, 4 * P1 * nCams});
        pieces.push_back({(void**)&t.totals, 32}), pieces.push_back({(void**)&t.stPts, 24 * (size_t)nCams * ptsStride});
        pieces.push_back({(void**)&t.stXY, 16 * (size_t)nCams * ptsStride}), pieces.push_back({(void**)&t.stMap, 4 * (size_t)nCams * ptsStride});
        pieces.push_back({(void**)&t.stCount, 4 * (size_t)nCams}), pieces.push_back({(void**)&t.dynMark, (size_t)nMapPts});
    }
    pieces.push_back({(void**)&ic->pairCnt, 4 * (nPairs + 1)}), pieces.push_back({(void**)&ic->pairTotal, 32});
    pieces.push_back({(void**)&ic->lastPointMap, 4 * P1});
    size_t total = 0;
    for (const Piece& q : pieces) total += (q.bytes + 255) & ~(size_t)255;
    if (hipMalloc((void**)&ic->slab, total) != hipSuccess ||
        hipHostMalloc((void**)&ic->h_totals, (8 + P1 + 1) * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&ic->h_plan, (P1 + 2) * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        cs_set_error("cs_ba_intercam_create: cannot allocate %zu KB", total >> 10);
        if (ic->slab) (void)hipFree(ic->slab);
        if (ic->h_totals) (void)hipHostFree(ic->h_totals);
        delete ic;
        return nullptr;
    }
    (void)hipMemset(ic->slab, 0, total);
    size_t off = 0;
    for (const Piece& q : pieces) {
        *q.ptr = ic->slab + off;
        off += (q.bytes + 255) & ~(size_t)255;
    }
    return ic;
}

void cs_ba_intercam_destroy(cs_ba_intercam* ic) {   // after cs_ba_wait() of every workspace that still has a request of it queued
    if (!ic) return;
    (void)hipSetDevice(ic->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(ic->slab);
    (void)hipHostFree(ic->h_totals);
    (void)hipHostFree(ic->h_plan);
    delete ic;
}

// The problem is BUILT when the request is made, on after_stream (two launches: the frame's records as they stand in that stream's
// order), into one of three staging records; workspace b's worker thread copies it and solves: bundleAdjustRobust(0, Ks, Rs, Ts,
// m_numStatic, pts, meas, maxErr, maxIter, innerMaxIter) (SL_InterCamPoseEstimator.cpp:95).  A fourth request while three records
// are still waiting for their workers blocks the caller.
int cs_ba_solve_intercam_async(cs_ba* b, cs_ba_intercam* ic, void* after_stream, const cs_intercam_cam* cams, int W, int H, int nColBlk,
                               int nRowBlk, const double* d_R, const double* d_t, const double* d_mapPts, const unsigned char* d_mapFlags,
                               const unsigned char* d_newPt, const int* d_pointFeat, double maxErr, int maxIter, int innerMaxIter) {
    if (!b || !ic || !cams || b->device != ic->device || W < 1 || H < 1 || nColBlk < 1 || nRowBlk < 1 || nColBlk * nRowBlk > IC_MAX_BLOCKS ||
        W / nColBlk < 1 || H / nRowBlk < 1 || !d_R || !d_t || !d_mapPts || !d_mapFlags || !d_newPt || !d_pointFeat || maxIter < 0 ||
        innerMaxIter < 0) {
        cs_set_error("cs_ba_solve_intercam_async: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(b->device));
    if (!b->worker) {
        BaWorker* wk = new BaWorker();
        if (hipHostMalloc((void**)&wk->h_state, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
            delete wk;
            cs_set_error("cs_ba_solve_intercam_async: cannot allocate the pinned state word");
            return CS_ERR_ALLOC;
        }
        wk->h_state[0] = wk->h_state[1] = 0;
        b->worker = wk;
        wk->th = std::thread(ba_worker_main, b, wk);
    }
    IcArgs A;
    memset(&A, 0, sizeof(A));
    A.nCams = ic->nCams, A.N = ic->N, A.W = W, A.H = H, A.nColBlk = nColBlk, A.nRowBlk = nRowBlk;
    A.blkW = W / nColBlk, A.blkH = H / nRowBlk;   // src/app/SL_SingleSLAM.cpp:270-271 (integer division)
    A.ptsStride = ic->ptsStride, A.nMap = ic->nMap, A.maxDyn = ic->maxDyn;
    A.R = d_R, A.t = d_t, A.mapPts = d_mapPts, A.mapFlags = d_mapFlags, A.newPt = d_newPt, A.pointFeat = d_pointFeat;
    for (int c = 0; c < ic->nCams; ++c) {
        const cs_intercam_cam& q = cams[c];
        if (!q.K || !q.xy || !q.state || !q.slot2map || !q.trackSpan || !q.isStatic) {
            cs_set_error("cs_ba_solve_intercam_async: null pointer in camera %d", c);
            return CS_ERR_INVALID;
        }
        A.cam.K[c] = q.K, A.cam.xy[c] = q.xy, A.cam.state[c] = q.state, A.cam.slot2map[c] = q.slot2map, A.cam.trackSpan[c] = q.trackSpan, A.cam.isStatic[c] = q.isStatic;
    }
    int slot;
    {
        std::unique_lock<std::mutex> lk(ic->mu);
        ic->cv.wait(lk, [&] { return ic->issued - ic->consumed < IC_STAGES; });
        slot = (int)(ic->issued % IC_STAGES);
        ic->issued += 1;
    }
    A.st = ic->st[slot];
    hipStream_t as = (hipStream_t)after_stream;
    // (dynMark: zero from the start, and every build clears the marks it read -- k_ic_assemble)
    hipLaunchKernelGGL(k_ic_gather, dim3(ic->nCams), dim3(1024), 0, as, A);
    hipLaunchKernelGGL(k_ic_assemble, dim3(1), dim3(1024), 0, as, A);
    CS_CHECK_LAUNCH();
    BaAsyncJob J;
    J.C = J.P = J.nObs = 0;
    J.nCamsCon = 0, J.nPtsCon = 0, J.maxIter = maxIter, J.innerMaxIter = innerMaxIter, J.maxErr = maxErr;
    J.R0 = J.T0 = J.M0 = nullptr;
    J.ic = ic, J.icSlot = slot;
    CS_HIP(hipEventCreateWithFlags(&J.ready, hipEventDisableTiming));
    CS_HIP(hipEventRecord(J.ready, as));
    {
        std::lock_guard<std::mutex> lk(b->worker->mu);
        b->worker->q.push_back(J);
        b->worker->inflight += 1;
    }
    b->worker->cv.notify_one();
    return CS_OK;
}

// sizes of the last problem a worker solved from ic (call after cs_ba_wait): cameras, points (static first), measurements, m_numStatic,
// and the map index of every point (device, P ints)
int cs_ba_intercam_last_problem(cs_ba_intercam* ic, int* C, int* P, int* nObs, int* nStatic, const int** d_pointMap) {
    if (!ic) return CS_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ic->mu);
    if (C) *C = ic->lastC;
    if (P) *P = ic->lastP;
    if (nObs) *nObs = ic->lastObs;
    if (nStatic) *nStatic = ic->lastStatic;
    if (d_pointMap) *d_pointMap = ic->lastPointMap;
    return CS_OK;
}

// InterCamPoseEstimator::apply's write-back (src/app/SL_InterCamPoseEstimator.cpp:100-136) behind a finished solve of ic on workspace b:
// the solved poses become the cameras' current poses (m_camPos.add(curFrame, ...), :100-103) -- d_Rcur / d_tcur, and the newest frame of
// the history h when one is given --, then per camera the gate over its static mapped track nodes with the new pose: Mahalanobis error
// below 2 -> reprojErr, seqTriangulate; else the pixel distance as reprojErr and the point uncertain (:105-136).  That loop is
// SingleSLAM::poseUpdate3D's own (src/app/SL_SingleSLAM.cpp:677-706) statement for statement: the same kernel (cs_pose_update3d_dev).
// The solve must have finished (cs_ba_wait(b)); the problem's cameras are the nCams cameras in order.
int cs_ba_intercam_apply_dev(cs_ba* b, cs_ba_intercam* ic, void* hip_stream, cs_track_history* h, const cs_poseupdate_cam* cams, int N,
                             const int* d_pointFeat, int nMap, double* d_Rcur, double* d_tcur, double* d_mapPts, double* d_mapCov,
                             unsigned char* d_mapFlags, double pixelErrVar, int* d_numNodes, int* d_numOut) {
    if (!b || !ic || !cams || !d_Rcur || !d_tcur || !d_mapPts || !d_mapCov || !d_mapFlags || (h && cs_track_history_cams(h) != ic->nCams)) {
        cs_set_error("cs_ba_intercam_apply_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    int C_ = 0;
    {
        std::lock_guard<std::mutex> lk(ic->mu);
        C_ = ic->lastC;
    }
    if (C_ != ic->nCams) {
        cs_set_error("cs_ba_intercam_apply_dev: no finished solve of this estimator (its last problem has %d cameras, the rig %d)", C_, ic->nCams);
        return CS_ERR_INVALID;
    }
    double *dRs = nullptr, *dTs = nullptr, *dPts = nullptr;
    int rc = cs_ba_result_buffers(b, &dRs, &dTs, &dPts);
    if (rc) return rc;
    CS_HIP(hipSetDevice(ic->device));
    hipStream_t s = (hipStream_t)hip_stream;
    CS_HIP(hipMemcpyAsync(d_Rcur, dRs, sizeof(double) * 9 * ic->nCams, hipMemcpyDeviceToDevice, s));
    CS_HIP(hipMemcpyAsync(d_tcur, dTs, sizeof(double) * 3 * ic->nCams, hipMemcpyDeviceToDevice, s));
    if (h && (rc = cs_track_history_set_span_dev(h, hip_stream, cs_track_history_newest_frame(h), 1, d_Rcur, d_tcur))) return rc;
    return cs_pose_update3d_dev(ic->device, hip_stream, ic->nCams, 0, ic->nCams, cams, N, d_pointFeat, nMap, d_Rcur, d_tcur, d_mapPts, d_mapCov,
                                d_mapFlags, /*largeErr*/ 0, pixelErrVar, d_numNodes, d_numOut);
}

}  // extern "C"
