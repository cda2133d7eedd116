// posegraph.hip -- pose-graph relaxation of the non-key frames after a bundle adjustment, all camera graphs in one launch,
// gfx950.
//
// Replaces GlobalPoseGraph::computeNewCameraRotations (src/slam/SL_GlobalPoseEstimation.cpp:52-219) followed by
// computeNewCameraTranslations (:220-359) as RobustBundleRTS::updateNonKeyCameraPoses runs them for every camera after a BA
// (src/app/SL_CoSLAMRobustBA.cpp:230-247), plus the two loops around them that are pure data movement: the edges'
// relative transforms (constructCameraGraphs, :216-227, getRigidTransFromTo) and the copy of the adjusted key poses into
// the fixed nodes (output(), :283-294).
//
// What the reference solves: every edge (i -> j, R_ij, T_ij) with a free end gives the linear equations
//       R_j - R_ij R_i = 0   (9 per edge)        T_j - R_ij T_i = T_ij   (3 per edge)
// in the poses of the free nodes (fixed nodes go to the right-hand side), solved in the least-squares sense by a sparse QR
// (sparseSolveLin, un-vendored LibVisualSLAM), the 3x3 blocks of the first solution then replaced by the nearest rotation
// (approxRotationMat).  Two facts shape the kernel:
//   * the rotation system is the translation system three times over: column a of R_j obeys r_j - R_ij r_i = 0, so both
//     have the SAME 3-unknowns-per-node matrix A and only differ in the right-hand side.  One factorisation, 4 right-hand
//     sides (3 rotation columns + the translation), instead of a 9n and a 3n system;
//   * fixed nodes cut the graph: the free nodes fall into connected components that share nothing (for CoSLAM's chains:
//     the runs of non-key frames between two key frames).  Each component is an independent banded problem.
// So: one wave per component, normal equations N = A^T A (3x3 blocks: +I / R^T R on the diagonal per incident edge, -R_ij
// for a free-free edge) assembled lane-per-node in edge order (deterministic, no atomics) into LDS band storage, a band
// L D L^T (no square roots; the window update and the forward substitution of the 4 right-hand sides are one set of lane
// items, ONE barrier per unknown), back substitution with the band striped over a quad and folded on the DPP path, and
// a lane-per-node polar projection (Newton iteration X <- (X + X^-T) / 2: the orthogonal polar factor IS U V^T).  Band
// width follows from the component's node order (node index order: half-bandwidth 5 for a chain), so any topology the
// reference's API can express is solved -- a wide band only costs time.  Components too large for LDS run the same code
// out of an HBM workspace.  The least-squares solution is unique (full column rank), so QR on A and L D L^T of A^T A
// agree to rounding: cond(A^T A) ~ (chain length)^2, far from 1e16.  Tolerance stated in tests/test_posegraph_gpu.py.
//
// Not built: edges with uncertainScale (extra scale unknowns; only src/app/SL_MergeCameraGroup.cpp:972-1025 creates them,
// out of SURVEY 8's scope) -- the C++ shim refuses a graph that has one.
#include "cs_common.h"

#include <algorithm>
#include <new>
#include <vector>

namespace {

constexpr int PG_LDS_DOUBLES = 7680;  // 60 KB: a component whose band + right-hand sides fit runs out of LDS

struct PgPlan {
    int nComp, nNodes, nEdges;
    const int* compPtr;         // nComp + 1, into compNode
    const int* compNode;        // global node index of every free node, component by component, ascending inside
    const int* compW;           // scalar half-bandwidth of the component's normal matrix
    const long long* compOff;   // < 0: LDS; else offset (doubles) of the component's workspace in `scratch`
    const int* adjPtr;          // per free-node slot, into adjEnt
    const int4* adjEnt;         // {edge, role (0: the node is id2, 1: it is id1), other end: position in the component or
                                //  -1 - global node when that end is fixed, 0}
    const int* nodeSlot;        // global node -> slot, -1 = fixed
    const int* ge1;             // global node index of the edges' ends
    const int* ge2;
    double* scratch;
    int* status;                // per component: 0 ok, 1 pivot <= 0 (a free node no edge constrains), 2 singular 3x3 block
};

// nearest orthogonal matrix of M (row-major): Newton iteration on the polar factor
__device__ __forceinline__ bool polar_rotation(const double M[9], double Q[9]) {
    double X[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) X[i] = M[i];
    bool ok = true;
    for (int it = 0; it < 12; ++it) {
        double C[9];  // cofactors: X^-T = C / det
        C[0] = X[4] * X[8] - X[5] * X[7];
        C[1] = X[5] * X[6] - X[3] * X[8];
        C[2] = X[3] * X[7] - X[4] * X[6];
        C[3] = X[2] * X[7] - X[1] * X[8];
        C[4] = X[0] * X[8] - X[2] * X[6];
        C[5] = X[1] * X[6] - X[0] * X[7];
        C[6] = X[1] * X[5] - X[2] * X[4];
        C[7] = X[2] * X[3] - X[0] * X[5];
        C[8] = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
        if (!(fabs(det) > 1e-30)) {
            ok = false;
            break;
        }
        const double id = 0.5 / det;
        double delta = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double y = 0.5 * X[i] + id * C[i];
            delta = fmax(delta, fabs(y - X[i]));
            X[i] = y;
        }
        if (delta < 1e-15) break;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) Q[i] = X[i];
    return ok;
}

// 1 / d for the pivots: hardware estimate + two Newton steps (the IEEE division sequence is three times as long and sits on
// the one dependency chain this kernel has)
__device__ __forceinline__ double pg_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// lower-triangle pair (a, b), 1 <= b <= a, from its running index t = (a - 1) a / 2 + (b - 1)
__device__ __forceinline__ void pair_of(int t, int& a, int& b) {
    int r = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    a = r + 1;
    b = t - r * (r + 1) / 2 + 1;
}

// one component, one wave.  `Bnd` is LDS or the HBM workspace (two inlined copies, so the LDS one gets ds_ instructions):
//   Bnd[c * ld + a] = N[c + a][c], a = 0..w   the lower band, column by column
//   G[row * 4 + rhs]                          the 4 right-hand sides, then the solution
//   invd[c]                                   1 / pivot
__device__ __forceinline__ void pg_component(const PgPlan& p, int k, int lane, double* Bnd, const double* __restrict__ nodeR,
                                             const double* __restrict__ nodeT, const double* __restrict__ edgeR,
                                             const double* __restrict__ edgeT, double* __restrict__ newR,
                                             double* __restrict__ newT) {
    const int s0 = p.compPtr[k], cnt = p.compPtr[k + 1] - s0, n = 3 * cnt, w = p.compW[k], ld = w + 1;
    double* G = Bnd + (size_t)n * ld;
    double* invd = G + (size_t)n * 4;
    for (int i = lane; i < n * (ld + 5); i += 64) Bnd[i] = 0.0;
    __syncthreads();

    // ---- normal equations, one lane per free node, its edges in edge order ----
    for (int q = lane; q < cnt; q += 64) {
        const int slot = s0 + q;
        double D[9], g[12];
#pragma unroll
        for (int i = 0; i < 9; ++i) D[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 12; ++i) g[i] = 0.0;
        for (int e = p.adjPtr[slot]; e < p.adjPtr[slot + 1]; ++e) {
            const int4 ent = p.adjEnt[e];
            double Re[9], b[12];  // b[i * 4 + rhs]: the edge's right-hand side
#pragma unroll
            for (int i = 0; i < 9; ++i) Re[i] = edgeR[9 * (size_t)ent.x + i];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                b[4 * i] = b[4 * i + 1] = b[4 * i + 2] = 0.0;
                b[4 * i + 3] = edgeT[3 * (size_t)ent.x + i];
            }
            if (ent.y == 0) {  // this node is the edge's id2:  x_j - R x_i = b
                if (ent.z < 0) {  // id1 fixed: + R X_1 (:148-150, :330)
                    const size_t f = (size_t)(-1 - ent.z);
                    double X1[12];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        X1[4 * i] = nodeR[9 * f + 3 * i], X1[4 * i + 1] = nodeR[9 * f + 3 * i + 1];
                        X1[4 * i + 2] = nodeR[9 * f + 3 * i + 2], X1[4 * i + 3] = nodeT[3 * f + i];
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            b[4 * i + r] += (Re[3 * i] * X1[r] + Re[3 * i + 1] * X1[4 + r]) + Re[3 * i + 2] * X1[8 + r];
                }
                D[0] += 1.0, D[4] += 1.0, D[8] += 1.0;
#pragma unroll
                for (int i = 0; i < 12; ++i) g[i] += b[i];
                if (ent.z >= 0 && ent.z < q) {  // block (rows q, columns ent.z) = I^T (-R) = -R
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) Bnd[(size_t)(3 * ent.z + c) * ld + (3 * (q - ent.z) + r - c)] -= Re[3 * r + c];
                }
            } else {  // this node is the edge's id1
                if (ent.z < 0) {  // id2 fixed: - X_2 (:183-193, :313-315)
                    const size_t f = (size_t)(-1 - ent.z);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        b[4 * i] -= nodeR[9 * f + 3 * i], b[4 * i + 1] -= nodeR[9 * f + 3 * i + 1];
                        b[4 * i + 2] -= nodeR[9 * f + 3 * i + 2], b[4 * i + 3] -= nodeT[3 * f + i];
                    }
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) D[3 * r + c] += (Re[r] * Re[c] + Re[3 + r] * Re[3 + c]) + Re[6 + r] * Re[6 + c];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[4 * i + r] -= (Re[i] * b[r] + Re[3 + i] * b[4 + r]) + Re[6 + i] * b[8 + r];
                if (ent.z >= 0 && ent.z < q) {  // block (rows q, columns ent.z) = (-R)^T I = -R^T
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) Bnd[(size_t)(3 * ent.z + c) * ld + (3 * (q - ent.z) + r - c)] -= Re[3 * c + r];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = c; r < 3; ++r)
                if (r - c <= w) Bnd[(size_t)(3 * q + c) * ld + (r - c)] += D[3 * r + c];
#pragma unroll
        for (int i = 0; i < 12; ++i) G[(size_t)(3 * q) * 4 + i] = g[i];
    }
    __syncthreads();

    // ---- band L D L^T (no square roots, columns stay unscaled: L = column / pivot), the forward substitution of the 4
    //      right-hand sides fused into the elimination step; one barrier per unknown ----
    const int nPairs = w * (w + 1) / 2, nItems = nPairs + 4 * w;
    bool bad = false;
    if (nItems <= 64) {  // chains (w = 5: 15 + 20 items) and other narrow bands: one item per lane, offsets fixed for the run
        int ia = 1, ib = 1, wo = 0;
        const bool isPair = lane < nPairs, isRhs = !isPair && lane < nItems;
        if (isPair) {
            pair_of(lane, ia, ib);
            wo = ib * ld + (ia - ib);  // N[j + a][j + b], relative to column j
        } else if (isRhs) {
            ia = 1 + ((lane - nPairs) >> 2);
            ib = (lane - nPairs) & 3;  // the right-hand side
            wo = 4 * ia + ib;          // G[j + a][rhs], relative to row j
        }
        double* col = Bnd;
        double* gj = G;
        for (int j = 0; j < n; ++j, col += ld, gj += 4) {
            const double d = col[0];
            if (!(d > 0.0)) bad = true;
            const double inv = pg_rcp(d > 0.0 ? d : 1.0);
            if (lane == 63) invd[j] = inv;
            if (j + ia < n) {
                if (isPair)
                    col[wo] -= (col[ia] * inv) * col[ib];
                else if (isRhs)
                    gj[wo] -= (col[ia] * inv) * gj[ib];
            }
            __syncthreads();
        }
    } else {
        for (int j = 0; j < n; ++j) {
            const int len = min(w, n - 1 - j);
            double* col = Bnd + (size_t)j * ld;
            const double d = col[0];
            if (!(d > 0.0)) bad = true;
            const double inv = pg_rcp(d > 0.0 ? d : 1.0);
            if (lane == 63) invd[j] = inv;
            for (int t = lane; t < nItems; t += 64) {
                if (t < nPairs) {
                    int a, b;
                    pair_of(t, a, b);
                    if (a <= len) col[b * ld + (a - b)] -= (col[a] * inv) * col[b];
                } else {
                    const int a = 1 + ((t - nPairs) >> 2), r = (t - nPairs) & 3;
                    if (a <= len) G[(size_t)(j + a) * 4 + r] -= (col[a] * inv) * G[(size_t)j * 4 + r];
                }
            }
            __syncthreads();
        }
    }
    // ---- back substitution x_j = (z_j - sum_a column_j[a] x_{j+a}) / pivot_j: lane = 4 * rhs + stripe of the band, the 4
    //      stripes of a quad folded on the DPP path ----
    {
        const int r = (lane >> 2) & 3, u = lane & 3;
        const double* col = Bnd + (size_t)(n - 1) * ld;
        double* gj = G + (size_t)(n - 1) * 4 + r;
        for (int j = n - 1; j >= 0; --j, col -= ld, gj -= 4) {
            const int len = min(w, n - 1 - j);
            double sum = 0.0;
            if (lane < 16)
                for (int a = 1 + u; a <= len; a += 4) sum += col[a] * gj[4 * a];
            sum += cs_dpp_d<0xB1, 0xf>(sum);  // quad_perm [1,0,3,2]
            sum += cs_dpp_d<0x4E, 0xf>(sum);  // quad_perm [2,3,0,1]
            if (lane < 16 && u == 0) gj[0] = (gj[0] - sum) * invd[j];
            __syncthreads();
        }
    }
    // ---- results: rotation = nearest orthogonal matrix of the 3 solved columns (:205-212), translation as solved (:346) ----
    bool singular = false;
    for (int q = lane; q < cnt; q += 64) {
        const size_t node = (size_t)p.compNode[s0 + q];
        double M[9], Q[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int a = 0; a < 3; ++a) M[3 * i + a] = G[(size_t)(3 * q + i) * 4 + a];
        if (!polar_rotation(M, Q)) singular = true;
#pragma unroll
        for (int i = 0; i < 9; ++i) newR[9 * node + i] = Q[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) newT[3 * node + i] = G[(size_t)(3 * q + i) * 4 + 3];
    }
    const int st = bad ? 1 : (__any(singular) ? 2 : 0);
    if (lane == 0) p.status[k] = st;
}

__global__ __launch_bounds__(64) void k_posegraph_relax(PgPlan p, const double* __restrict__ nodeR,
                                                        const double* __restrict__ nodeT, const double* __restrict__ edgeR,
                                                        const double* __restrict__ edgeT, double* __restrict__ newR,
                                                        double* __restrict__ newT) {
    extern __shared__ double pg_lds[];
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= p.nComp) {  // the tail of the grid copies the fixed nodes (:213-214, :343-344)
        const int i = (k - p.nComp) * 64 + lane;
        if (i < p.nNodes && p.nodeSlot[i] < 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) newR[9 * (size_t)i + q] = nodeR[9 * (size_t)i + q];
#pragma unroll
            for (int q = 0; q < 3; ++q) newT[3 * (size_t)i + q] = nodeT[3 * (size_t)i + q];
        }
        return;
    }
    if (p.compOff[k] < 0)
        pg_component(p, k, lane, pg_lds, nodeR, nodeT, edgeR, edgeT, newR, newT);
    else
        pg_component(p, k, lane, p.scratch + p.compOff[k], nodeR, nodeT, edgeR, edgeT, newR, newT);
}

// relative transform of every edge from the poses of its ends (getRigidTransFromTo: R = R2 R1^T, t = t2 - R t1)
__global__ void k_posegraph_edges(int nEdges, const int* __restrict__ ge1, const int* __restrict__ ge2,
                                  const double* __restrict__ nodeR, const double* __restrict__ nodeT,
                                  double* __restrict__ edgeR, double* __restrict__ edgeT) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nEdges) return;
    const size_t i = (size_t)ge1[e], j = (size_t)ge2[e];
    double R1[9], R2[9], R[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) R1[q] = nodeR[9 * i + q], R2[q] = nodeR[9 * j + q];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[3 * r + c] = (R2[3 * r] * R1[3 * c] + R2[3 * r + 1] * R1[3 * c + 1]) + R2[3 * r + 2] * R1[3 * c + 2];
#pragma unroll
    for (int q = 0; q < 9; ++q) edgeR[9 * (size_t)e + q] = R[q];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        edgeT[3 * (size_t)e + r] = nodeT[3 * j + r] - ((R[3 * r] * nodeT[3 * i] + R[3 * r + 1] * nodeT[3 * i + 1]) + R[3 * r + 2] * nodeT[3 * i + 2]);
}

__global__ void k_posegraph_set_poses(int n, const int* __restrict__ nodeIdx, const double* __restrict__ R,
                                      const double* __restrict__ t, double* __restrict__ nodeR, double* __restrict__ nodeT) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int node = nodeIdx[i];
    if (node < 0) return;
#pragma unroll
    for (int q = 0; q < 9; ++q) nodeR[9 * (size_t)node + q] = R[9 * (size_t)i + q];
#pragma unroll
    for (int q = 0; q < 3; ++q) nodeT[3 * (size_t)node + q] = t[3 * (size_t)i + q];
}

int find_root(std::vector<int>& parent, int i) {
    while (parent[i] != i) {
        parent[i] = parent[parent[i]];
        i = parent[i];
    }
    return i;
}

}  // namespace

struct cs_posegraph {
    int device = 0;
    int nGraphs = 0, nNodes = 0, nEdges = 0, nComp = 0;
    size_t ldsBytes = 0;
    char* dev = nullptr;  // one allocation: plan arrays | status | workspace
    PgPlan plan{};
    std::vector<int> compGraph;  // component -> graph (error messages)
    double* dIn = nullptr;  // staging of the host form (inside `dev`)
};

extern "C" int cs_posegraph_create(int device, int nGraphs, const int* nodePtr, const int* edgePtr, const unsigned char* fixed,
                                   const int* id1, const int* id2, cs_posegraph** out) {
    if (!out) {
        cs_set_error("cs_posegraph_create: null out pointer");
        return CS_ERR_INVALID;
    }
    *out = nullptr;
    if (nGraphs < 0 || !nodePtr || !edgePtr || nodePtr[0] != 0 || edgePtr[0] != 0) {
        cs_set_error("cs_posegraph_create: bad graph table");
        return CS_ERR_INVALID;
    }
    for (int g = 0; g < nGraphs; ++g)
        if (nodePtr[g + 1] < nodePtr[g] || edgePtr[g + 1] < edgePtr[g]) {
            cs_set_error("cs_posegraph_create: nodePtr / edgePtr must be non-decreasing (graph %d)", g);
            return CS_ERR_INVALID;
        }
    const int N = nGraphs ? nodePtr[nGraphs] : 0, E = nGraphs ? edgePtr[nGraphs] : 0;
    if ((N > 0 && !fixed) || (E > 0 && (!id1 || !id2))) {
        cs_set_error("cs_posegraph_create: null pointer");
        return CS_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        cs_set_error("cs_posegraph_create: no usable HIP device %d (there is no CPU fallback)", device);
        return CS_ERR_NO_DEVICE;
    }
    // global ends of the edges, connected components of the free nodes (fixed nodes cut the graph)
    std::vector<int> ge1(E), ge2(E), parent(N);
    for (int i = 0; i < N; ++i) parent[i] = i;
    for (int g = 0; g < nGraphs; ++g) {
        const int nb = nodePtr[g], nn = nodePtr[g + 1] - nb;
        for (int e = edgePtr[g]; e < edgePtr[g + 1]; ++e) {
            if (id1[e] < 0 || id1[e] >= nn || id2[e] < 0 || id2[e] >= nn || id1[e] == id2[e]) {
                cs_set_error("cs_posegraph_create: edge %d of graph %d joins nodes %d -> %d (graph has %d nodes)", e - edgePtr[g], g,
                             id1[e], id2[e], nn);
                return CS_ERR_INVALID;
            }
            ge1[e] = nb + id1[e];
            ge2[e] = nb + id2[e];
            if (!fixed[ge1[e]] && !fixed[ge2[e]]) {
                const int a = find_root(parent, ge1[e]), b = find_root(parent, ge2[e]);
                if (a != b) parent[std::max(a, b)] = std::min(a, b);  // root = smallest node: components come out in node order
            }
        }
    }
    std::vector<int> compOfRoot(N, -1), nodeSlot(N, -1), nodePos(N, -1), compPtr(1, 0), compNode, compGraph;
    int nComp = 0;
    {
        std::vector<int> graphOf(N);
        for (int g = 0; g < nGraphs; ++g)
            for (int i = nodePtr[g]; i < nodePtr[g + 1]; ++i) graphOf[i] = g;
        std::vector<int> compCount;
        for (int i = 0; i < N; ++i) {
            if (fixed[i]) continue;
            const int r = find_root(parent, i);
            if (compOfRoot[r] < 0) {
                compOfRoot[r] = nComp++;
                compCount.push_back(0);
                compGraph.push_back(graphOf[i]);
            }
            ++compCount[compOfRoot[r]];
        }
        for (int c = 0; c < nComp; ++c) compPtr.push_back(compPtr.back() + compCount[c]);
        compNode.resize(compPtr.back());
        std::vector<int> fill(compPtr.begin(), compPtr.end() - 1);
        for (int i = 0; i < N; ++i) {
            if (fixed[i]) continue;
            const int c = compOfRoot[find_root(parent, i)];
            nodePos[i] = fill[c] - compPtr[c];
            nodeSlot[i] = fill[c];
            compNode[fill[c]++] = i;
        }
    }
    const int nFree = (int)compNode.size();
    // adjacency of the free nodes in edge order, block bandwidth of every component
    std::vector<int> adjPtr(nFree + 1, 0), compBw(nComp, 0);
    for (int e = 0; e < E; ++e) {
        const int s1 = nodeSlot[ge1[e]], s2 = nodeSlot[ge2[e]];
        if (s1 >= 0) ++adjPtr[s1 + 1];
        if (s2 >= 0) ++adjPtr[s2 + 1];
        if (s1 >= 0 && s2 >= 0) {
            const int c = compOfRoot[find_root(parent, ge1[e])];
            compBw[c] = std::max(compBw[c], std::abs(nodePos[ge1[e]] - nodePos[ge2[e]]));
        }
    }
    for (int s = 0; s < nFree; ++s) adjPtr[s + 1] += adjPtr[s];
    std::vector<int4> adjEnt(adjPtr[nFree]);
    {
        std::vector<int> fill(adjPtr.begin(), adjPtr.end() - 1);
        for (int e = 0; e < E; ++e) {
            const int s1 = nodeSlot[ge1[e]], s2 = nodeSlot[ge2[e]];
            if (s2 >= 0) adjEnt[fill[s2]++] = make_int4(e, 0, s1 >= 0 ? nodePos[ge1[e]] : -1 - ge1[e], 0);
            if (s1 >= 0) adjEnt[fill[s1]++] = make_int4(e, 1, s2 >= 0 ? nodePos[ge2[e]] : -1 - ge2[e], 0);
        }
    }
    std::vector<int> compW(nComp);
    std::vector<long long> compOff(nComp);
    size_t ldsDoubles = 0, scratchDoubles = 0;
    for (int c = 0; c < nComp; ++c) {
        const int n = 3 * (compPtr[c + 1] - compPtr[c]);
        compW[c] = std::min(3 * compBw[c] + 2, n - 1);
        const size_t need = (size_t)n * (compW[c] + 1 + 4 + 1);  // band | right-hand sides | 1 / pivots
        if (need <= (size_t)PG_LDS_DOUBLES) {
            compOff[c] = -1;
            ldsDoubles = std::max(ldsDoubles, need);
        } else {
            compOff[c] = (long long)scratchDoubles;
            scratchDoubles += need;
        }
    }
    cs_posegraph* G = new (std::nothrow) cs_posegraph();
    if (!G) {
        cs_set_error("cs_posegraph_create: out of memory");
        return CS_ERR_ALLOC;
    }
    G->device = device;
    G->nGraphs = nGraphs, G->nNodes = N, G->nEdges = E, G->nComp = nComp;
    G->ldsBytes = ldsDoubles * sizeof(double);
    G->compGraph = compGraph;
    // one device block
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t oCompPtr = 0, oCompNode = oCompPtr + pad(4 * (size_t)(nComp + 1)), oCompW = oCompNode + pad(4 * (size_t)nFree),
                 oCompOff = oCompW + pad(4 * (size_t)nComp), oAdjPtr = oCompOff + pad(8 * (size_t)nComp),
                 oAdjEnt = oAdjPtr + pad(4 * (size_t)(nFree + 1)), oNodeSlot = oAdjEnt + pad(16 * adjEnt.size()),
                 oGe1 = oNodeSlot + pad(4 * (size_t)N), oGe2 = oGe1 + pad(4 * (size_t)E), oStatus = oGe2 + pad(4 * (size_t)E),
                 oScratch = oStatus + pad(4 * (size_t)std::max(nComp, 1)), planBytes = oScratch,
                 oStage = oScratch + pad(8 * scratchDoubles), total = oStage + pad(8 * (24 * (size_t)N + 12 * (size_t)E));
    std::vector<char> h(planBytes, 0);
    auto put = [&](size_t off, const void* src, size_t bytes) {
        if (bytes) memcpy(h.data() + off, src, bytes);
    };
    put(oCompPtr, compPtr.data(), 4 * (size_t)(nComp + 1));
    put(oCompNode, compNode.data(), 4 * (size_t)nFree);
    put(oCompW, compW.data(), 4 * (size_t)nComp);
    put(oCompOff, compOff.data(), 8 * (size_t)nComp);
    put(oAdjPtr, adjPtr.data(), 4 * (size_t)(nFree + 1));
    put(oAdjEnt, adjEnt.data(), 16 * adjEnt.size());
    put(oNodeSlot, nodeSlot.data(), 4 * (size_t)N);
    put(oGe1, ge1.data(), 4 * (size_t)E);
    put(oGe2, ge2.data(), 4 * (size_t)E);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void**)&G->dev, total);
    if (e == hipSuccess) e = hipMemcpy(G->dev, h.data(), planBytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        cs_set_error("cs_posegraph_create: %s", hipGetErrorString(e));
        if (G->dev) (void)hipFree(G->dev);
        delete G;
        return CS_ERR_HIP;
    }
    PgPlan& p = G->plan;
    p.nComp = nComp, p.nNodes = N, p.nEdges = E;
    p.compPtr = (const int*)(G->dev + oCompPtr);
    p.compNode = (const int*)(G->dev + oCompNode);
    p.compW = (const int*)(G->dev + oCompW);
    p.compOff = (const long long*)(G->dev + oCompOff);
    p.adjPtr = (const int*)(G->dev + oAdjPtr);
    p.adjEnt = (const int4*)(G->dev + oAdjEnt);
    p.nodeSlot = (const int*)(G->dev + oNodeSlot);
    p.ge1 = (const int*)(G->dev + oGe1);
    p.ge2 = (const int*)(G->dev + oGe2);
    p.status = (int*)(G->dev + oStatus);
    p.scratch = (double*)(G->dev + oScratch);
    G->dIn = (double*)(G->dev + oStage);  // staging of the host form: nodeR | nodeT | edgeR | edgeT | newR | newT
    *out = G;
    return CS_OK;
}

extern "C" void cs_posegraph_destroy(cs_posegraph* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->dev) (void)hipFree(g->dev);
    delete g;
}

extern "C" int cs_posegraph_counts(const cs_posegraph* g, int* nNodes, int* nEdges, int* nComponents, int* maxHalfBandwidth) {
    if (!g) {
        cs_set_error("cs_posegraph_counts: null handle");
        return CS_ERR_INVALID;
    }
    if (nNodes) *nNodes = g->nNodes;
    if (nEdges) *nEdges = g->nEdges;
    if (nComponents) *nComponents = g->nComp;
    if (maxHalfBandwidth) {
        std::vector<int> w(g->nComp);
        if (g->nComp) {
            CS_HIP(hipSetDevice(g->device));
            CS_HIP(hipMemcpy(w.data(), g->plan.compW, 4 * (size_t)g->nComp, hipMemcpyDeviceToHost));
        }
        *maxHalfBandwidth = g->nComp ? *std::max_element(w.begin(), w.end()) : 0;
    }
    return CS_OK;
}

extern "C" int cs_posegraph_edges_dev(cs_posegraph* g, void* hip_stream, const double* d_nodeR, const double* d_nodeT,
                                      double* d_edgeR, double* d_edgeT) {
    if (!g || (g->nEdges > 0 && (!d_nodeR || !d_nodeT || !d_edgeR || !d_edgeT))) {
        cs_set_error("cs_posegraph_edges_dev: null pointer");
        return CS_ERR_INVALID;
    }
    if (g->nEdges == 0) return CS_OK;
    CS_HIP(hipSetDevice(g->device));
    hipLaunchKernelGGL(k_posegraph_edges, dim3((g->nEdges + 127) / 128), dim3(128), 0, (hipStream_t)hip_stream, g->nEdges, g->plan.ge1,
                       g->plan.ge2, d_nodeR, d_nodeT, d_edgeR, d_edgeT);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_posegraph_set_poses_dev(int device, void* hip_stream, int n, const int* d_nodeIdx, const double* d_R,
                                          const double* d_t, double* d_nodeR, double* d_nodeT) {
    if (n < 0 || (n > 0 && (!d_nodeIdx || !d_R || !d_t || !d_nodeR || !d_nodeT))) {
        cs_set_error("cs_posegraph_set_poses_dev: bad argument");
        return CS_ERR_INVALID;
    }
    if (n == 0) return CS_OK;
    CS_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(k_posegraph_set_poses, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)hip_stream, n, d_nodeIdx, d_R, d_t,
                       d_nodeR, d_nodeT);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_posegraph_relax_dev(cs_posegraph* g, void* hip_stream, const double* d_nodeR, const double* d_nodeT,
                                      const double* d_edgeR, const double* d_edgeT, double* d_newR, double* d_newT) {
    if (!g || (g->nNodes > 0 && (!d_nodeR || !d_nodeT || !d_newR || !d_newT)) || (g->nEdges > 0 && (!d_edgeR || !d_edgeT))) {
        cs_set_error("cs_posegraph_relax_dev: null pointer");
        return CS_ERR_INVALID;
    }
    if (d_newR == d_nodeR || d_newT == d_nodeT) {
        cs_set_error("cs_posegraph_relax_dev: the new poses must not alias the node poses (free nodes read their fixed neighbours)");
        return CS_ERR_INVALID;
    }
    if (g->nNodes == 0) return CS_OK;
    CS_HIP(hipSetDevice(g->device));
    const int grid = g->nComp + (g->nNodes + 63) / 64;
    hipLaunchKernelGGL(k_posegraph_relax, dim3(grid), dim3(64), g->ldsBytes, (hipStream_t)hip_stream, g->plan, d_nodeR, d_nodeT, d_edgeR,
                       d_edgeT, d_newR, d_newT);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

extern "C" int cs_posegraph_status(cs_posegraph* g, void* hip_stream, int* nFailed, int* firstFailedGraph) {
    if (!g) {
        cs_set_error("cs_posegraph_status: null handle");
        return CS_ERR_INVALID;
    }
    if (nFailed) *nFailed = 0;
    if (firstFailedGraph) *firstFailedGraph = -1;
    CS_HIP(hipSetDevice(g->device));
    std::vector<int> st(g->nComp);
    if (g->nComp)
        CS_HIP(hipMemcpyAsync(st.data(), g->plan.status, 4 * (size_t)g->nComp, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    CS_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    int bad = 0, first = -1, code = 0;
    for (int c = 0; c < g->nComp; ++c)
        if (st[c]) {
            if (first < 0) first = g->compGraph[c], code = st[c];
            ++bad;
        }
    if (nFailed) *nFailed = bad;
    if (firstFailedGraph) *firstFailedGraph = first;
    if (bad) {
        cs_set_error("pose-graph relaxation: %d component(s) failed, first in graph %d (%s)", bad, first,
                     code == 1 ? "a free node is not constrained by any edge" : "a solved 3x3 block is singular");
        return CS_ERR_NUMERIC;
    }
    return CS_OK;
}

extern "C" int cs_posegraph_relax(cs_posegraph* g, const double* nodeR, const double* nodeT, const double* edgeR,
                                  const double* edgeT, double* newR, double* newT) {
    if (!g || (g->nNodes > 0 && (!nodeR || !nodeT || !newR || !newT)) || (g->nEdges > 0 && (!edgeR || !edgeT))) {
        cs_set_error("cs_posegraph_relax: null pointer");
        return CS_ERR_INVALID;
    }
    if (g->nNodes == 0) return CS_OK;
    CS_HIP(hipSetDevice(g->device));
    const size_t N = g->nNodes, E = g->nEdges, inD = 12 * N + 12 * E;
    double* d = g->dIn;
    hipStream_t s = nullptr;
    // (pageable copies: a few KB each; a pinned staging block would cost more to allocate than these copies take)
    CS_HIP(hipMemcpyAsync(d, nodeR, 9 * N * 8, hipMemcpyHostToDevice, s));
    CS_HIP(hipMemcpyAsync(d + 9 * N, nodeT, 3 * N * 8, hipMemcpyHostToDevice, s));
    if (E) {
        CS_HIP(hipMemcpyAsync(d + 12 * N, edgeR, 9 * E * 8, hipMemcpyHostToDevice, s));
        CS_HIP(hipMemcpyAsync(d + 12 * N + 9 * E, edgeT, 3 * E * 8, hipMemcpyHostToDevice, s));
    }
    int rc = cs_posegraph_relax_dev(g, s, d, d + 9 * N, d + 12 * N, d + 12 * N + 9 * E, d + inD, d + inD + 9 * N);
    if (rc != CS_OK) return rc;
    CS_HIP(hipMemcpyAsync(newR, d + inD, 9 * N * 8, hipMemcpyDeviceToHost, s));
    CS_HIP(hipMemcpyAsync(newT, d + inD + 9 * N, 3 * N * 8, hipMemcpyDeviceToHost, s));
    rc = cs_posegraph_status(g, s, nullptr, nullptr);  // synchronises the stream
    return rc;
}

extern "C" int cs_posegraph_after_ba(void* hip_stream, void* rec) {
    const cs_posegraph_after_ba_rec* r = (const cs_posegraph_after_ba_rec*)rec;
    if (!r || !r->g) {
        cs_set_error("cs_posegraph_after_ba: null record");
        return CS_ERR_INVALID;
    }
    int rc = cs_posegraph_set_poses_dev(r->device, hip_stream, r->nCams, r->d_camNode, r->d_Rs, r->d_Ts, r->d_nodeR, r->d_nodeT);
    if (rc == CS_OK) rc = cs_posegraph_relax_dev(r->g, hip_stream, r->d_nodeR, r->d_nodeT, r->d_edgeR, r->d_edgeT, r->d_newR, r->d_newT);
    return rc;
}
