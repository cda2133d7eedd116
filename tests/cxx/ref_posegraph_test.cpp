// ref_posegraph_test.cpp -- the reference's OWN GlobalPoseGraph::computeNewCameraRotations / computeNewCameraTranslations
// against the pose-graph relaxation kernel.
//
// oracle/Makefile compiles /root/reference/src/slam/SL_GlobalPoseEstimation.cpp IN PLACE against oracle/ref_shim/ (Triplets,
// sparseSolveLin, approxRotationMat, mat33* are un-vendored LibVisualSLAM: stand-ins, see ref_shim/ref_posegraph_impl.cpp).
// This driver builds camera graphs with the reference's classes the way RobustBundleRTS::constructCameraGraphs / output()
// do (src/app/SL_CoSLAMRobustBA.cpp:182-229, 283-294: chain of frames, key frames fixed, edges = getRigidTransFromTo of the
// poses BEFORE the adjustment, fixed nodes then moved to the adjusted poses) plus graphs the API allows but CoSLAM's BA
// never builds (loop edges, free ends, fixed-fixed and doubled edges).
//   ref_posegraph_test golden <out.bin>   CPU only: graphs + the reference's newR / newt -- tests/golden/make_golden.py turns
//                                          them into tests/golden/posegraph_golden.npz, the fixture that pins
//                                          oracle/posegraph_oracle.c (and, on the GPU box, the kernel)
//   ref_posegraph_test                     MI355X: relaxPoseGraphs (include/shim/slam/coslam_posegraph.h -> cs_posegraph_*,
//                                          all graphs in one launch) on the same graphs, compared with the reference's
//                                          methods run in this process
// Built a second time with -DPOSEGRAPH_HIP_METHODS and WITHOUT the reference's .cpp (oracle/_ref/ref_posegraph_methods_test):
// the two member functions then come from the shim (COSLAM_HIP_DEFINE_POSEGRAPH_METHODS), the expected values from a golden
// file written by the first binary:   ref_posegraph_methods_test <golden.bin>
// TEST INFRASTRUCTURE; built into oracle/_ref/ where the reference tree exists, run by tests/test_cxx_dropin_gpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "geometry/SL_RigidTransform.h"
#include "math/SL_LinAlg.h"
#include "slam/SL_GlobalPoseEstimation.h"

#ifdef POSEGRAPH_HIP_METHODS
#define COSLAM_HIP_DEFINE_POSEGRAPH_METHODS
#endif
#include "slam/coslam_posegraph.h"

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                    \
        }                                                                \
    } while (0)

static unsigned long long g_rng = 0xD1B54A32D192ED03ull;
static double urand() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return (double)(g_rng >> 11) / 9007199254740992.0;
}
static double srand1() { return 2 * urand() - 1; }

static void rodrigues(const double w[3], double R[9]) {
    const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double k[3] = {th > 0 ? w[0] / th : 1, th > 0 ? w[1] / th : 0, th > 0 ? w[2] / th : 0};
    const double c = cos(th), s = sin(th), v = 1 - c;
    R[0] = c + k[0] * k[0] * v, R[1] = k[0] * k[1] * v - k[2] * s, R[2] = k[0] * k[2] * v + k[1] * s;
    R[3] = k[1] * k[0] * v + k[2] * s, R[4] = c + k[1] * k[1] * v, R[5] = k[1] * k[2] * v - k[0] * s;
    R[6] = k[2] * k[0] * v - k[1] * s, R[7] = k[2] * k[1] * v + k[0] * s, R[8] = c + k[2] * k[2] * v;
}

struct Pose {
    double R[9], t[3];
};

// a smooth hand-held trajectory of n frames
static std::vector<Pose> trajectory(int n) {
    std::vector<Pose> tr(n);
    double w[3] = {0.3 * srand1(), 0.3 * srand1(), 0.3 * srand1()}, p[3] = {srand1(), srand1(), 4 + srand1()};
    double dw[3] = {0.02 * srand1(), 0.02 * srand1(), 0.02 * srand1()}, dp[3] = {0.05 * srand1(), 0.05 * srand1(), 0.05 * srand1()};
    for (int i = 0; i < n; ++i) {
        rodrigues(w, tr[i].R);
        memcpy(tr[i].t, p, sizeof(p));
        for (int q = 0; q < 3; ++q) {
            dw[q] += 0.004 * srand1();
            dp[q] += 0.01 * srand1();
            w[q] += dw[q];
            p[q] += dp[q];
        }
    }
    return tr;
}

// what a BA does to a key pose: a small rigid correction
static Pose adjusted(const Pose& a, double rot, double trans) {
    const double w[3] = {rot * srand1(), rot * srand1(), rot * srand1()};
    double dR[9];
    Pose b;
    rodrigues(w, dR);
    mat33AB(dR, a.R, b.R);
    for (int q = 0; q < 3; ++q) b.t[q] = a.t[q] + trans * srand1();
    return b;
}

static void add_edge(GlobalPoseGraph& g, const std::vector<Pose>& tr, int i, int j, double noise) {
    double R[9], t[3];
    getRigidTransFromTo(tr[i].R, tr[i].t, tr[j].R, tr[j].t, R, t);
    if (noise > 0) {  // a relative pose measured some other way: not exactly consistent with the chain
        const double w[3] = {noise * srand1(), noise * srand1(), noise * srand1()};
        double dR[9], R2[9];
        rodrigues(w, dR);
        mat33AB(dR, R, R2);
        memcpy(R, R2, sizeof(R));
        for (int q = 0; q < 3; ++q) t[q] += noise * srand1();
    }
    g.addEdge()->set(i, j, R, t);
}

// graph `which` of the test set; `fixedList` = the nodes held (already moved to their adjusted poses)
static void build_graph(GlobalPoseGraph& g, int which) {
    int n = 0, keyEvery = 0, extra = 0;
    std::vector<int> fixedList;
    switch (which) {
    case 0: n = 41, keyEvery = 10; break;                                  // CoSLAM's shape: key frames 0, 10, ..., 40
    case 1: n = 27; fixedList = {0, 8, 16}; break;                         // frames after the last key frame: free tail
    case 2: n = 19; fixedList = {5, 12}; break;                            // free head and free tail
    case 3: n = 20; fixedList = {0, 19}; extra = 1; break;                 // loop edges on top of the chain (band > chain's)
    case 4: n = 2; fixedList = {0}; break;                                 // one free node, one edge
    case 5: n = 12; fixedList = {0, 1, 6, 11}; extra = 2; break;           // fixed-fixed edge 0 -> 1, doubled edge
    case 6: n = 33, keyEvery = 4; break;                                   // dense key frames: many 3-node components
    case 7: n = 90; fixedList = {0, 89}; break;                            // one long component (88 free nodes)
    default: n = 6, keyEvery = 5; break;
    }
    if (keyEvery)
        for (int i = 0; i < n; i += keyEvery) fixedList.push_back(i);
    std::vector<Pose> tr = trajectory(n);
    g.reserve(n, 3 * n);
    for (int i = 0; i < n; ++i) g.newNode()->set(i, which, tr[i].R, tr[i].t);
    for (int i = 1; i < n; ++i) add_edge(g, tr, i - 1, i, 0.0);
    if (extra == 1) {
        for (int i = 0; i + 3 < n; i += 2) add_edge(g, tr, i, i + 3, 0.01);
        add_edge(g, tr, 2, 9, 0.01);
        add_edge(g, tr, 15, 4, 0.01);  // backwards
    } else if (extra == 2) {
        add_edge(g, tr, 3, 4, 0.02);   // a second 3 -> 4
        add_edge(g, tr, 8, 10, 0.02);
    }
    for (size_t k = 0; k < fixedList.size(); ++k) {
        CamPoseNode& nd = g.poseNodes[fixedList[k]];
        nd.fixed = true;
        const Pose a = adjusted(tr[fixedList[k]], 0.01, 0.05);
        memcpy(nd.R, a.R, sizeof(a.R));  // output(): only R / t of the fixed nodes are overwritten (:290-293)
        memcpy(nd.t, a.t, sizeof(a.t));
    }
}
static const int N_GRAPHS = 9;

static void write_graph(FILE* f, const GlobalPoseGraph& g, bool withNew) {
    const int hdr[2] = {g.nNodes, g.nEdges};
    fwrite(hdr, sizeof(int), 2, f);
    for (int i = 0; i < g.nNodes; ++i) {
        const int fx = g.poseNodes[i].fixed ? 1 : 0;
        fwrite(&fx, sizeof(int), 1, f);
        fwrite(g.poseNodes[i].R, sizeof(double), 9, f);
        fwrite(g.poseNodes[i].t, sizeof(double), 3, f);
        if (withNew) {
            fwrite(g.poseNodes[i].newR, sizeof(double), 9, f);
            fwrite(g.poseNodes[i].newt, sizeof(double), 3, f);
        }
    }
    for (int k = 0; k < g.nEdges; ++k) {
        const int ids[2] = {g.poseEdges[k].id1, g.poseEdges[k].id2};
        fwrite(ids, sizeof(int), 2, f);
        fwrite(g.poseEdges[k].R, sizeof(double), 9, f);
        fwrite(g.poseEdges[k].t, sizeof(double), 3, f);
    }
}

static double max_diff(const GlobalPoseGraph& a, const GlobalPoseGraph& b, double* tdiff) {
    double dr = 0, dt = 0;
    for (int i = 0; i < a.nNodes; ++i) {
        for (int q = 0; q < 9; ++q) dr = fmax(dr, fabs(a.poseNodes[i].newR[q] - b.poseNodes[i].newR[q]));
        for (int q = 0; q < 3; ++q) dt = fmax(dt, fabs(a.poseNodes[i].newt[q] - b.poseNodes[i].newt[q]));
    }
    *tdiff = dt;
    return dr;
}

#ifndef POSEGRAPH_HIP_METHODS
static int run_golden(const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) return 2;
    const int hdr[2] = {N_GRAPHS, 0};
    fwrite(hdr, sizeof(int), 2, f);
    for (int w = 0; w < N_GRAPHS; ++w) {
        GlobalPoseGraph g;
        build_graph(g, w);
        g.computeNewCameraRotations();
        g.computeNewCameraTranslations();
        write_graph(f, g, true);
    }
    fclose(f);
    printf("ref_posegraph_test: wrote %d graphs\n", N_GRAPHS);
    return 0;
}

static int run_gpu() {
    GlobalPoseGraph ref[N_GRAPHS], hip[N_GRAPHS];
    const unsigned long long seed = g_rng;
    for (int w = 0; w < N_GRAPHS; ++w) build_graph(ref[w], w);
    g_rng = seed;
    for (int w = 0; w < N_GRAPHS; ++w) build_graph(hip[w], w);
    for (int w = 0; w < N_GRAPHS; ++w) {
        ref[w].computeNewCameraRotations();
        ref[w].computeNewCameraTranslations();
    }
    relaxPoseGraphs(hip, N_GRAPHS);  // one launch for all of them
    double worstR = 0, worstT = 0;
    for (int w = 0; w < N_GRAPHS; ++w) {
        double dt, dr = max_diff(ref[w], hip[w], &dt);
        printf("  graph %d: %3d nodes %3d edges  |dR| %.2e  |dt| %.2e\n", w, ref[w].nNodes, ref[w].nEdges, dr, dt);
        worstR = fmax(worstR, dr), worstT = fmax(worstT, dt);
        // the relaxation really moved the free nodes (the key frames were adjusted by ~0.01 rad / 0.05)
        double moved = 0;
        for (int i = 0; i < ref[w].nNodes; ++i)
            for (int q = 0; q < 3; ++q) moved = fmax(moved, fabs(ref[w].poseNodes[i].newt[q] - ref[w].poseNodes[i].t[q]));
        CHECK(moved > 1e-3);
        for (int i = 0; i < hip[w].nNodes; ++i)
            if (hip[w].poseNodes[i].fixed) {
                CHECK(!memcmp(hip[w].poseNodes[i].newR, hip[w].poseNodes[i].R, sizeof(double) * 9));
                CHECK(!memcmp(hip[w].poseNodes[i].newt, hip[w].poseNodes[i].t, sizeof(double) * 3));
            }
    }
    // tolerance: two different factorisations of the same full-rank least-squares problem (QR of A there, Cholesky of
    // A^T A here) and two different routes to the polar factor; poses are O(1)..O(10)
    CHECK(worstR < 1e-10 && worstT < 1e-9);
    // an uncertainScale edge is refused, not silently mis-solved
    hip[4].poseEdges[0].uncertainScale = true;
    bool threw = false;
    try {
        relaxPoseGraphs(&hip[4], 1);
    } catch (const std::exception& e) {
        threw = true;
    }
    CHECK(threw);
    printf("ref_posegraph_test: OK (%d graphs, worst |dR| %.2e |dt| %.2e vs the reference's own methods)\n", N_GRAPHS, worstR, worstT);
    return 0;
}
#else
static int run_methods(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        return 2;
    }
    int hdr[2];
    CHECK(fread(hdr, sizeof(int), 2, f) == 2 && hdr[0] == N_GRAPHS);
    double worstR = 0, worstT = 0;
    for (int w = 0; w < N_GRAPHS; ++w) {
        GlobalPoseGraph g;
        build_graph(g, w);
        g.computeNewCameraRotations();  // the shim's member functions
        for (int i = 0; i < g.nNodes; ++i) CHECK(!memcmp(g.poseNodes[i].newt, g.poseNodes[i].t, sizeof(double) * 3));  // :211
        g.computeNewCameraTranslations();
        int ne[2];
        CHECK(fread(ne, sizeof(int), 2, f) == 2 && ne[0] == g.nNodes && ne[1] == g.nEdges);
        for (int i = 0; i < g.nNodes; ++i) {
            int fx;
            double v[24];
            CHECK(fread(&fx, sizeof(int), 1, f) == 1 && fread(v, sizeof(double), 24, f) == 24);
            CHECK(fx == (g.poseNodes[i].fixed ? 1 : 0));
            for (int q = 0; q < 9; ++q) CHECK(v[q] == g.poseNodes[i].R[q]);  // same graph as the golden run built
            for (int q = 0; q < 9; ++q) worstR = fmax(worstR, fabs(v[12 + q] - g.poseNodes[i].newR[q]));
            for (int q = 0; q < 3; ++q) worstT = fmax(worstT, fabs(v[21 + q] - g.poseNodes[i].newt[q]));
        }
        CHECK(fseek(f, (long)g.nEdges * (2 * sizeof(int) + 12 * sizeof(double)), SEEK_CUR) == 0);
    }
    fclose(f);
    CHECK(worstR < 1e-10 && worstT < 1e-9);
    printf("ref_posegraph_methods_test: OK (worst |dR| %.2e |dt| %.2e vs the golden file)\n", worstR, worstT);
    return 0;
}
#endif

int main(int argc, char** argv) {
#ifndef POSEGRAPH_HIP_METHODS
    if (argc >= 3 && !strcmp(argv[1], "golden")) return run_golden(argv[2]);
    return run_gpu();
#else
    if (argc < 2) {
        fprintf(stderr, "usage: %s <golden.bin>\n", argv[0]);
        return 2;
    }
    return run_methods(argv[1]);
#endif
}
