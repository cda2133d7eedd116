// include/shim/slam/coslam_posegraph.h -- GlobalPoseGraph::computeNewCameraRotations / computeNewCameraTranslations
// (src/slam/SL_GlobalPoseEstimation.cpp:52-359) over libcoslam_hip's pose-graph relaxation, for the caller that runs them
// after every bundle adjustment (RobustBundleRTS::updateNonKeyCameraPoses, src/app/SL_CoSLAMRobustBA.cpp:230-247).
//
//   relaxPoseGraphs(graphs, n)   all n camera graphs in ONE launch: fills poseNodes[i].newR / newt of every graph, i.e. what
//                                the reference's loop `for c: camGraphs[c].computeNewCameraRotations();
//                                camGraphs[c].computeNewCameraTranslations();` leaves behind.  This is the call a maintainer
//                                puts in place of that loop.
//   COSLAM_HIP_DEFINE_POSEGRAPH_METHODS   define it before including this header (after slam/SL_GlobalPoseEstimation.h) in
//                                ONE translation unit to get the two member functions themselves, for builds that drop
//                                SL_GlobalPoseEstimation.cpp's versions; each call solves its own graph (a launch per call).
//
// The class is not redefined: the template binds to the members the reference's own code touches (nNodes, poseNodes[i].
// fixed / R / t / newR / newt, nEdges, poseEdges[k].id1 / id2 / R / t / uncertainScale; SL_GlobalPoseEstimation.h:12-106).
// Graphs with an uncertainScale edge (camera-group merge only) are refused: std::runtime_error.
#ifndef COSLAM_SHIM_POSEGRAPH_H
#define COSLAM_SHIM_POSEGRAPH_H

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "coslam_hip.h"

#ifndef COSLAM_HIP_DEVICE
#define COSLAM_HIP_DEVICE 0
#endif

// what: 1 newR (+ newt = t on every node, as computeNewCameraRotations leaves it), 2 newt only, 3 both
template <class Graph>
inline void relaxPoseGraphs(Graph* graphs, int nGraphs, int what = 3) {
    std::vector<int> nodePtr(nGraphs + 1, 0), edgePtr(nGraphs + 1, 0);
    for (int g = 0; g < nGraphs; ++g) {
        nodePtr[g + 1] = nodePtr[g] + graphs[g].nNodes;
        edgePtr[g + 1] = edgePtr[g] + graphs[g].nEdges;
    }
    const size_t N = nodePtr[nGraphs], E = edgePtr[nGraphs];
    if (N == 0) return;
    std::vector<unsigned char> fixed(N);
    std::vector<int> id1(E ? E : 1), id2(E ? E : 1);
    std::vector<double> nR(9 * N), nT(3 * N), eR(9 * (E ? E : 1)), eT(3 * (E ? E : 1)), oR(9 * N), oT(3 * N);
    for (int g = 0; g < nGraphs; ++g) {
        for (int i = 0; i < graphs[g].nNodes; ++i) {
            const size_t o = nodePtr[g] + i;
            fixed[o] = graphs[g].poseNodes[i].fixed ? 1 : 0;
            memcpy(&nR[9 * o], graphs[g].poseNodes[i].R, sizeof(double) * 9);
            memcpy(&nT[3 * o], graphs[g].poseNodes[i].t, sizeof(double) * 3);
        }
        for (int k = 0; k < graphs[g].nEdges; ++k) {
            const size_t o = edgePtr[g] + k;
            if (graphs[g].poseEdges[k].uncertainScale)
                throw std::runtime_error("relaxPoseGraphs: edges with uncertainScale are not supported by the HIP path");
            id1[o] = graphs[g].poseEdges[k].id1;
            id2[o] = graphs[g].poseEdges[k].id2;
            memcpy(&eR[9 * o], graphs[g].poseEdges[k].R, sizeof(double) * 9);
            memcpy(&eT[3 * o], graphs[g].poseEdges[k].t, sizeof(double) * 3);
        }
    }
    cs_posegraph* h = 0;
    int rc = cs_posegraph_create(COSLAM_HIP_DEVICE, nGraphs, nodePtr.data(), edgePtr.data(), fixed.data(), id1.data(), id2.data(), &h);
    if (rc == CS_OK) rc = cs_posegraph_relax(h, nR.data(), nT.data(), eR.data(), eT.data(), oR.data(), oT.data());
    const std::string err = rc == CS_OK ? std::string() : std::string(cs_last_error());
    cs_posegraph_destroy(h);
    if (rc != CS_OK) throw std::runtime_error("relaxPoseGraphs: " + err);
    for (int g = 0; g < nGraphs; ++g)
        for (int i = 0; i < graphs[g].nNodes; ++i) {
            const size_t o = nodePtr[g] + i;
            if (what & 1) memcpy(graphs[g].poseNodes[i].newR, &oR[9 * o], sizeof(double) * 9);
            if (what & 2)
                memcpy(graphs[g].poseNodes[i].newt, &oT[3 * o], sizeof(double) * 3);
            else
                memcpy(graphs[g].poseNodes[i].newt, graphs[g].poseNodes[i].t, sizeof(double) * 3);
        }
}

#ifdef COSLAM_HIP_DEFINE_POSEGRAPH_METHODS
void GlobalPoseGraph::computeNewCameraRotations() { relaxPoseGraphs(this, 1, 1); }
void GlobalPoseGraph::computeNewCameraTranslations() { relaxPoseGraphs(this, 1, 2); }
#endif

#endif
