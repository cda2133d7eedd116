// ref_shim/ref_posegraph_standin.cpp -- GlobalPoseGraph (pose-graph relaxation of the non-key frames after a BA,
// src/slam/SL_GlobalPoseEstimation.cpp:52-337) is SURVEY.md 8f-4; the drop-in drivers below do not need its solve
// methods (ref_posegraph_test compiles the real source instead).  RobustBundleRTS holds GlobalPoseGraph members (src/app/SL_CoSLAMRobustBA.h:62), so creating one needs the
// constructor / destructor; the solve methods are only reached from RobustBundleRTS::output(), which the drop-in test
// does not drive -- they abort loudly if they ever are.  TEST INFRASTRUCTURE (see math/SL_Matrix.h).
#include <cstdio>
#include <cstdlib>

#include "math/SL_Matrix.h"
#include "slam/SL_GlobalPoseEstimation.h"

GlobalPoseGraph::GlobalPoseGraph() : nNodes(0), poseNodes(0), nEdges(0), poseEdges(0), nFixedNode(0), nConstraintEdge(0), nMaxNodes(0), nMaxEdges(0) {}
GlobalPoseGraph::~GlobalPoseGraph() { clear(); }
void GlobalPoseGraph::clear() {
    delete[] poseNodes;
    delete[] poseEdges;
    poseNodes = 0;
    poseEdges = 0;
    nNodes = nEdges = nMaxNodes = nMaxEdges = 0;
}
void GlobalPoseGraph::reserve(int n, int e) {
    clear();
    poseNodes = new CamPoseNode[n > 0 ? n : 1];
    poseEdges = new CamPoseEdge[e > 0 ? e : 1];
    nMaxNodes = n;
    nMaxEdges = e;
}
#ifndef POSEGRAPH_METHODS_ELSEWHERE  // ref_posegraph_methods_test takes the two solve methods from include/shim/slam/coslam_posegraph.h
static void not_built(const char* what) {
    fprintf(stderr, "GlobalPoseGraph::%s: pose-graph relaxation (SURVEY 8f-4) is not built\n", what);
    abort();
}
void GlobalPoseGraph::computeNewCameraRotations() { not_built("computeNewCameraRotations"); }
void GlobalPoseGraph::computeNewCameraTranslations() { not_built("computeNewCameraTranslations"); }
#endif
