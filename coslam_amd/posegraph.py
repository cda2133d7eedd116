"""Pose-graph relaxation of the non-key frames after a bundle adjustment (cs_posegraph_*): GlobalPoseGraph::
computeNewCameraRotations + computeNewCameraTranslations (reference src/slam/SL_GlobalPoseEstimation.cpp:52-359) for all
camera graphs in one launch, as RobustBundleRTS::updateNonKeyCameraPoses (src/app/SL_CoSLAMRobustBA.cpp:230-247) needs
them after every BA."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class PoseGraphs:
    """Topology of nGraphs pose graphs (cs_posegraph).  graphs: list of (fixed uint8[n], id1 int[e], id2 int[e]) with ids
    local to the graph -- CamPoseNode::fixed, CamPoseEdge::id1, id2."""

    def __init__(self, graphs, device=0):
        self.device = int(device)
        self.node_ptr = np.zeros(len(graphs) + 1, dtype=np.int32)
        self.edge_ptr = np.zeros(len(graphs) + 1, dtype=np.int32)
        fx, a, b = [], [], []
        for g, (fixed, id1, id2) in enumerate(graphs):
            fixed = np.asarray(fixed, dtype=np.uint8).reshape(-1)
            id1 = np.asarray(id1, dtype=np.int32).reshape(-1)
            id2 = np.asarray(id2, dtype=np.int32).reshape(-1)
            assert len(id1) == len(id2)
            self.node_ptr[g + 1] = self.node_ptr[g] + len(fixed)
            self.edge_ptr[g + 1] = self.edge_ptr[g] + len(id1)
            fx.append(fixed), a.append(id1), b.append(id2)
        cat = lambda v, dt: np.ascontiguousarray(np.concatenate(v) if v else np.zeros(0), dtype=dt)  # noqa: E731
        self.fixed, self.id1, self.id2 = cat(fx, np.uint8), cat(a, np.int32), cat(b, np.int32)
        self.n_nodes, self.n_edges = int(self.node_ptr[-1]), int(self.edge_ptr[-1])
        self._h = C.c_void_p()
        p = lambda v: C.c_void_p(v.ctypes.data)  # noqa: E731
        check(lib().cs_posegraph_create(self.device, len(graphs), p(self.node_ptr), p(self.edge_ptr), p(self.fixed), p(self.id1),
                                        p(self.id2), C.byref(self._h)), "cs_posegraph_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().cs_posegraph_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def counts(self):
        """dict(nodes, edges, components, max_half_bandwidth)"""
        v = [C.c_int() for _ in range(4)]
        check(lib().cs_posegraph_counts(self._h, *[C.byref(x) for x in v]), "cs_posegraph_counts")
        return dict(zip(("nodes", "edges", "components", "max_half_bandwidth"), [x.value for x in v]))

    def global_ends(self):
        """(ge1, ge2): the edges' ends as indices into the flat node arrays"""
        base = np.repeat(self.node_ptr[:-1], np.diff(self.edge_ptr))
        return self.id1 + base, self.id2 + base

    def relax(self, nodeR, nodeT, edgeR, edgeT):
        """host arrays in and out (cs_posegraph_relax): returns (newR [N,9], newT [N,3])"""
        nodeR = np.ascontiguousarray(nodeR, dtype=np.float64).reshape(self.n_nodes, 9)
        nodeT = np.ascontiguousarray(nodeT, dtype=np.float64).reshape(self.n_nodes, 3)
        edgeR = np.ascontiguousarray(edgeR, dtype=np.float64).reshape(self.n_edges, 9)
        edgeT = np.ascontiguousarray(edgeT, dtype=np.float64).reshape(self.n_edges, 3)
        newR, newT = np.zeros((self.n_nodes, 9)), np.zeros((self.n_nodes, 3))
        p = lambda v: C.c_void_p(v.ctypes.data)  # noqa: E731
        check(lib().cs_posegraph_relax(self._h, p(nodeR), p(nodeT), p(edgeR), p(edgeT), p(newR), p(newT)), "cs_posegraph_relax")
        return newR, newT

    def relax_dev(self, stream_ptr, d_nodeR, d_nodeT, d_edgeR, d_edgeT, d_newR, d_newT):
        vp = C.c_void_p
        check(lib().cs_posegraph_relax_dev(self._h, vp(stream_ptr), vp(d_nodeR), vp(d_nodeT), vp(d_edgeR), vp(d_edgeT), vp(d_newR),
                                           vp(d_newT)), "cs_posegraph_relax_dev")

    def edges_dev(self, stream_ptr, d_nodeR, d_nodeT, d_edgeR, d_edgeT):
        vp = C.c_void_p
        check(lib().cs_posegraph_edges_dev(self._h, vp(stream_ptr), vp(d_nodeR), vp(d_nodeT), vp(d_edgeR), vp(d_edgeT)),
              "cs_posegraph_edges_dev")

    def status(self, stream_ptr=None):
        """synchronises the stream; raises on a failed graph"""
        check(lib().cs_posegraph_status(self._h, C.c_void_p(stream_ptr), None, None), "cs_posegraph_status")


def posegraph_set_poses_dev(stream_ptr, n, d_nodeIdx, d_R, d_t, d_nodeR, d_nodeT, device=0):
    vp = C.c_void_p
    check(lib().cs_posegraph_set_poses_dev(int(device), vp(stream_ptr), int(n), vp(d_nodeIdx), vp(d_R), vp(d_t), vp(d_nodeR),
                                           vp(d_nodeT)), "cs_posegraph_set_poses_dev")


class AfterBARec(C.Structure):
    """== cs_posegraph_after_ba_rec (include/coslam_hip.h)."""

    _fields_ = [("g", C.c_void_p), ("device", C.c_int), ("nCams", C.c_int), ("d_camNode", C.c_void_p), ("d_Rs", C.c_void_p),
                ("d_Ts", C.c_void_p), ("d_nodeR", C.c_void_p), ("d_nodeT", C.c_void_p), ("d_edgeR", C.c_void_p),
                ("d_edgeT", C.c_void_p), ("d_newR", C.c_void_p), ("d_newT", C.c_void_p)]


def after_ba_function():
    """address of the native cs_posegraph_after_ba (for BAWorkspace.set_followup)"""
    return C.cast(lib().cs_posegraph_after_ba, C.c_void_p).value


def after_ba_record(graphs, n_cams, d_camNode, d_Rs, d_Ts, d_nodeR, d_nodeT, d_edgeR, d_edgeT, d_newR, d_newT, device=0):
    """the record cs_posegraph_after_ba reads; keep it (and `graphs`) alive while it is installed"""
    return AfterBARec(graphs._h, int(device), int(n_cams), d_camNode, d_Rs, d_Ts, d_nodeR, d_nodeT, d_edgeR, d_edgeT, d_newR, d_newT)
