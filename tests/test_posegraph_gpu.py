"""cs_posegraph_* (pose-graph relaxation of the non-key frames after a BA; reference src/slam/SL_GlobalPoseEstimation.cpp:
52-359 as src/app/SL_CoSLAMRobustBA.cpp:230-247 calls it) against the reference's own outputs
(tests/golden/posegraph_golden.npz) and the oracle's restatement.

Tolerance (binary64, stated once): the reference factorises the over-determined system with a QR, the kernel factorises its
normal equations (band L D L^T) and reaches the polar factor by Newton iteration instead of an SVD -- the same unique
least-squares solution by different arithmetic.  With poses O(1)..O(10) and cond(A^T A) <= (component length)^2:
|dR| <= 1e-10 per entry, |dt| <= 1e-9.  Measured: ~1e-15 / ~1e-14 on CoSLAM-shaped chains, ~1e-12 on a 400-node component."""
import os

import numpy as np
import pytest
import torch

import coslam_amd
import oracle
from coslam_amd.synth import make_pose_graphs

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_R, TOL_T = 1e-10, 1e-9


def _oracle_all(pg):
    newR, newT = np.zeros_like(pg["nodeR"]), np.zeros_like(pg["nodeT"])
    for g, (fixed, id1, id2) in enumerate(pg["graphs"]):
        ns = slice(pg["node_ptr"][g], pg["node_ptr"][g + 1])
        es = slice(pg["edge_ptr"][g], pg["edge_ptr"][g + 1])
        rc, newR[ns], newT[ns] = oracle.posegraph_relax(fixed, pg["nodeR"][ns], pg["nodeT"][ns], id1, id2, pg["edgeR"][es], pg["edgeT"][es])
        assert rc == 0
    return newR, newT


def test_posegraph_matches_the_reference_golden(hip):
    """the 9 graphs of the reference run, ALL in one launch"""
    g = np.load(os.path.join(GOLD, "posegraph_golden.npz"))
    graphs = [(g["fixed"][g["node_ptr"][k]:g["node_ptr"][k + 1]], g["id1"][g["edge_ptr"][k]:g["edge_ptr"][k + 1]],
               g["id2"][g["edge_ptr"][k]:g["edge_ptr"][k + 1]]) for k in range(len(g["node_ptr"]) - 1)]
    h = coslam_amd.PoseGraphs(graphs)
    c = h.counts()
    assert c["nodes"] == len(g["fixed"]) and c["edges"] == len(g["id1"]) and c["components"] >= 20
    assert c["max_half_bandwidth"] >= 3 * 3 + 2                       # the loop edges of graph 3 widen its band
    newR, newT = h.relax(g["nodeR"], g["nodeT"], g["edgeR"], g["edgeT"])
    dR, dT = np.abs(newR - g["newR"]).max(), np.abs(newT - g["newT"]).max()
    print(f"vs the reference's own methods: |dR| {dR:.2e} |dt| {dT:.2e}")
    assert dR < TOL_R and dT < TOL_T
    fx = g["fixed"] != 0
    assert np.array_equal(newR[fx], g["nodeR"][fx]) and np.array_equal(newT[fx], g["nodeT"][fx])
    for q in newR[~fx]:
        Q = q.reshape(3, 3)
        assert np.abs(Q @ Q.T - np.eye(3)).max() < 1e-14 and np.linalg.det(Q) > 0.999
    h.close()


@pytest.mark.parametrize("kw", [dict(n_cams=8, n_frames=21, key_every=5, seed=1),                       # the bench's shape
                                dict(n_cams=8, n_frames=151, key_every=30, seed=2, free_tail=7),        # long intervals
                                dict(n_cams=3, n_frames=41, key_every=10, seed=3, loop_edges=8),        # wide bands
                                dict(n_cams=16, n_frames=9, key_every=2, seed=4),                       # 1-node components
                                dict(n_cams=1, n_frames=2, key_every=5, seed=5)])                       # one edge
def test_posegraph_matches_oracle_on_camera_graphs(hip, kw):
    pg = make_pose_graphs(**kw)
    h = coslam_amd.PoseGraphs(pg["graphs"])
    newR, newT = h.relax(pg["nodeR"], pg["nodeT"], pg["edgeR"], pg["edgeT"])
    oR, oT = _oracle_all(pg)
    dR, dT = np.abs(newR - oR).max(), np.abs(newT - oT).max()
    print(f"{kw}: {h.counts()} |dR| {dR:.2e} |dt| {dT:.2e}")
    assert dR < TOL_R and dT < TOL_T
    assert np.abs(newT - pg["nodeT"]).max() > 1e-3
    # run to run: no atomics, fixed summation order -> identical bits
    again = h.relax(pg["nodeR"], pg["nodeT"], pg["edgeR"], pg["edgeT"])
    assert np.array_equal(again[0], newR) and np.array_equal(again[1], newT)


def test_posegraph_large_component_runs_out_of_hbm_workspace(hip):
    """one chain of 1200 frames held only at its ends and at one key frame in the middle: components of ~600 free nodes
    (n = 1800 unknowns) do not fit the LDS budget and take the HBM workspace path -- same code, same answer.  The oracle's dense
    QR would need minutes here; the check is the least-squares stationarity A^T (A x - b) = 0 of both systems, and a 400-node
    component against the oracle."""
    pg = make_pose_graphs(n_cams=1, n_frames=1201, key_every=600, seed=7)
    fixed, id1, id2 = pg["graphs"][0]
    h = coslam_amd.PoseGraphs(pg["graphs"])
    assert h.counts()["components"] == 2
    newR, newT = h.relax(pg["nodeR"], pg["nodeT"], pg["edgeR"], pg["edgeT"])
    free = fixed == 0
    gT, gR = np.zeros_like(newT), np.zeros((len(fixed), 3, 3))
    for e in range(len(id1)):
        Re = pg["edgeR"][e].reshape(3, 3)
        r = newT[id2[e]] - Re @ newT[id1[e]] - pg["edgeT"][e]
        gT[id2[e]] += r
        gT[id1[e]] -= Re.T @ r
    assert np.abs(gT[free]).max() < 1e-10 and np.abs(newT[free] - pg["nodeT"][free]).max() > 1e-3
    for q in newR[free][::37]:
        Q = q.reshape(3, 3)
        assert np.abs(Q @ Q.T - np.eye(3)).max() < 1e-13
    pg = make_pose_graphs(n_cams=1, n_frames=402, key_every=401, seed=8)
    h2 = coslam_amd.PoseGraphs(pg["graphs"])
    newR, newT = h2.relax(pg["nodeR"], pg["nodeT"], pg["edgeR"], pg["edgeT"])
    oR, oT = _oracle_all(pg)
    dR, dT = np.abs(newR - oR).max(), np.abs(newT - oT).max()
    print(f"400-node component: |dR| {dR:.2e} |dt| {dT:.2e}")
    assert dR < TOL_R and dT < TOL_T


def test_posegraph_device_form_edges_and_key_pose_scatter(hip):
    """the order RobustBundleRTS::output() follows, all on the device: edges from the poses BEFORE the adjustment
    (cs_posegraph_edges_dev), adjusted key poses scattered into the fixed nodes (cs_posegraph_set_poses_dev), relaxation."""
    pg = make_pose_graphs(n_cams=8, n_frames=21, key_every=5, seed=11)
    h = coslam_amd.PoseGraphs(pg["graphs"])
    dev = torch.device("cuda:0")
    s = torch.cuda.Stream(device=dev)
    N, E = h.n_nodes, h.n_edges
    with torch.cuda.stream(s):
        d_R = torch.from_numpy(pg["nodeR0"]).to(dev)
        d_T = torch.from_numpy(pg["nodeT0"]).to(dev)
        d_eR = torch.zeros(E, 9, dtype=torch.float64, device=dev)
        d_eT = torch.zeros(E, 3, dtype=torch.float64, device=dev)
        h.edges_dev(s.cuda_stream, d_R.data_ptr(), d_T.data_ptr(), d_eR.data_ptr(), d_eT.data_ptr())
        key = np.nonzero(h.fixed)[0].astype(np.int32)
        idx = np.concatenate([key, [-1]]).astype(np.int32)                     # a skipped entry
        kR = torch.from_numpy(np.concatenate([pg["nodeR"][key], np.zeros((1, 9))])).to(dev)
        kT = torch.from_numpy(np.concatenate([pg["nodeT"][key], np.zeros((1, 3))])).to(dev)
        d_idx = torch.from_numpy(idx).to(dev)
        coslam_amd.posegraph_set_poses_dev(s.cuda_stream, len(idx), d_idx.data_ptr(), kR.data_ptr(), kT.data_ptr(), d_R.data_ptr(),
                                           d_T.data_ptr())
        d_nR, d_nT = torch.zeros_like(d_R), torch.zeros_like(d_T)
        h.relax_dev(s.cuda_stream, d_R.data_ptr(), d_T.data_ptr(), d_eR.data_ptr(), d_eT.data_ptr(), d_nR.data_ptr(), d_nT.data_ptr())
    h.status(s.cuda_stream)
    oeR, oeT = oracle.posegraph_edges(pg["nodeR0"], pg["nodeT0"], pg["ge1"], pg["ge2"])
    assert np.array_equal(d_eR.cpu().numpy(), oeR) and np.array_equal(d_eT.cpu().numpy(), oeT)   # same expression order, no FMA
    assert np.array_equal(d_R.cpu().numpy(), pg["nodeR"]) and np.array_equal(d_T.cpu().numpy(), pg["nodeT"])
    pg2 = dict(pg, edgeR=oeR, edgeT=oeT)
    oR, oT = _oracle_all(pg2)
    assert np.abs(d_nR.cpu().numpy() - oR).max() < TOL_R and np.abs(d_nT.cpu().numpy() - oT).max() < TOL_T


def test_posegraph_failure_and_argument_errors(hip):
    """a free node no edge constrains fails its graph loudly (CS_ERR_NUMERIC naming the graph) while the other graphs of the
    launch are still solved; bad topology is refused at create."""
    pg = make_pose_graphs(n_cams=2, n_frames=11, key_every=5, seed=13)
    fixed, id1, id2 = pg["graphs"][1]
    graphs = [pg["graphs"][0], (np.concatenate([fixed, [0]]).astype(np.uint8), id1, id2)]     # an extra free node, no edge
    nodeR = np.concatenate([pg["nodeR"], pg["nodeR"][-1:]])
    nodeT = np.concatenate([pg["nodeT"], pg["nodeT"][-1:]])
    h = coslam_amd.PoseGraphs(graphs)
    with pytest.raises(RuntimeError, match="graph 1"):
        h.relax(nodeR, nodeT, pg["edgeR"], pg["edgeT"])
    with pytest.raises(RuntimeError):
        coslam_amd.PoseGraphs([(np.array([1, 0], np.uint8), [0], [2])])       # edge end out of range
    with pytest.raises(RuntimeError):
        coslam_amd.PoseGraphs([(np.array([1, 0], np.uint8), [1], [1])])       # self loop
    # an empty set of graphs and a graph with every node fixed are fine
    h0 = coslam_amd.PoseGraphs([])
    assert h0.counts()["nodes"] == 0
    h1 = coslam_amd.PoseGraphs([(np.array([1, 1, 1], np.uint8), [0, 1], [1, 2])])
    R, T = h1.relax(pg["nodeR"][:3], pg["nodeT"][:3], pg["edgeR"][:2], pg["edgeT"][:2])
    assert np.array_equal(R, pg["nodeR"][:3]) and np.array_equal(T, pg["nodeT"][:3])


@pytest.mark.parametrize("mode", ["dev", "async"])
def test_posegraph_runs_as_the_follow_up_of_a_bundle_adjustment(hip, mode):
    """RobustBundleRTS::output() (reference src/app/SL_CoSLAMRobustBA.cpp:273-316) on the device: the adjusted key poses go
    from the BA workspace into the fixed nodes and the non-key frames are relaxed, enqueued by the library right behind the
    solve's last kernel (cs_ba_set_followup + cs_posegraph_after_ba) -- for the up-front schedule and for the worker thread.
    Expected: the oracle's relaxation with the fixed nodes at the BA's result."""
    from coslam_amd.synth import make_ba_problem

    n_cam, n_kf, key_every = 2, 5, 5
    pr = make_ba_problem(n_cams=n_cam * n_kf, n_pts=300, noise=0.3, outlier_frac=0.02, seed=21, n_cams_con=2)
    ptr, cam, xy = oracle.csr_by_point(len(pr["pts0"]), pr["obs_pt"], pr["obs_cam"], pr["obs_xy"])[:3]
    pg = make_pose_graphs(n_cams=n_cam, n_frames=(n_kf - 1) * key_every + 1, key_every=key_every, seed=22)
    h = coslam_amd.PoseGraphs(pg["graphs"])
    key_nodes = np.nonzero(h.fixed)[0].astype(np.int32)              # BA camera j <-> j-th key node (camera-major)
    assert len(key_nodes) == n_cam * n_kf
    before_R, before_T = pg["nodeR0"].copy(), pg["nodeT0"].copy()
    before_R[key_nodes], before_T[key_nodes] = pr["Rs0"].reshape(-1, 9), pr["ts0"]      # the key frames' poses before the BA
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    d_nodeR, d_nodeT = T(before_R), T(before_T)
    d_eR = torch.zeros(h.n_edges, 9, dtype=torch.float64, device=dev)
    d_eT = torch.zeros(h.n_edges, 3, dtype=torch.float64, device=dev)
    d_newR, d_newT = torch.zeros_like(d_nodeR), torch.zeros_like(d_nodeT)
    d_camNode = T(key_nodes)
    s = torch.cuda.current_stream().cuda_stream
    h.edges_dev(s, d_nodeR.data_ptr(), d_nodeT.data_ptr(), d_eR.data_ptr(), d_eT.data_ptr())     # constructCameraGraphs
    ws = coslam_amd.BAWorkspace(0)
    ws.upload(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy)
    d_R0, d_T0, d_M0 = T(pr["Rs0"].reshape(-1)), T(pr["ts0"].reshape(-1)), T(pr["pts0"].reshape(-1))
    bR, bT, _ = ws.result_buffers()
    rec = coslam_amd.after_ba_record(h, len(key_nodes), d_camNode.data_ptr(), bR, bT, d_nodeR.data_ptr(), d_nodeT.data_ptr(),
                                     d_eR.data_ptr(), d_eT.data_ptr(), d_newR.data_ptr(), d_newT.data_ptr())
    import ctypes
    ws.set_followup(coslam_amd.after_ba_function(), ctypes.addressof(rec))
    if mode == "dev":
        ws.solve_dev(s, d_R0.data_ptr(), d_T0.data_ptr(), d_M0.data_ptr(), 2, 0, 6.0, 2, 10)
    else:
        ws.solve_async(s, d_R0.data_ptr(), d_T0.data_ptr(), d_M0.data_ptr(), 2, 0, 6.0, 2, 10)
        ws.wait()
    torch.cuda.synchronize()
    h.status(s)
    Rs, Ts, _, _, st = ws.download()
    assert st.nIterTotal > 0 and np.abs(Ts - pr["ts0"]).max() > 1e-3          # the BA moved the key frames
    after_R, after_T = before_R.copy(), before_T.copy()
    after_R[key_nodes], after_T[key_nodes] = Rs.reshape(-1, 9), Ts
    assert np.array_equal(d_nodeR.cpu().numpy(), after_R) and np.array_equal(d_nodeT.cpu().numpy(), after_T)
    oeR, oeT = oracle.posegraph_edges(before_R, before_T, pg["ge1"], pg["ge2"])
    oR, oT = _oracle_all(dict(pg, nodeR=after_R, nodeT=after_T, edgeR=oeR, edgeT=oeT))
    newR, newT = d_newR.cpu().numpy(), d_newT.cpu().numpy()
    assert np.abs(newR - oR).max() < TOL_R and np.abs(newT - oT).max() < TOL_T
    free = h.fixed == 0
    assert np.abs(newT[free] - before_T[free]).max() > 1e-4                  # and the non-key frames followed
    ws.set_followup(0, 0)
    ws.close()
