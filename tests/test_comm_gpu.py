"""RCCL behind the C-ABI (coslam_amd/csrc/comm.hip): cs_comm_*, cs_exchange_* (one pack kernel + ncclAllGather for all of
a rank's cameras) and cs_ba_dist_solve (the sliced joint BA with its ncclAllReduce calls enqueued natively).
On a one-GPU box the communicator has ONE rank: the RCCL calls, data types, streams and the pack kernel are exercised and
checked bit for bit; the two-rank test below runs where two GPUs are visible and skips itself otherwise."""
import os
import socket
import sys

import numpy as np
import pytest

import coslam_amd
import oracle
from coslam_amd.synth import Scene, make_joint_ba_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records_ok(rec, nb, cams, feats, R, t, N):
    for g in range(cams):
        r = rec[g * nb: (g + 1) * nb]
        if not np.array_equal(r[: N * 20].view(np.int32), feats[g].view(np.int32)):
            return False
        if not np.array_equal(r[N * 20: N * 20 + 72].view(np.float64), R[g]) or not np.array_equal(r[N * 20 + 72:].view(np.float64), t[g]):
            return False
    return True


def test_single_rank_communicator_exchange_and_sliced_ba(hip):
    import torch
    from coslam_amd import multicam

    dev = torch.device("cuda:0")
    comm = multicam.NativeComm(1, 0, 0)
    N, cams = 500, 3
    rng = np.random.default_rng(3)
    feats = []
    for _ in range(cams):
        f = np.zeros(N, dtype=coslam_amd.KLT_TrackedFeature)
        f["status"] = rng.integers(-1, 2, N)
        f["pos"] = rng.random((N, 2)).astype(np.float32)
        f["gain"] = rng.random(N).astype(np.float32)
        f["fed"] = -1
        feats.append(f)
    R, t = rng.standard_normal((cams, 9)), rng.standard_normal((cams, 3))
    d_dest = [torch.from_numpy(f.view(np.int32).copy()).to(dev) for f in feats]
    d_R, d_t = torch.from_numpy(R).to(dev), torch.from_numpy(t).to(dev)
    x = multicam.CameraExchange(N * cams, dev, native=comm, cams_per_rank=cams)
    s = torch.cuda.Stream(device=dev)
    for _ in range(3):
        x.pack_group(d_dest, d_R, d_t, s)
        x.all_gather(s)
    torch.cuda.synchronize()
    rec, nb = x.native_records(0)
    assert nb == N * 20 + 96
    assert _records_ok(rec.cpu().numpy(), nb, cams, feats, R, t, N)
    # every gathered camera's pose back out of the records (what a rank does with the OTHER ranks' cameras each frame)
    d_R2, d_t2 = torch.zeros((cams, 9), dtype=torch.float64, device=dev), torch.zeros((cams, 3), dtype=torch.float64, device=dev)
    x.unpack_poses(d_R2, d_t2, s, skip_own=False)
    torch.cuda.synchronize()
    assert np.array_equal(d_R2.cpu().numpy(), R) and np.array_equal(d_t2.cpu().numpy(), t)
    x.close()

    # the frame loop's other collectives through the same communicator: a plain all-gather (registration candidates, NCC records) and
    # a broadcast (a window's packed bundle-adjustment result) -- one rank: the data must come back unchanged
    import ctypes as C

    L = coslam_amd.lib()
    L.cs_comm_allgather_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.cs_comm_broadcast_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    send = torch.arange(0, 100003, dtype=torch.int32, device=dev)
    recv = torch.zeros_like(send)
    assert L.cs_comm_allgather_dev(comm.exchange_comm, C.c_void_p(s.cuda_stream), C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel() * 4) == 0
    buf = torch.arange(7, 445 * 1024 + 7, dtype=torch.int32, device=dev)
    ref = buf.clone()
    assert L.cs_comm_broadcast_dev(comm.exchange_comm, C.c_void_p(s.cuda_stream), C.c_void_p(buf.data_ptr()), buf.numel() * 4, 0) == 0
    torch.cuda.synchronize()
    assert torch.equal(recv, send) and torch.equal(buf, ref)

    # the sliced joint BA through cs_ba_dist_solve == the oracle's bundleAdjustRobust
    sc = Scene(8, 640, 480, 7000, seed=0xC051A + 2, sigma=1.0)
    pr = make_joint_ba_problem(sc, pts_per_cam=200, pool=600, seed=77)
    ptr, cam, xy, _ = oracle.csr_by_point(len(pr["pts0"]), pr["obs_pt"], pr["obs_cam"], pr["obs_xy"])
    ws = coslam_amd.BAWorkspace(0)
    ws.upload(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy)
    d0 = [torch.from_numpy(pr[k].reshape(-1).copy()).to(dev) for k in ("Rs0", "ts0", "pts0")]
    multicam.bundle_adjust_sliced(ws, s, d0[0].data_ptr(), d0[1].data_ptr(), d0[2].data_ptr(), pr["n_cams_con"],
                                  pr["n_pts_con"], 6.0, 2, 10, 0, native=comm)
    torch.cuda.synchronize()
    Rg, Tg, Mg, out_g, st_g = ws.download()
    R_o, T_o, M_o, out_o, st_o = oracle.ba_robust(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy, pr["n_cams_con"],
                                                  pr["n_pts_con"], 6.0, 2, 10)
    assert np.array_equal(out_g, out_o) and st_g.nIterTotal == st_o.nIterTotal
    sane = np.linalg.norm(M_o, axis=1) < 1e3   # (a two-view point with a gross outlier runs off along its ray on both sides)
    assert np.max(np.abs(Rg - R_o)) < 1e-6 and np.max(np.abs(Tg - T_o)) < 1e-6 and np.max(np.abs(Mg[sane] - M_o[sane])) < 1e-5
    assert abs(st_g.cost - st_o.cost) <= 1e-7 * max(1.0, st_o.cost)
    # the same through torch's DEFAULT stream (cuda_stream == 0 -> NULL at the C-ABI): phases and all-reduces must land on
    # one and the same stream (the workspace's own), not the phases there and the collectives on the legacy null stream
    d1 = [torch.from_numpy(pr[k].reshape(-1).copy()).to(dev) for k in ("Rs0", "ts0", "pts0")]
    torch.cuda.synchronize()
    multicam.bundle_adjust_sliced(ws, torch.cuda.default_stream(dev), d1[0].data_ptr(), d1[1].data_ptr(), d1[2].data_ptr(),
                                  pr["n_cams_con"], pr["n_pts_con"], 6.0, 2, 10, 0, native=comm)
    R1, T1, M1, out1, st1 = ws.download()
    assert np.array_equal(out1, out_g) and st1.nIterTotal == st_g.nIterTotal
    assert np.array_equal(R1, Rg) and np.array_equal(T1, Tg) and np.array_equal(M1, Mg)
    ws.close()
    comm.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import coslam_amd as ca
    import oracle as orc
    from coslam_amd import multicam
    from coslam_amd.synth import Scene as Sc, make_joint_ba_problem as mk

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    comm = multicam.NativeComm(world, rank, rank)
    N, cams = 300, 2
    feats, Rs, ts = [], [], []
    for g in range(world * cams):   # every rank can rebuild every camera's record
        rng = np.random.default_rng(50 + g)
        f = np.zeros(N, dtype=ca.KLT_TrackedFeature)
        f["status"] = rng.integers(-1, 2, N)
        f["pos"] = rng.random((N, 2)).astype(np.float32)
        f["gain"] = rng.random(N).astype(np.float32)
        f["fed"] = -1
        feats.append(f)
        Rs.append(rng.standard_normal(9))
        ts.append(rng.standard_normal(3))
    mine = range(rank * cams, (rank + 1) * cams)
    d_dest = [torch.from_numpy(feats[g].view(np.int32).copy()).to(dev) for g in mine]
    d_R = torch.from_numpy(np.stack([Rs[g] for g in mine])).to(dev)
    d_t = torch.from_numpy(np.stack([ts[g] for g in mine])).to(dev)
    x = multicam.CameraExchange(N * cams, dev, native=comm, cams_per_rank=cams)
    s = torch.cuda.Stream(device=dev)
    x.pack_group(d_dest, d_R, d_t, s)
    x.all_gather(s)
    torch.cuda.synchronize()
    rec, nb = x.native_records(rank)
    ok = _records_ok(rec.cpu().numpy(), nb, world * cams, feats, np.stack(Rs), np.stack(ts), N)
    sc = Sc(8, 640, 480, 7000, seed=0xC051A + 2, sigma=1.0)
    pr = mk(sc, pts_per_cam=200, pool=600, seed=77)
    ptr, cam, xy, _ = orc.csr_by_point(len(pr["pts0"]), pr["obs_pt"], pr["obs_cam"], pr["obs_xy"])
    ws = ca.BAWorkspace(rank)
    ws.upload(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy)
    d0 = [torch.from_numpy(pr[k].reshape(-1).copy()).to(dev) for k in ("Rs0", "ts0", "pts0")]
    multicam.bundle_adjust_sliced(ws, s, d0[0].data_ptr(), d0[1].data_ptr(), d0[2].data_ptr(), pr["n_cams_con"],
                                  pr["n_pts_con"], 6.0, 2, 10, rank, native=comm)
    torch.cuda.synchronize()
    Rg, Tg, Mg, out_g, st_g = ws.download()
    R_o, T_o, M_o, out_o, st_o = orc.ba_robust(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy, pr["n_cams_con"],
                                               pr["n_pts_con"], 6.0, 2, 10)
    ok = ok and bool(np.array_equal(out_g, out_o)) and st_g.nIterTotal == st_o.nIterTotal
    sane = np.linalg.norm(M_o, axis=1) < 1e3
    ok = ok and np.max(np.abs(Rg - R_o)) < 1e-6 and np.max(np.abs(Tg - T_o)) < 1e-6 and np.max(np.abs(Mg[sane] - M_o[sane])) < 1e-5
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([bool(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_rccl_exchange_and_sliced_ba(hip, tmp_path):
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    world, port = 2, _free_port()
    mp.spawn(_rccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert bool(np.load(tmp_path / f"ok{r}.npy")[0]), f"rank {r}"


def _host_transport_worker(rank, world, name, out_dir):
    """One rank of the TEST transport (cs_comm_create_host): the same entry points as the RCCL ones, both ranks on device 0."""
    import ctypes as C

    import torch

    L = coslam_amd.lib()
    L.cs_comm_create_host.restype = C.c_void_p
    L.cs_comm_create_host.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.cs_comm_destroy.argtypes = [C.c_void_p]
    L.cs_comm_world.argtypes = [C.c_void_p]
    L.cs_comm_rank.argtypes = [C.c_void_p]
    L.cs_comm_allgather_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.cs_comm_broadcast_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    dev = torch.device("cuda:0")
    c = L.cs_comm_create_host(name.encode(), world, rank, 0)
    ok = bool(c) and L.cs_comm_world(c) == world and L.cs_comm_rank(c) == rank
    s = torch.cuda.Stream(dev)
    n = 100003  # not a multiple of anything the staging might assume
    for rnd in range(3):  # the segment is reused: a round must not see the previous one's bytes
        mine = (torch.arange(n, dtype=torch.int32, device=dev) * (rank + 1) + rnd * 7919)
        recv = torch.zeros(world * n, dtype=torch.int32, device=dev)
        ok = ok and L.cs_comm_allgather_dev(c, C.c_void_p(s.cuda_stream), C.c_void_p(mine.data_ptr()), C.c_void_p(recv.data_ptr()), n * 4) == 0
        torch.cuda.synchronize()
        want = torch.cat([torch.arange(n, dtype=torch.int32, device=dev) * (r + 1) + rnd * 7919 for r in range(world)])
        ok = ok and torch.equal(recv, want)
        # in place, the way the loop gathers the NCC blocks: the rank's own part already sits at rank * bytes of the receive buffer
        inpl = torch.full((world * n,), -1, dtype=torch.int32, device=dev)
        inpl[rank * n: (rank + 1) * n] = mine
        own = inpl[rank * n:]
        ok = ok and L.cs_comm_allgather_dev(c, C.c_void_p(s.cuda_stream), C.c_void_p(own.data_ptr()), C.c_void_p(inpl.data_ptr()), n * 4) == 0
        torch.cuda.synchronize()
        ok = ok and torch.equal(inpl, want)
        for root in range(world):
            buf = torch.arange(445 * 1024, dtype=torch.int32, device=dev) + (1000 * (rank + 1) + rnd)
            ok = ok and L.cs_comm_broadcast_dev(c, C.c_void_p(s.cuda_stream), C.c_void_p(buf.data_ptr()), buf.numel() * 4, root) == 0
            torch.cuda.synchronize()
            ok = ok and torch.equal(buf, torch.arange(445 * 1024, dtype=torch.int32, device=dev) + (1000 * (root + 1) + rnd))
    if c:
        L.cs_comm_destroy(c)
    np.save(os.path.join(out_dir, f"host_ok{rank}.npy"), np.array([bool(ok)]))


@pytest.mark.timeout(300)
def test_host_test_transport_two_ranks_on_one_device(hip, tmp_path):
    """cs_comm_create_host, the transport the one-GPU tests of the N > 1 frame loop ride on (tests/test_cxx_dropin_gpu.py, bench.py's
    BENCH_FORCE_DEVICE hook): all-gather (separate and in-place receive), broadcast from either root, over three rounds of the same
    segment -- every rank must end with exactly the bytes RCCL's calls would leave."""
    import torch.multiprocessing as mp

    name = f"/coslam_test_{os.getpid()}"
    mp.spawn(_host_transport_worker, args=(2, name, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert bool(np.load(tmp_path / f"host_ok{r}.npy")[0]), f"rank {r}"
