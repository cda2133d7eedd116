"""Collective 2 on CPU: the sliced-BA schedule of coslam_amd.multicam.run_sliced_ba driven
(a) by two real processes over gloo and (b) by three emulated ranks in one process, with the numpy phase engine
(tests/ba_phases_numpy.py), must reproduce the single-process oracle (oracle/ba_oracle.c) on the same problem."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from coslam_amd.multicam import point_slice, run_sliced_ba  # noqa: E402
from coslam_amd.synth import make_ba_problem  # noqa: E402


def problem():
    pr = make_ba_problem(n_cams=4, n_pts=60, n_cams_con=1, n_pts_con=2, seed=21, outlier_frac=0.06, visibility=0.85)
    ptr, cam, xy, _ = oracle.csr_by_point(len(pr["pts0"]), pr["obs_pt"], pr["obs_cam"], pr["obs_xy"])
    return pr, ptr, cam, xy


ARGS = dict(nCamsCon=1, nPtsCon=2, maxErr=6.0, maxIter=3, inner=8)


def reference(pr, ptr, cam, xy):
    return oracle.ba_robust(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy, ARGS["nCamsCon"], ARGS["nPtsCon"],
                            ARGS["maxErr"], ARGS["maxIter"], ARGS["inner"])


def engine(pr, ptr, cam, xy, rank, world):
    from tests.ba_phases_numpy import NumpySlicedBA

    lo, hi = point_slice(rank, world, len(pr["pts0"]))
    return NumpySlicedBA(pr["Ks"], pr["Rs0"], pr["ts0"], pr["pts0"], ptr, cam, xy, ARGS["nCamsCon"], ARGS["nPtsCon"],
                         ARGS["maxErr"], ARGS["inner"], lo, hi, rank == 0)


def check(e, ref):
    R_o, T_o, M_o, out_o, st_o = ref
    assert np.array_equal(e._out[: e.nObs], out_o)
    assert e.st["nOuter"] == st_o.nOuter and e.st["nIter"] == st_o.nIterTotal
    assert np.max(np.abs(e.R.reshape(-1, 9) - R_o.reshape(-1, 9))) < 1e-9
    assert np.max(np.abs(e.T - T_o)) < 1e-8
    sane = np.linalg.norm(M_o, axis=1) < 1e3
    assert np.max(np.abs(e._pts.reshape(-1, 3)[sane] - M_o[sane])) < 1e-7
    assert abs(e.final_cost - st_o.cost) <= 1e-8 * max(1.0, st_o.cost)


def test_three_emulated_ranks_match_the_oracle():
    pr, ptr, cam, xy = problem()
    engs = [engine(pr, ptr, cam, xy, r, 3) for r in range(3)]

    def reduce_fn(name):
        tot = sum(getattr(e, name).clone() for e in engs)
        for e in engs:
            getattr(e, name).copy_(tot)

    run_sliced_ba(engs, reduce_fn, ARGS["maxIter"], ARGS["inner"])
    ref = reference(pr, ptr, cam, xy)
    for e in engs:
        check(e, ref)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr, ptr, cam, xy = problem()
    e = engine(pr, ptr, cam, xy, rank, world)
    run_sliced_ba([e], lambda name: dist.all_reduce(getattr(e, name)), ARGS["maxIter"], ARGS["inner"])
    ok = True
    try:
        check(e, reference(pr, ptr, cam, xy))
    except AssertionError as ex:
        ok = False
        print("rank", rank, "failed:", ex, flush=True)
    np.save(os.path.join(out_dir, f"ba_ok{rank}.npy"), np.array([ok]))
    dist.destroy_process_group()


def test_two_processes_over_gloo_match_the_oracle(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert bool(np.load(tmp_path / f"ba_ok{r}.npy")[0]), f"rank {r}"
