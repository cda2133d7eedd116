"""N > 1 path on CPU: two ranks over gloo exchange {features, pose} with the same CameraExchange code the
bench uses over RCCL.  Each rank owns one camera; after the all-gather every rank must hold both records."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_feat, out_dir):
    sys.path.insert(0, ROOT)
    from coslam_amd.klt import KLT_TrackedFeature
    from coslam_amd.multicam import CameraExchange, features_from_words

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    feats = np.zeros(n_feat, dtype=KLT_TrackedFeature)
    feats["status"] = rng.integers(-1, 2, n_feat)
    feats["pos"] = rng.random((n_feat, 2)).astype(np.float32)
    feats["gain"] = rng.random(n_feat).astype(np.float32)
    feats["fed"] = -1
    R = torch.from_numpy(rng.standard_normal(9))
    t = torch.from_numpy(rng.standard_normal(3))
    x = CameraExchange(n_feat, torch.device("cpu"))
    for frame in range(3):  # several frames through the same buffers
        feats["pos"] += np.float32(0.001 * frame)
        x.pack(torch.from_numpy(feats.view(np.int32).copy()), R + frame, t)
        x.all_gather()
    ok = True
    for cam in range(world):
        rr = np.random.default_rng(100 + cam)
        ref = np.zeros(n_feat, dtype=KLT_TrackedFeature)
        ref["status"] = rr.integers(-1, 2, n_feat)
        ref["pos"] = rr.random((n_feat, 2)).astype(np.float32)
        ref["gain"] = rr.random(n_feat).astype(np.float32)
        ref["fed"] = -1
        Rr, tr = rr.standard_normal(9), rr.standard_normal(3)
        for frame in range(3):
            ref["pos"] += np.float32(0.001 * frame)
        w, Rg, tg = x.unpack(cam)
        got = features_from_words(w)
        ok &= bool(np.array_equal(got, ref))
        ok &= bool(np.allclose(Rg.numpy(), Rr + 2) and np.allclose(tg.numpy(), tr))
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_camera_exchange(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, 300, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert bool(np.load(tmp_path / f"ok{r}.npy")[0]), f"rank {r} saw a wrong record"


def test_single_rank_exchange_is_identity():
    sys.path.insert(0, ROOT)
    from coslam_amd.multicam import CameraExchange, record_words

    x = CameraExchange(10, torch.device("cpu"))
    d = torch.arange(50, dtype=torch.int32)
    x.pack(d, torch.arange(9, dtype=torch.float64), torch.arange(3, dtype=torch.float64))
    x.all_gather()
    w, R, t = x.unpack(0)
    assert torch.equal(w.reshape(-1), d) and R[8] == 8 and t[2] == 2
    assert record_words(10) == 74


def _group_worker(rank, world, port, n_feat, cams, out_dir):
    sys.path.insert(0, ROOT)
    from coslam_amd.klt import KLT_TrackedFeature
    from coslam_amd.multicam import FEATURE_WORDS, CameraExchange

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def camera(g):
        rng = np.random.default_rng(200 + g)
        f = np.zeros(n_feat, dtype=KLT_TrackedFeature)
        f["status"] = rng.integers(-1, 2, n_feat)
        f["pos"] = rng.random((n_feat, 2)).astype(np.float32)
        f["fed"] = -1
        return f, rng.standard_normal(9), rng.standard_normal(3)

    mine = [camera(rank * cams + i) for i in range(cams)]
    x = CameraExchange(n_feat * cams, torch.device("cpu"), cams_per_rank=cams)
    x.pack_group([torch.from_numpy(f.view(np.int32).copy()) for f, _, _ in mine], torch.from_numpy(np.stack([r for _, r, _ in mine])),
                 torch.from_numpy(np.stack([t for _, _, t in mine])))
    x.all_gather()
    ok = True
    for g in range(world * cams):     # per-camera records, indexed by GLOBAL camera -- the same layout as the native path
        f, R, t = camera(g)
        wds, Rg, tg = x.unpack(g)
        ok &= bool(np.array_equal(wds.numpy().reshape(-1), f.view(np.int32)))
        ok &= bool(np.array_equal(Rg.numpy(), R)) and bool(np.array_equal(tg.numpy(), t))
    np.save(os.path.join(out_dir, f"gok{rank}.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_with_four_cameras_each(tmp_path):
    """the bench's N = 2 layout: the rank's cameras travel in one record, one all-gather per frame"""
    world, port = 2, _free_port()
    for n_feat in (120, 121):   # (an odd slot count: the feature part is padded so that R | t stay 8-byte aligned)
        mp.spawn(_group_worker, args=(world, port, n_feat, 4, str(tmp_path)), nprocs=world, join=True)
        for r in range(world):
            assert bool(np.load(tmp_path / f"gok{r}.npy")[0]), f"rank {r} saw a wrong record ({n_feat} slots)"
        port = _free_port()
