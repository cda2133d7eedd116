"""The registration DECISION on the device (cs_register_decide_static_dev) against the sequential restatement of
CoSLAM::curStaticPointsRegInGroup (oracle.register_decide_static): random search tables with many conflicts -- several points whose
nearest feature is the same one, walks that end at a feature an earlier walk has just taken, points visited under several cameras."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(seed, P, C, N, p_conflict):
    rng = np.random.default_rng(seed)
    slot = rng.integers(0, max(int(N * p_conflict), 4), (P, C)).astype(np.int32)     # few distinct features: many claimants each
    slot[rng.random((P, C)) < 0.25] = rng.choice([-2, -3, -4], size=int((rng.random((P, C)) < 0.25).sum()) or 1)[0]
    flags = rng.integers(0, 8, (P, C)).astype(np.int32)
    flags[rng.random((P, C)) < 0.8] &= ~2                                              # mostly not dynamic
    merg = np.where(rng.random((P, C)) < 0.75, 1, 0).astype(np.uint8)
    mf = np.where(rng.random(P) < 0.1, rng.choice([1, 2, 4], P), 0).astype(np.uint8)
    pf = np.full((P, C), -1, np.int32)
    has = rng.random((P, C)) < 0.3
    pf[has] = rng.integers(0, N, int(has.sum()))
    slot[has] = -1
    s2m = [np.where(rng.random(N) < 0.3, rng.integers(0, P, N), -1).astype(np.int32) for _ in range(C)]
    return slot, flags, merg, mf, pf, s2m


@pytest.mark.parametrize("seed,P,C,N,pc", [(1, 1536, 8, 2000, 0.2), (2, 700, 3, 500, 0.05), (3, 4096, 8, 2000, 0.5), (4, 64, 1, 100, 0.3),
                                            (5, 2000, 16, 300, 0.02)])
def test_decision_equals_the_sequential_walks(hip, seed, P, C, N, pc):
    import torch

    import oracle
    from coslam_amd.register import register_decide_scratch_bytes, register_decide_static_dev

    slot, flags, merg, mf, pf, s2m = _case(seed, P, C, N, pc)
    o_pf, o_s2m = pf.copy(), [x.copy() for x in s2m]
    att_o, reg_o = oracle.register_decide_static(slot, flags, merg, mf, o_pf, o_s2m, map_base=7)
    dev = torch.device("cuda:0")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    d_slot, d_flags, d_merg, d_mf, d_pf = d(slot), d(flags), d(merg), d(mf), d(pf)
    d_s2m = [d(x) for x in s2m]
    d_att, d_reg = torch.zeros((P, C), dtype=torch.uint8, device=dev), torch.zeros(P, dtype=torch.uint8, device=dev)
    d_scr = torch.zeros(register_decide_scratch_bytes(C, N, P), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    register_decide_static_dev(torch.cuda.current_stream().cuda_stream, C, N, P, 7, d_slot.data_ptr(), d_flags.data_ptr(), d_merg.data_ptr(),
                               d_mf.data_ptr(), d_pf.data_ptr(), [x.data_ptr() for x in d_s2m], d_att.data_ptr(), d_reg.data_ptr(),
                               d_scr.data_ptr(), d_cnt.data_ptr(), n_sweeps=12)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().tolist()
    assert cnt[3] == 1 and cnt[2] == 12, cnt
    assert np.array_equal(d_att.cpu().numpy(), att_o) and np.array_equal(d_reg.cpu().numpy(), reg_o)
    assert np.array_equal(d_pf.cpu().numpy(), o_pf)
    for c in range(C):
        assert np.array_equal(d_s2m[c].cpu().numpy(), o_s2m[c]), c
    assert cnt[0] == int(att_o.sum()) and cnt[1] == int(reg_o.sum())
    if seed in (1, 3):
        assert att_o.sum() > 100
