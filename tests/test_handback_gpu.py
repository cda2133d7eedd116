"""cs_klt_handback_dev (on-device GPUKLT::addToFeaturePoints + SingleSLAM::chooseStaticFeatPts + the Ms/ms packing of
poseUpdate3D) against the oracle's restatement of those host loops: every output array bit for bit (binary64, same
operation order), over several frames of evolving track state, with and without lens distortion."""
import numpy as np
import pytest

import coslam_amd
import oracle

pytestmark = pytest.mark.gpu

W, H, N, P = 640, 480, 2000, 900


def _frame(rng, prev_status):
    f = np.zeros(N, dtype=coslam_amd.KLT_TrackedFeature)
    # tracked slots mostly stay tracked; some die, dead ones are refilled as new
    r = rng.uniform(size=N)
    st = np.where(prev_status >= 0, np.where(r < 0.9, 0, -1), np.where(r < 0.6, 1, -1)).astype(np.int32)
    f["status"] = st
    f["pos"] = rng.uniform(0.005, 0.995, (N, 2)).astype(np.float32)
    f["gain"] = 1.0
    f["fed"] = -1
    return f


@pytest.mark.parametrize("distort,n_cams", [(False, 1), (True, 3), (False, 8)])
def test_handback_matches_oracle_over_a_sequence(hip, distort, n_cams):
    import torch

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7 + n_cams)
    K = np.array([[0.82 * W, 0.3, W / 2.0 + 1.5], [0, 0.8 * W, H / 2.0 - 2.0], [0, 0, 1.0]])
    kud = np.zeros(7)
    if distort:
        kud[:3] = [0.08, 0.02, -0.01]   # pushes border points outwards: some land at >= W | H and are dropped
    mapPts = rng.uniform(-5, 5, (P, 3))
    d_K, d_kud, d_map = (torch.from_numpy(a.copy()).to(dev) for a in (K.ravel(), kud, mapPts))
    st_o = [dict(s2m=np.full(N, -1, np.int32), tl=np.full(2 * N, -1, np.int32), xy=np.zeros(2 * N), prev=np.full(N, -1, np.int32),
                 stat=(rng.uniform(size=N) < 0.3).astype(np.uint8)) for _ in range(n_cams)]
    d = [dict(dest=torch.zeros(N * 5, dtype=torch.int32, device=dev), s2m=torch.full((N,), -1, dtype=torch.int32, device=dev),
              tl=torch.full((2 * N,), -1, dtype=torch.int32, device=dev), xy=torch.zeros(2 * N, dtype=torch.float64, device=dev),
              state=torch.zeros(N, dtype=torch.int32, device=dev), selBlk=torch.zeros(192, dtype=torch.int32, device=dev),
              Ms=torch.zeros(192 * 3, dtype=torch.float64, device=dev), ms=torch.zeros(192 * 2, dtype=torch.float64, device=dev),
              sel=torch.zeros(192, dtype=torch.int32, device=dev), npts=torch.zeros(1, dtype=torch.int32, device=dev),
              opt=torch.zeros(96, dtype=torch.uint8, device=dev), stat=torch.from_numpy(st_o[c]["stat"]).to(dev))
         for c in range(n_cams)]
    d_pf = torch.full((P, n_cams), -7, dtype=torch.int32, device=dev)   # P x nCams table, one column per camera
    stream = torch.cuda.current_stream().cuda_stream
    total_dropped = 0
    for frame in range(6):
        feats = []
        for c in range(n_cams):
            f = _frame(rng, st_o[c]["prev"])
            feats.append(f)
            d[c]["dest"].copy_(torch.from_numpy(f.view(np.int32).copy()))
            if frame == 1:   # "map initialisation": associate a third of the live slots with map points, on both sides
                live = np.nonzero(st_o[c]["tl"][:N] >= 0)[0]
                pick = rng.choice(live, size=len(live) // 3, replace=False)
                st_o[c]["s2m"][pick] = rng.integers(0, P, len(pick))
                d[c]["s2m"].copy_(torch.from_numpy(st_o[c]["s2m"]))
        cams = [dict(dest=x["dest"].data_ptr(), K=d_K.data_ptr(), kud=d_kud.data_ptr(), mapPts=d_map.data_ptr(),
                     isStatic=x["stat"].data_ptr(), slot2map=x["s2m"].data_ptr(), trackSpan=x["tl"].data_ptr(),
                     xy=x["xy"].data_ptr(), state=x["state"].data_ptr(), selBlk=x["selBlk"].data_ptr(), Ms=x["Ms"].data_ptr(),
                     ms=x["ms"].data_ptr(), sel=x["sel"].data_ptr(), npts=x["npts"].data_ptr(), opt=x["opt"].data_ptr(),
                     pointFeat=d_pf.data_ptr() + 4 * c, pointFeatStride=n_cams, nPointFeat=P)
                for c, x in enumerate(d)]
        coslam_amd.handback_dev(stream, cams, N, W, H, 16, 12, 192, frame=10 + frame)
        torch.cuda.synchronize()
        for c in range(n_cams):
            o = st_o[c]
            r = oracle.handback(feats[c], W, H, K, kud, mapPts, o["s2m"], o["tl"], o["xy"], 10 + frame, isStatic=o["stat"])
            o["prev"] = np.where(r["state"] == -2, o["prev"], feats[c]["status"]).astype(np.int32)
            g = d[c]
            assert np.array_equal(g["state"].cpu().numpy(), r["state"]), (frame, c)
            assert np.array_equal(g["tl"].cpu().numpy(), o["tl"]) and np.array_equal(g["s2m"].cpu().numpy(), o["s2m"])
            assert np.array_equal(g["xy"].cpu().numpy(), o["xy"]), (frame, c)
            assert np.array_equal(g["selBlk"].cpu().numpy(), r["selBlk"]), (frame, c)
            assert np.array_equal(d_pf[:, c].cpu().numpy(), oracle.point_features(r["state"], o["s2m"], P)), (frame, c)
            n = int(g["npts"].item())
            assert n == r["npts"]
            assert np.array_equal(g["sel"].cpu().numpy()[:n], r["sel"])
            assert np.array_equal(g["Ms"].cpu().numpy().reshape(-1, 3)[:n], r["Ms"])
            assert np.array_equal(g["ms"].cpu().numpy().reshape(-1, 2)[:n], r["ms"])
            opt = np.frombuffer(g["opt"].cpu().numpy().tobytes(), dtype=np.uint8)
            assert np.array_equal(opt, np.frombuffer(bytes(coslam_amd.IntraCamPoseOption()), dtype=np.uint8))
            total_dropped += int((r["state"] == -2).sum())
            if frame >= 2:
                assert n > 20
    if distort:
        assert total_dropped > 0   # the out >= W | H rule was exercised


def test_handback_argument_checks(hip):
    with pytest.raises(coslam_amd.CoslamHipError):
        coslam_amd.handback_dev(0, [dict(dest=1)], N, W, H)          # null pointers in the record
    with pytest.raises(coslam_amd.CoslamHipError):
        coslam_amd.handback_dev(0, [], N, W, H)                      # no cameras
