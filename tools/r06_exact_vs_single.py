"""How often does the headline's single-pass registration differ from the reference's order (camera loop after camera loop with a search and a
refine per loop, CoSLAM::currentMapPointsRegister, src/app/SL_CoSLAM.cpp:834-898)?  Two frame loops on the same video from the same first frame:
A runs the headline's single pass, B the step-for-step mode (FrameLoop.sequential_registration, bit-exact against the reference's own run on
tests/golden/decide_golden.npz).  Both are deterministic, so as long as their states are byte-identical a frame is a controlled experiment: the
first frame after which the digests differ is the first frame in which the single pass did not do what the reference's order does.  From then on
B's state is copied into A (the map, the slot tables, the feature references ...: every tensor the loops own) and the count goes on.
Usage: r06_exact_vs_single.py [frames]   ->  one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from coslam_amd.frameloop import FrameLoop, LoopConfig  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
dev = torch.device("cuda", 0)
NA = bench.N_CAMS
video = {c: torch.from_numpy(frames[c]).to(dev) for c in range(NA)}


def make(seq):
    sc = bench.build_scene()
    # (no bMerge frames: the merge walk is sequential in both modes; no key-frame solves: their write-back lags by wall-clock-free but
    # thread-scheduled steps that the two loops would have to share to stay comparable)
    cfg = LoopConfig(n_cams=NA, W=bench.W, H=bench.H, levels=bench.LEVELS, fw=bench.FW, fh=bench.FH, pts_stride=bench.PTS_STRIDE, n_col_blk=bench.N_COL_BLK,
                     n_row_blk=bench.N_ROW_BLK, key_every=bench.KEY_EVERY, p_reg=bench.P_REG, merge_every=0, sequential_registration=seq)
    lp = FrameLoop(cfg, sc, video, None, bench.klt_config(), bench.reg_covariances(len(sc.points)), rank=0, world=1, device=0, associate=bench.associate)
    lp.first_frame()
    return lp


def tensors(lp):
    out = {}
    for k, v in vars(lp).items():
        if torch.is_tensor(v):
            out[k] = v
        elif isinstance(v, (list, tuple)) and v and all(torch.is_tensor(x) for x in v):
            for j, x in enumerate(v):
                out[f"{k}[{j}]"] = x
    return out


A, B = make(False), make(True)
differ_frames, attach_A, attach_B, first = [], 0, 0, None
for i in range(1, T + 1):
    key = False   # (see make(): the solves are off the comparison)
    A.step(i, key), B.step(i, key)
    torch.cuda.synchronize()
    pa, pb = A.d_pf.cpu().numpy(), B.d_pf.cpu().numpy()
    same = np.array_equal(pa, pb) and all(np.array_equal(x.cpu().numpy(), y.cpu().numpy()) for x, y in zip(A.d_slot2map, B.d_slot2map)) and \
        np.array_equal(A.d_map.cpu().numpy(), B.d_map.cpu().numpy())
    if not same:
        differ_frames.append({"frame": i, "attachments_differing": int((pa != pb).sum()), "attached_A": int(A._dec["cnt"][0].item()),
                              "attached_B": int(B._dec["cnt"][0].item()), "features_on_points": int((pb >= 0).sum())})
        if first is None:
            first = i
        ta, tb = tensors(A), tensors(B)   # B is the reference's order: A continues from it
        for k in ta:
            if k in tb and ta[k].shape == tb[k].shape and ta[k].dtype == tb[k].dtype:
                ta[k].copy_(tb[k])
        torch.cuda.synchronize()
def span(lo, hi):
    fr = [d for d in differ_frames if lo <= d["frame"] <= hi]
    return {"frames_differing": len(fr), "of": hi - lo + 1, "entries_differing_total": sum(d["attachments_differing"] for d in fr)}


print(json.dumps({"frames": T, "by_span": {"1-50": span(1, 50), "51-150": span(51, 150), "151-300": span(151, 300)}, "frames_in_which_the_single_pass_differs_from_the_reference_order": len(differ_frames), "first": first,
                  "frames_that_differ": differ_frames[:12] + differ_frames[-12:],
                  "what": "two loops from the same state, frame by frame (tools/r06_exact_vs_single.py): the headline's single-pass registration against "
                          "the step-for-step mode; no bMerge frames, no key-frame solves"}))
