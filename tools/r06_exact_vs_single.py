"""How often does the headline's registration differ from the reference's order (camera loop after camera loop with a search and a refine per
loop, CoSLAM::currentMapPointsRegister, src/app/SL_CoSLAM.cpp:834-898)?  Two frame loops on the same video: B runs the step-for-step mode
(FrameLoop.sequential_registration, bit-exact against the reference's own run on tests/golden/decide_golden.npz) from the first frame; A runs
the SAME mode up to frame F0 -- both are deterministic, so their states are byte-identical there -- and from F0 + 1 on the mode under test: the
single pass alone (rounds 0) or with the second visits' rounds behind it (LoopConfig.revisit_rounds).  As long as the states stay identical a
frame is a controlled experiment; the run of a start frame ends at the first frame whose state differs (what follows would compare two
different histories).  No bMerge frames (the merge walk is sequential in both modes), no key-frame solves (their write-back is thread-timed).
Usage: r06_exact_vs_single.py <rounds> <frames per run> <F0> [<F0> ...]   ->  one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from coslam_amd.frameloop import FrameLoop, LoopConfig  # noqa: E402

ROUNDS, T = int(sys.argv[1]), int(sys.argv[2])
STARTS = [int(a) for a in sys.argv[3:]] or [150]
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
dev = torch.device("cuda", 0)
NA = bench.N_CAMS
video = {c: torch.from_numpy(frames[c]).to(dev) for c in range(NA)}


def make():
    sc = bench.build_scene()
    cfg = LoopConfig(n_cams=NA, W=bench.W, H=bench.H, levels=bench.LEVELS, fw=bench.FW, fh=bench.FH, pts_stride=bench.PTS_STRIDE, n_col_blk=bench.N_COL_BLK,
                     n_row_blk=bench.N_ROW_BLK, key_every=bench.KEY_EVERY, p_reg=bench.P_REG, merge_every=0, sequential_registration=True,
                     revisit_rounds=ROUNDS, merge_tol_pix=0.0, feature_chains=False)   # (feature_chains off: the step-for-step mode refines over this frame's features only -- the same refine on both sides)
    lp = FrameLoop(cfg, sc, video, None, bench.klt_config(), bench.reg_covariances(len(sc.points)), rank=0, world=1, device=0, associate=bench.associate)
    lp.first_frame()
    return lp


def state(lp):
    return [lp.d_pf.cpu().numpy(), np.stack([x.cpu().numpy() for x in lp.d_slot2map]), lp.d_map.cpu().numpy(), lp.d_cov.cpu().numpy(),
            lp.d_mapflags.cpu().numpy()]


runs = []
for F0 in STARTS:
    A, B = make(), make()
    for i in range(1, F0 + 1):
        A.step(i, False), B.step(i, False)
    torch.cuda.synchronize()
    assert all(np.array_equal(x, y) for x, y in zip(state(A), state(B))), "the two loops are not deterministic"
    A.sequential_registration = False
    first, att, reg_frames = None, 0, 0
    for i in range(F0 + 1, F0 + T + 1):
        A.step(i, False), B.step(i, False)
        torch.cuda.synchronize()
        n_att = int(A._dec["cnt"][0].item()) if hasattr(A, "_dec") else 0
        att += n_att
        reg_frames += n_att > 0
        sa, sb = state(A), state(B)
        if not all(np.array_equal(x, y) for x, y in zip(sa, sb)):
            rows = np.nonzero((sa[2] != sb[2]).any(1))[0][:4]
            detail = [{"point": int(r), "flags": int(sa[4][r]), "features_A": sa[0][r].tolist(), "features_B": sb[0][r].tolist(), "M_A": sa[2][r].tolist(), "M_B": sb[2][r].tolist(),
                       "registered_by_the_single_pass": int(A._dec["reg"][r].item()), "registered_in_a_round": [int(x[r].item()) for x in A.d_rv_reg],
                       "visit_loop": int(A.d_rv_visit[r].item()), "attached_row": A._dec["att"][r].cpu().tolist()} for r in rows]
            first = {"frame": i, "rows": detail, "frames_identical_before_it": i - F0 - 1, "feature_table_entries_differing": int((sa[0] != sb[0]).sum()),
                     "map_rows_differing": int((sa[2] != sb[2]).any(1).sum())}
            break
    rv = A.d_rv_counts.cpu().tolist()
    runs.append({"start_frame": F0, "frames_compared": (first["frame"] - F0) if first else T, "first_difference": first,
                 "features_attached_by_the_single_pass": att, "frames_with_a_registration": reg_frames,
                 "second_visits": dict(zip(("features_attached", "registrations", "conflicts_counted", "rounds_unsettled"), rv)),
                 "second_visit_points_beyond_the_list": int(A.d_rv_listcounts[1].item())})
    del A, B
print(json.dumps({"second_visit_rounds": ROUNDS, "runs": runs,
                  "what": "two loops from the same state (tools/r06_exact_vs_single.py): the registration under test against the step-for-step mode, frame by "
                          "frame until the states part; no bMerge frames, no key-frame solves"}))
