"""diagnostic: the key-frame decision's inputs and answers along a run of the headline loop"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
import torch
from coslam_amd.frameloop import FrameLoop, LoopConfig
dev = torch.device("cuda", 0)
sc = bench.build_scene()
NA = bench.N_CAMS
video = {c: torch.from_numpy(frames[c]).to(dev) for c in range(NA)}
cfg = LoopConfig(n_cams=NA, W=bench.W, H=bench.H, levels=bench.LEVELS, fw=bench.FW, fh=bench.FH, pts_stride=bench.PTS_STRIDE, n_col_blk=bench.N_COL_BLK,
                 n_row_blk=bench.N_ROW_BLK, key_every=bench.KEY_EVERY, p_reg=bench.P_REG, keyframe_decision=True)
loop = FrameLoop(cfg, sc, video, None, bench.klt_config(), bench.reg_covariances(len(sc.points)), rank=0, world=1, device=0, associate=bench.associate)
loop.first_frame()
print("at enable: keyFrame", loop.kf["frame"].cpu().tolist(), "keyMapped", loop.kf["mapped"].cpu().tolist(), "minTrans", loop.kf["min_translation"])
for i in range(1, int(sys.argv[1]) + 1 if len(sys.argv) > 1 else 601):
    loop.step(i, (i - 1) % cfg.key_every == 0)
    if i % 50 == 0:
        loop.drain()
        k = loop.kf
        print(i, "ready", k["ready"].cpu().tolist(), "static|num", k["cnt"].cpu().tolist(), "keyFrame", k["frame"].cpu().tolist()[:3], "keyMapped", k["mapped"].cpu().tolist()[:3], flush=True)
print(loop.keyframe_stats())
