"""diagnostic: the headline loop with its key frames where the decision puts them (LoopConfig.keyframe_drives) -- the key-pose state re-based
at frame BASE (as a key frame added there would leave it), then m_mappedPtsReduceRatio = argv ratios: where the key frames fall"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
import torch
from coslam_amd.frameloop import FrameLoop, LoopConfig
dev = torch.device("cuda", 0)
NA = bench.N_CAMS
video = {c: torch.from_numpy(frames[c]).to(dev) for c in range(NA)}
BASE, T = int(os.environ.get("BASE", "60")), int(os.environ.get("T", "300"))
for ratio in [float(a) for a in sys.argv[1:]] or [1.15]:
    sc = bench.build_scene()
    cfg = LoopConfig(n_cams=NA, W=bench.W, H=bench.H, levels=bench.LEVELS, fw=bench.FW, fh=bench.FH, pts_stride=bench.PTS_STRIDE, n_col_blk=bench.N_COL_BLK,
                     n_row_blk=bench.N_ROW_BLK, key_every=bench.KEY_EVERY, p_reg=bench.P_REG, keyframe_drives=True, keyframe_ratio=ratio)
    loop = FrameLoop(cfg, sc, video, None, bench.klt_config(), bench.reg_covariances(len(sc.points)), rank=0, world=1, device=0, associate=bench.associate)
    loop.first_frame()
    for i in range(1, BASE + 1):
        loop.step(i, False)
    loop.drain()
    early = loop.keyframe_stats()["key_frames_placed_by_the_decision"]
    loop.enable_keyframe_decision(BASE, BASE & 1)
    for i in range(BASE + 1, T + 1):
        loop.step(i, False)
    loop.drain()
    st = loop.keyframe_stats()
    R = loop.d_R[T & 1].cpu().numpy().reshape(NA, 3, 3)
    t = loop.d_t[T & 1].cpu().numpy()
    tt = np.stack([sc.pose(c, loop.vid(T))[1] for c in range(NA)])
    print("ratio", ratio, "placed before the re-base", early, "placed", st["key_frames_placed_by_the_decision"], "pushed", loop.n_pushed, "windows", loop.n_windows, "applied", loop.applied,
          "not applied", st["windows_not_applied_history_too_short"], "wait errors", loop.out.wait_errors() if loop.out is not None else None,
          "t err", float(np.abs(t - tt).max()), flush=True)
