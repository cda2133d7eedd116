"""the key-frame decision placing the key frames (LoopConfig.keyframe_drives, ratio 1.2 re-based at frame 60 as tests/test_keyframe_drives_gpu.py)
with the host's read-back every frame (keyframe_lag 0) against the lagged read from pinned memory (keyframe_lag 1, 2, 4): frames/s over the
240 frames behind the re-base, the key frames placed, the windows applied.  One JSON line per mode."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from coslam_amd.frameloop import FrameLoop, LoopConfig  # noqa: E402

frames = bench.render_video(list(range(bench.N_CAMS)), bench.N_FRAMES)
dev = torch.device("cuda", 0)
NA = bench.N_CAMS
video = {c: torch.from_numpy(frames[c]).to(dev) for c in range(NA)}
BASE, T = 60, 300
for rep in range(2):
    for lag in [int(a) for a in sys.argv[1:]] or [0, 2]:
        sc = bench.build_scene()
        cfg = LoopConfig(n_cams=NA, W=bench.W, H=bench.H, levels=bench.LEVELS, fw=bench.FW, fh=bench.FH, pts_stride=bench.PTS_STRIDE, n_col_blk=bench.N_COL_BLK,
                         n_row_blk=bench.N_ROW_BLK, key_every=bench.KEY_EVERY, p_reg=bench.P_REG, keyframe_drives=True, keyframe_ratio=1.2, keyframe_lag=lag)
        loop = FrameLoop(cfg, sc, video, None, bench.klt_config(), bench.reg_covariances(len(sc.points)), rank=0, world=1, device=0, associate=bench.associate)
        loop.first_frame()
        for i in range(1, BASE + 1):
            loop.step(i, False)
        loop.drain()
        loop.enable_keyframe_decision(BASE, BASE & 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(BASE + 1, T + 1):
            loop.step(i, False)
        loop.drain()
        dt = time.perf_counter() - t0
        st = loop.keyframe_stats()
        t = loop.d_t[T & 1].cpu().numpy()
        tt = np.stack([sc.pose(c, loop.vid(T))[1] for c in range(NA)])
        print(json.dumps(dict(keyframe_lag=lag, frames_per_s=(T - BASE) / dt, key_frames_placed=len(st["key_frames_placed_by_the_decision"]),
                              first_placed=st["key_frames_placed_by_the_decision"][:8], windows=loop.n_windows, applied=loop.applied,
                              wait_errors=loop.out.wait_errors(), max_pose_error_m=float(np.abs(t - tt).max()),
                              host_found_the_event_pending=st["host_waits_that_blocked"])), flush=True)
        del loop
