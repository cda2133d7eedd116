"""diagnostic: host time and device time of cs_klt_group_stage_h alone (8 x 640x480 images per call)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, coslam_amd
W, H, L, fw, fh, nc = 640, 480, 4, 50, 40, 8
cfg = coslam_amd.KLT_SequenceTrackerConfig(nIterations=10, nLevels=L, levelSkip=1, windowWidth=7, trackWithGain=1, minCornerness=3000.0, convergenceThreshold=1.0, SSD_Threshold=20000.0, minDistance=4)
ts = []
for _ in range(nc):
    t = coslam_amd.KLT_SequenceTracker(cfg, 0); t.allocate(W, H, L, fw, fh); ts.append(t)
grp = coslam_amd.KLT_TrackerGroup(ts)
s = torch.cuda.Stream(); grp.set_stream(s.cuda_stream)
lib = coslam_amd.lib(); lib.cs_pinned_alloc.restype = C.c_void_p; lib.cs_pinned_alloc.argtypes = [C.c_size_t]
for kind in ("torch_pinned", "cs_pinned", "pageable"):
    n = nc * W * H
    if kind == "torch_pinned":
        buf = torch.zeros(n, dtype=torch.uint8).pin_memory(); base = buf.data_ptr()
    elif kind == "cs_pinned":
        base = lib.cs_pinned_alloc(n)
    else:
        buf = np.zeros(n, np.uint8); base = buf.ctypes.data
    ptrs = [base + c * W * H for c in range(nc)]
    for contiguous in (True, False):
        p = ptrs if contiguous else [ptrs[c] for c in (1, 0, 3, 2, 5, 4, 7, 6)]
        for _ in range(5): grp.stage_h(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): grp.stage_h(p)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{kind:13s} contiguous={contiguous}: host {1e6*(t1-t0)/50:7.1f} us per call, total {1e6*(t2-t0)/50:7.1f} us per call ({n/((t2-t0)/50)/1e9:.1f} GB/s)")
