#!/usr/bin/env python
"""Diagnostic: is the frame schedule run-to-run deterministic?  Runs the same sequence several times, diffs dest[]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, coslam_amd
dev = torch.device("cuda:0")
sc, frames, *_ = bench.build_inputs(0, 1, 0xC051A + 2)
order = bench.frame_order(bench.N_FRAMES)
d_frames = torch.from_numpy(frames).to(dev)
stream = torch.cuda.current_stream().cuda_stream
def run(fused, n=8, graphs=0):
    d_dest = torch.zeros(2000 * 5, dtype=torch.int32, device=dev); d_counts = torch.zeros(4, dtype=torch.int32, device=dev)
    trk = coslam_amd.KLT_SequenceTracker(bench.klt_config(), 0)
    trk.allocate(640, 480, 4, 50, 40); trk.set_stream(stream); trk.set_fused(fused); trk.enable_graphs(graphs)
    out = []
    trk.detect_dev(d_frames[0].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr()); trk.advanceFrame()
    out.append((d_dest.cpu().numpy().view(coslam_amd.KLT_TrackedFeature).copy(), trk.read_features().copy(), d_counts.cpu().numpy().copy()))
    for i in range(n):
        trk.redetect_dev(d_frames[order[(i + 1) % len(order)]].data_ptr(), d_dest.data_ptr(), d_counts.data_ptr())
        trk.advanceFrame()
        out.append((d_dest.cpu().numpy().view(coslam_amd.KLT_TrackedFeature).copy(), trk.read_features().copy(), d_counts.cpu().numpy().copy()))
    trk.close()
    return out
def diff(A, B, tag):
    for f, ((da, fa, ca), (db, fb, cb)) in enumerate(zip(A, B)):
        live = (da["status"] >= 0) | (db["status"] >= 0)
        bad = np.nonzero((da["status"] != db["status"]) | (live & ((da["pos"] != db["pos"]).any(1) | (da["gain"] != db["gain"]))))[0]
        fbad = np.nonzero((fa != fb).any(1))[0]
        if len(bad) or len(fbad) or (ca != cb).any():
            print(f"{tag}: frame {f}: {len(bad)} dest slots differ, {len(fbad)} feature rows differ, counts {ca} vs {cb}")
            for i in bad[:5]:
                print("   slot", i, da[i], db[i])
            for i in fbad[:3]:
                print("   feat", i, fa[i], fb[i])
            return False
    print(f"{tag}: identical over {len(A)} frames")
    return True
r = [run(1) for _ in range(3)]
diff(r[0], r[1], "fused run0 vs run1"); diff(r[1], r[2], "fused run1 vs run2")
p = [run(0) for _ in range(2)]
diff(p[0], p[1], "per-pass run0 vs run1")
diff(r[0], p[0], "fused vs per-pass")
g = run(1, graphs=1)
diff(r[0], g, "fused eager vs graphs")
