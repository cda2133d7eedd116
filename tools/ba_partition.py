"""BA latency on CU-masked streams of different sizes (diagnostic for bench.py's partition choice)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import coslam_amd, bench
from coslam_amd.ba import BAWorkspace
dev = torch.device("cuda:0")
from coslam_amd.synth import make_ba_problem
ba = make_ba_problem(n_cams=bench.BA_KF, n_pts=bench.BA_PTS, seed=0xC051A + 2 + 99)
P = len(ba["pts0"]); obs_pt = np.asarray(ba["obs_pt"]); o = np.argsort(obs_pt, kind="stable")
ptr = np.zeros(P + 1, np.int32); np.add.at(ptr, obs_pt + 1, 1); ptr = np.cumsum(ptr).astype(np.int32)
ws = BAWorkspace(0); ws.upload(ba["Ks"], ba["Rs0"], ba["ts0"], ba["pts0"], ptr, ba["obs_cam"][o], ba["obs_xy"][o])
d_baR = torch.from_numpy(ba["Rs0"].reshape(-1).copy()).to(dev); d_baT = torch.from_numpy(ba["ts0"].reshape(-1).copy()).to(dev)
d_baM = torch.from_numpy(ba["pts0"].reshape(-1).copy()).to(dev)
L = coslam_amd.lib(); L.cs_stream_create_cu_range.restype = C.c_void_p
cases = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(0, 256), (192, 64), (192, 63), (192, 60), (193, 63), (224, 32), (128, 128)]
for first, count in cases:
    h = L.cs_stream_create_cu_range(0, first, count)
    if not h:
        print(first, count, "unavailable", L.cs_last_error().decode()); continue
    for w in range(5):
        ws.solve_dev(h, d_baR.data_ptr(), d_baT.data_ptr(), d_baM.data_ptr(), 2, 2, 6.0, 2, 10)
    torch.cuda.synchronize(); t = time.perf_counter()
    for w in range(50):
        ws.solve_dev(h, d_baR.data_ptr(), d_baT.data_ptr(), d_baM.data_ptr(), 2, 2, 6.0, 2, 10)
    torch.cuda.synchronize()
    print(f"BA on CUs [{first},{first + count}): {(time.perf_counter() - t) / 50 * 1e6:.1f} us")
