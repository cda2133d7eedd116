#!/usr/bin/env python
"""Stand-alone timing of cs_register_search_dev at the headline's size (1536 points x 8 cameras x 2000 slots)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import coslam_amd
from test_register_gpu import _rig, MODES, W, H

dev = torch.device("cuda:0")
nC, N, P = 8, 2000, 1536
Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf = _rig(nC, N, P, seed=5)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d = dict(K=T(Ks), R=T(Rs), t=T(ts), xy=[T(a) for a in xy], st=[T(a) for a in state], s2=[T(a) for a in s2m], M=T(Ms), cov=T(covs), pf=T(pf))
out = dict(slot=torch.zeros(P * nC, dtype=torch.int32, device=dev), m=torch.zeros(P * nC * 2, dtype=torch.float64, device=dev),
           var=torch.zeros(P * nC * 4, dtype=torch.float64, device=dev), dist=torch.zeros(P * nC, dtype=torch.float64, device=dev),
           flags=torch.zeros(P * nC, dtype=torch.int32, device=dev))
cams = [dict(K=d["K"].data_ptr() + 72 * c, R=d["R"].data_ptr() + 72 * c, t=d["t"].data_ptr() + 24 * c, xy=d["xy"][c].data_ptr(),
             state=d["st"][c].data_ptr(), slot2map=d["s2"][c].data_ptr()) for c in range(nC)]
s = torch.cuda.current_stream().cuda_stream
sS, mD, sM = MODES["static"]
def run():
    coslam_amd.register_search_dev(s, cams, N, W, H, P, d["M"].data_ptr(), d["cov"].data_ptr(), d["pf"].data_ptr(), sS, mD, sM,
                                   out["slot"].data_ptr(), out["m"].data_ptr(), out["var"].data_ptr(), out["dist"].data_ptr(), out["flags"].data_ptr())
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print(f"cs_register_search_dev {P} x {nC} x {N}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch (back to back)")
import oracle
t0 = time.perf_counter()
oracle.register_search(W, H, Ks, Rs, ts, xy, state, s2m, dyn, Ms, covs, pf, sS, mD, sM)
print(f"oracle (1 host core): {(time.perf_counter() - t0) * 1e3:.1f} ms")
