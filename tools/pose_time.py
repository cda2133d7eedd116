"""k_intracam on its own: 8 cameras x ~150 correspondences (the headline's hand-back output size), started a frame's motion away
from the solution; HIP-event time per launch and the LM steps it took (cs_pose_option.verboseRW on return)."""
import ctypes as C
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coslam_amd  # noqa: E402
from coslam_amd.pose import IntraCamPoseOption, intraCamEstimate_batch_dev  # noqa: E402
from coslam_amd.synth import Scene, rodrigues  # noqa: E402

nC, stride, npts = 8, 192, int(sys.argv[1]) if len(sys.argv) > 1 else 150
sc = Scene(nC, 640, 480, 7000, seed=0xC051A + 2, sigma=1.0, loop_period=120)
rng = np.random.default_rng(1)
dev = torch.device("cuda:0")
Ms, ms, R0, t0, nn = np.zeros((nC, stride, 3)), np.zeros((nC, stride, 2)), np.zeros((nC, 9)), np.zeros((nC, 3)), np.zeros(nC, np.int32)
for c in range(nC):
    uv, vis = sc.project(c, 31)
    idx = rng.choice(np.nonzero(vis)[0], npts, replace=False)
    Ms[c, :npts], ms[c, :npts] = sc.points[idx], uv[idx] + rng.normal(0, 0.4, (npts, 2))
    ms[c, :6] += 30.0   # a few gross outliers: the Tukey rounds have something to do
    R, t = sc.pose(c, 30)   # the previous frame's pose
    R0[c], t0[c], nn[c] = R.ravel(), t, npts
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
dK = d(np.tile(sc.K.ravel(), nC))
dMs, dms, dR0, dt0, dn = d(Ms), d(ms), d(R0), d(t0), d(nn)
opt0 = IntraCamPoseOption()
coslam_amd.lib().cs_pose_option_default(C.byref(opt0))
dopt0 = torch.from_numpy(np.frombuffer(bytes(opt0) * nC, dtype=np.uint8).copy()).to(dev)
dopt = dopt0.clone()
dR, dt, dok = torch.zeros_like(dR0), torch.zeros_like(dt0), torch.zeros(nC, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream


def run():
    dopt.copy_(dopt0)
    intraCamEstimate_batch_dev(s, nC, stride, dK.data_ptr(), dR0.data_ptr(), dt0.data_ptr(), dn.data_ptr(), 0, dMs.data_ptr(), dms.data_ptr(), 10.0,
                               dR.data_ptr(), dt.data_ptr(), dopt.data_ptr(), dok.data_ptr())


for _ in range(20):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
tot = 0.0
for _ in range(n):
    dopt.copy_(dopt0)
    e0.record()
    intraCamEstimate_batch_dev(s, nC, stride, dK.data_ptr(), dR0.data_ptr(), dt0.data_ptr(), dn.data_ptr(), 0, dMs.data_ptr(), dms.data_ptr(), 10.0,
                               dR.data_ptr(), dt.data_ptr(), dopt.data_ptr(), dok.data_ptr())
    e1.record()
    e1.synchronize()
    tot += e0.elapsed_time(e1)
opts = [IntraCamPoseOption.from_buffer_copy(dopt[96 * c: 96 * c + 96].cpu().numpy().tobytes()) for c in range(nC)]
steps = [o.verboseRW for o in opts]
terr = max(np.abs(dt.cpu().numpy()[c] - sc.pose(c, 31)[1]).max() for c in range(nC))
print(f"k_intracam x {nC} cameras x {npts} pts: {tot / n * 1e3:.1f} us per launch; rounds {[o.nIterRW for o in opts]}, LM steps {steps} "
      f"-> {tot / n * 1e3 / max(steps):.2f} us per LM step of the slowest camera; ok {dok.cpu().tolist()}, |t - truth| {terr:.4f}")
