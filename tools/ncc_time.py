#!/usr/bin/env python
"""Stand-alone timing of the NCC stage at the headline's size: blocks of 2 x 2000 features, then the 2000 x 2000 matrices."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coslam_amd, oracle

dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
W, H, n, scale = 192, 144, 2000, 0.3
img = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(2)]
xy = [(rng.uniform(20, W / scale - 20, n), rng.uniform(20, H / scale - 20, n)) for _ in range(2)]
F = rng.normal(size=9)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d_img = [T(a) for a in img]
d_x = [T(a[0]) for a in xy]
d_y = [T(a[1]) for a in xy]
d_blk = [torch.zeros(n * 128, dtype=torch.uint8, device=dev) for _ in range(2)]
d_abc = [torch.zeros(n * 4, dtype=torch.float64, device=dev) for _ in range(2)]
d_val = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
d_epi = torch.zeros(n * n, dtype=torch.float64, device=dev)
d_ncc = torch.zeros(n * n, dtype=torch.float64, device=dev)
st = torch.cuda.Stream(device=dev)
s = st.cuda_stream


def blocks():
    for c in range(2):
        coslam_amd.ncc_blocks_dev(s, d_img[c].data_ptr(), W, H, n, d_x[c].data_ptr(), d_y[c].data_ptr(), scale, d_blk[c].data_ptr(),
                                  d_abc[c].data_ptr(), d_val[c].data_ptr())


def mats():
    coslam_amd.ncc_epi_mat_dev(s, F, n, d_x[0].data_ptr(), d_y[0].data_ptr(), d_blk[0].data_ptr(), d_abc[0].data_ptr(),
                               d_val[0].data_ptr(), n, d_x[1].data_ptr(), d_y[1].data_ptr(), d_blk[1].data_ptr(), d_abc[1].data_ptr(),
                               d_val[1].data_ptr(), 50.0, 0.8, -1.0, d_epi.data_ptr(), d_ncc.data_ptr())


for fn, name in ((blocks, "cs_ncc_blocks_dev x 2 cameras"), (mats, f"cs_ncc_epi_mat_dev {n} x {n}")):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us")
print(f"  matrices: {16 * n * n / 1e6:.0f} MB written per call")
t0 = time.perf_counter()
b0 = oracle.ncc_blocks(img[0], *xy[0], scale)
b1 = oracle.ncc_blocks(img[1], *xy[1], scale)
t1 = time.perf_counter()
oracle.ncc_epi_mat(F.reshape(3, 3), *xy[0], *b0, *xy[1], *b1, 50.0, 0.8)
print(f"oracle (1 host core): blocks {1e3 * (t1 - t0):.1f} ms (python loop), matrices {1e3 * (time.perf_counter() - t1):.1f} ms")
