#!/usr/bin/env python
"""Driver for rocprofv3 --pmc passes: the KLT stage of the 8-camera group only (redetect loop with frame-front prefetch:
k_track_rows_fused, k_tail_nonmax_level0, k_tail_select_down; the first frames also show the stand-alone front kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
sys.argv = sys.argv[:1]
from tools import group_cam as G

n_cams = int(os.environ.get("PMC_CAMS", "8"))
prefetch = os.environ.get("PMC_PREFETCH", "1") != "0"
G.run(n_cams, n_frames=30, prefetch=prefetch, profile=False)
torch.cuda.synchronize()
