"""Inter-camera merge step for the one-camera-per-GPU layout.

The reference tracks all cameras serially inside one process and reads every camera's features and pose
directly (src/app/SL_CoSLAM.cpp:299-305, SL_InterCamPoseEstimator.cpp:24-37).  With one camera per rank the
same information is exchanged once per frame with a single all-gather of a fixed-size record:
    N x KLT_TrackedFeature (20 B each)  ||  R (9 f64)  ||  t (3 f64)
packed as int32 words.  `torch.distributed` backend "nccl" is RCCL over xGMI on MI355X; the CPU tests run the
same code over gloo.  The payload is ~40 KB per rank: latency-bound, so it is ONE collective per frame.
"""
import numpy as np
import torch
import torch.distributed as dist

FEATURE_WORDS = 5   # sizeof(KLT_TrackedFeature) / 4
POSE_WORDS = 24     # 12 doubles


def record_words(n_features):
    return n_features * FEATURE_WORDS + POSE_WORDS


class CameraExchange:
    """Preallocated send/recv buffers + the per-frame all-gather."""

    def __init__(self, n_features, device, group=None):
        self.n = n_features
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        w = record_words(n_features)
        self.send = torch.zeros(w, dtype=torch.int32, device=device)
        self.recv = torch.zeros(w * self.world, dtype=torch.int32, device=device)

    def pack(self, dest_words, R, t):
        """dest_words: int32[N*5] view of the KLT_TrackedFeature array; R: f64[9]; t: f64[3] (same device)."""
        nf = self.n * FEATURE_WORDS
        self.send[:nf].copy_(dest_words, non_blocking=True)
        self.send[nf: nf + 18].copy_(R.view(torch.int32), non_blocking=True)
        self.send[nf + 18:].copy_(t.view(torch.int32), non_blocking=True)

    def all_gather(self):
        if self.world == 1:
            self.recv.copy_(self.send, non_blocking=True)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        return self.recv

    def unpack(self, cam):
        """-> (features int32[N,5] view, R f64[9], t f64[3]) of camera `cam` from the last all_gather."""
        w = record_words(self.n)
        rec = self.recv[cam * w: (cam + 1) * w]
        nf = self.n * FEATURE_WORDS
        return rec[:nf].view(self.n, FEATURE_WORDS), rec[nf: nf + 18].view(torch.float64), rec[nf + 18:].view(torch.float64)


def features_from_words(words):
    """int32[N,5] (host tensor / array) -> numpy structured KLT_TrackedFeature[N]"""
    from .klt import KLT_TrackedFeature

    a = words.cpu().numpy() if hasattr(words, "cpu") else np.asarray(words)
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1).view(KLT_TrackedFeature)
