"""The reference's new-map-point scenes (tests/golden/newpts_golden.npz, made by oracle/_ref/ref_newpts_test: the reference's own
featTracksFromMatches + NewMapPtsNCC::reconstructTracks + decidePointType) as the structure-of-arrays records the oracle restatement
and cs_newpts_from_pairs_dev take.  Slots [0, N) of a camera are the reference's candidate list; behind them the features of this
frame that belong to map points which exist already: certain dynamic ones (they draw decidePointType's mask) and uncertain-dynamic /
static ones (they must not).  Every such feature gets a map point of its own, so no pair has seeds: the matches go in as candidate
lists with equal scores and the unguided greedy walk takes all of them (they are one-to-one)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "newpts_golden.npz")


def scene(g, sc, spare=1024):
    k = lambda n: g[f"s{sc}_{n}"]   # noqa: E731
    nc, N, frame, W, H = (int(v) for v in k("dims"))
    extra = max(len(k(f"dyn{c}")) + len(k(f"other{c}")) for c in range(nc))
    NS = N + extra
    n_old = sum(len(k(f"dyn{c}")) + len(k(f"other{c}")) for c in range(nc))
    cap = n_old + spare
    xy = [np.zeros(2 * NS) for _ in range(nc)]
    state = [np.full(NS, -1, dtype=np.int32) for _ in range(nc)]
    s2m = [np.full(NS, -1, dtype=np.int32) for _ in range(nc)]
    is_static = [np.ones(NS, dtype=np.uint8) for _ in range(nc)]
    flags = np.zeros(cap, dtype=np.uint8)
    pf = np.full((cap, nc), -1, dtype=np.int32)
    m = 0
    for c in range(nc):
        xy[c][:N], xy[c][NS:NS + N] = k("xy")[c][:N], k("xy")[c][N:]
        state[c][:N] = 0
        is_static[c][:N] = k("isStatic")[c]
        s = N
        for name, fl in ((f"dyn{c}", 1), (f"other{c}", None)):
            for q, (x, y) in enumerate(k(name)):
                xy[c][s], xy[c][NS + s], state[c][s], s2m[c][s] = x, y, 0, m
                flags[m] = fl if fl is not None else (5 if q & 1 else 0)   # uncertain dynamic / certain static
                pf[m, c] = s
                s, m = s + 1, m + 1
    assert m == n_old
    pairs = [[(int(i), int(j), 0.0, 0.9) for i, j in k(f"match{a}")] for a in range(nc - 1)]
    K = k("K").reshape(3, 3)
    want = dict(track_len=k("track_len"), track_views=k("track_views"), M=k("new_M"), cov=k("new_cov"), flags=k("new_flags"), first=k("new_first"),
                feat=k("new_feat"), reproj=k("reproj"))
    return dict(nc=nc, N=N, NS=NS, frame=frame, W=W, H=H, K=K, iK=np.linalg.inv(K), R=k("R"), t=k("t"), xy=xy, state=state, s2m=s2m,
                is_static=is_static, flags=flags, pf=pf, n_old=n_old, cap=cap, pairs=pairs, want=want)


def exact_inverse_of(K):
    """the reference driver's iK for its K (no skew): 1 / fx, 0, -cx / fx; 0, 1 / fy, -cy / fy; 0, 0, 1"""
    return np.array([[1 / K[0, 0], 0, -K[0, 2] / K[0, 0]], [0, 1 / K[1, 1], -K[1, 2] / K[1, 1]], [0, 0, 1]])
