"""A small multi-camera sequence for the pose-update tests: static and moving scene points, some of them map points, tracked in
slots over T frames (hand-back style records per frame)."""
import numpy as np


def rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


class Scene:
    def __init__(self, nC=3, N=512, nMap=600, T=14, seed=5, W=640, H=480):
        rng = np.random.default_rng(seed)
        self.nC, self.N, self.nMap, self.T, self.W, self.H = nC, N, nMap, T, W, H
        self.K = np.array([[0.82 * W, 0, W / 2.0], [0, 0.82 * W, H / 2.0], [0, 0, 1]])
        self.iK = np.linalg.inv(self.K)
        nPts = nMap + 700   # scene points: the first nMap are map points, the rest are never mapped
        self.P0 = np.stack([rng.uniform(-5, 5, nPts), rng.uniform(-3, 3, nPts), rng.uniform(6, 14, nPts)], 1)
        self.vel = np.zeros((nPts, 3))
        moving = rng.random(nPts) < 0.25
        self.vel[moving] = rng.normal(0, 0.06, (int(moving.sum()), 3))
        self.moving = moving
        # map flags: moving map points are mostly DYNAMIC, a few (wrongly) static; some uncertain / false ones
        fl = np.zeros(nMap, dtype=np.uint8)
        fl[moving[:nMap] & (rng.random(nMap) < 0.8)] |= 1
        fl[rng.random(nMap) < 0.03] |= 4
        fl[rng.random(nMap) < 0.02] = 2
        self.flags0 = fl
        # the map's estimate of the points (static ones: near the truth) and their covariances
        self.map0 = self.P0[:nMap] + rng.normal(0, 0.03, (nMap, 3))
        A = rng.normal(0, 0.03, (nMap, 3, 3))
        self.cov0 = (A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3)).reshape(nMap, 9)
        # poses per frame and camera (truth) and what the pose solve "returned" (truth + a little noise)
        self.Rt, self.tt, self.Re, self.te = [], [], [], []
        for f in range(T):
            Rf, tf, Ref, tef = [], [], [], []
            for c in range(nC):
                a = (c - (nC - 1) / 2.0) * 0.25
                pos = np.array([3 * np.sin(a) + 0.05 * f, 0.1 * c, 1 - np.cos(a) + 0.02 * f])
                z = np.array([0, 0, 10.0]) - pos
                z /= np.linalg.norm(z)
                x = np.array([z[2], 0, -z[0]])
                x /= np.linalg.norm(x)
                R = np.stack([x, np.cross(z, x), z])
                R = rodrigues(np.array([0, 0.003 * f, 0])) @ R
                t = -R @ pos
                Rf.append(R)
                tf.append(t)
                Ref.append(R @ rodrigues(rng.normal(0, 2e-4, 3)))
                tef.append(t + rng.normal(0, 1e-3, 3))
            self.Rt.append(Rf), self.tt.append(tf), self.Re.append(Ref), self.te.append(tef)
        # slots: each slot of a camera follows one scene point from a birth frame on (or is dead)
        self.slotPt = rng.integers(0, nPts, (nC, N))
        for c in range(nC):   # one slot per (camera, scene point): MapPoint::pFeatures[iCam] holds one feature
            _, first = np.unique(self.slotPt[c], return_index=True)
            dup = np.ones(N, dtype=bool)
            dup[first] = False
            self.slotPt[c][dup] = -1
        self.birth = rng.integers(0, max(T - 2, 1), (nC, N))
        self.birth[rng.random((nC, N)) < 0.5] = 0
        self.noise = rng.normal(0, 0.5, (T, nC, N, 2))
        self.gross = rng.random((T, nC, N)) < 0.02

    def point(self, p, f):
        return self.P0[p] + self.vel[p] * f

    def frame(self, f):
        """Per camera: xy float64[2N], state int32[N], slot2map int32[N], trackSpan int32[2N] (frame numbers = f)."""
        out = []
        for c in range(self.nC):
            N = self.N
            xy = np.zeros(2 * N)
            st = np.full(N, -1, dtype=np.int32)
            s2m = np.full(N, -1, dtype=np.int32)
            span = np.full(2 * N, -1, dtype=np.int32)
            R, t = self.Rt[f][c], self.tt[f][c]
            for i in range(N):
                p = self.slotPt[c, i]
                if p < 0 or f < self.birth[c, i]:
                    continue
                X = R @ self.point(p, f) + t
                u = self.K @ X
                m = u[:2] / u[2] + self.noise[f, c, i] + (25.0 if self.gross[f, c, i] else 0.0)
                xy[i], xy[N + i] = m
                st[i] = 1 if f == self.birth[c, i] else 0
                span[i], span[N + i] = self.birth[c, i], f
                if p < self.nMap and st[i] == 0:
                    s2m[i] = p   # (a new track carries no map point)
            out.append(dict(xy=xy, state=st, slot2map=s2m, trackSpan=span))
        return out

    @staticmethod
    def point_feat(recs, nMap):
        """The hand-back's pointFeat table: nMap x nCams, the highest slot of this frame carrying the point, else -1."""
        nC = len(recs)
        pf = np.full((nMap, nC), -1, dtype=np.int32)
        for c, r in enumerate(recs):
            for i in range(len(r["state"])):
                m = r["slot2map"][i]
                if m >= 0 and r["state"][i] in (0, 1):
                    pf[m, c] = max(pf[m, c], i)
        return pf
