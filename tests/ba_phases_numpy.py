"""CPU restatement of the rank-local phases of the sliced BA (coslam_amd/csrc/ba.hip cs_ba_dist_phase), for the
multi-process tests: numpy + the oracle's residual / Jacobian (oracle/ba_oracle.c oba_residual).  Test infrastructure."""
import numpy as np
import torch

import oracle
from coslam_amd.multicam import (PH_CONTROL0, PH_CONTROL1, PH_COST0, PH_FINAL_PREP, PH_FINISH, PH_FLAG, PH_LIN_SCHUR,
                                 PH_OUTER_END, PH_SOLVE_UPDATE)


def so3_exp(w):
    th = np.linalg.norm(w)
    if th == 0:
        return np.eye(3)
    h = w / th
    st, ct = np.sin(th), 1 - np.cos(th)
    Kx = np.array([[0, -h[2], h[1]], [h[2], 0, -h[0]], [-h[1], h[0], 0]])
    return np.eye(3) + st * Kx + ct * (Kx @ Kx)


class NumpySlicedBA:
    def __init__(self, Ks, Rs, Ts, pts, ptr, cam, xy, nCamsCon, nPtsCon, maxErr, inner, pLo, pHi, add_lambda):
        self.C, self.P, self.nObs = len(Rs), len(pts), len(cam)
        self.Ks = np.asarray(Ks, float).reshape(self.C, 9)
        self.R = np.asarray(Rs, float).reshape(self.C, 3, 3).copy()
        self.T = np.asarray(Ts, float).reshape(self.C, 3).copy()
        self._pts = np.asarray(pts, float).reshape(-1).copy()
        self.ptr, self.cam, self.xy = np.asarray(ptr), np.asarray(cam), np.asarray(xy, float).reshape(-1, 2)
        self.obs_pt = np.repeat(np.arange(self.P), np.diff(self.ptr))
        self.ncon, self.npcon = min(nCamsCon, self.C), min(nPtsCon, self.P)
        self.nc = self.C - self.ncon
        self.n = 6 * self.nc
        self.maxErr, self.inner, self.lo, self.hi, self.add_lambda = maxErr, inner, pLo, pHi, add_lambda
        self._out = np.zeros(max(self.nObs, 1), np.int32)
        self._S = np.zeros(max(self.n * self.n + self.n, 1))
        self._scal = np.zeros(4)
        # torch views for the collectives
        self.S_rhs, self.scal = torch.from_numpy(self._S), torch.from_numpy(self._scal)
        self.pts, self.outlier = torch.from_numpy(self._pts), torch.from_numpy(self._out)
        self.st = dict(lam=1e-3, cost=0.0, inner_it=0, inner_done=0, all_done=0, chol_ok=1, changed=0, nIter=0, nOuter=0,
                       nOut=0, first=1, cost0=0.0)
        self.Rn, self.Tn, self.Mn = self.R.copy(), self.T.copy(), self._pts.copy()
        self.cam_step2 = 0.0
        self.final_cost = None

    def M(self, arr, i):
        return arr[3 * i: 3 * i + 3]

    def own(self, i):
        return self.lo <= i < self.hi

    def cost_own(self, R, T, pts):
        c = 0.0
        for o in range(self.nObs):
            i = self.obs_pt[o]
            if self._out[o] or not self.own(i):
                continue
            j = self.cam[o]
            _, e, _, _ = oracle.ba_residual(self.Ks[j], R[j], T[j], self.M(pts, i), self.xy[o], jac=False)
            c += e @ e
        return c

    def active(self):
        return not self.st["all_done"] and not self.st["inner_done"]

    def phase(self, ph):
        st, n, nc = self.st, self.n, self.nc
        if ph == PH_COST0:
            if st["all_done"]:
                return
            self._scal[0], self._scal[1] = self.cost_own(self.R, self.T, self._pts), 0.0
        elif ph == PH_CONTROL0:
            if st["all_done"]:
                return
            st["cost"], st["lam"], st["inner_it"] = self._scal[0], 1e-3, 0
            st["inner_done"] = 1 if self.inner <= 0 else 0
            if st["first"]:
                st["cost0"], st["first"] = self._scal[0], 0
        elif ph == PH_LIN_SCHUR:
            self._S[:] = 0
            if not self.active():
                return
            S = self._S[: n * n].reshape(n, n)
            rhs = self._S[n * n: n * n + n]
            lam = st["lam"]
            self.W, self.Vinv, self.gp = {}, {}, {}
            for i in range(self.lo, self.hi):
                obs = [o for o in range(self.ptr[i], self.ptr[i + 1]) if not self._out[o]]
                freeP = i >= self.npcon and len(obs) >= 2
                V, g = np.zeros((3, 3)), np.zeros(3)
                loc = []
                for o in obs:
                    j = self.cam[o]
                    _, e, Jc, Jp = oracle.ba_residual(self.Ks[j], self.R[j], self.T[j], self.M(self._pts, i), self.xy[o])
                    if j >= self.ncon:
                        a = 6 * (j - self.ncon)
                        S[a: a + 6, a: a + 6] += Jc.T @ Jc
                        rhs[a: a + 6] += Jc.T @ e
                    if freeP:
                        V += Jp.T @ Jp
                        g += Jp.T @ e
                    loc.append((o, j, Jc, Jp))
                Vi = np.zeros((3, 3))
                if freeP:
                    Vd = V + lam * np.eye(3)
                    if abs(np.linalg.det(Vd)) > 0:
                        Vi = np.linalg.inv(Vd)
                self.Vinv[i], self.gp[i] = Vi, g
                for (o, j, Jc, Jp) in loc:
                    self.W[o] = (Jc.T @ Jp) if (freeP and j >= self.ncon) else np.zeros((6, 3))
                if i >= self.npcon:
                    for (oa, ja, _, _) in loc:
                        if ja < self.ncon:
                            continue
                        Y = self.W[oa] @ Vi
                        a = 6 * (ja - self.ncon)
                        rhs[a: a + 6] -= Y @ g
                        for (ob, jb, _, _) in loc:
                            if jb < self.ncon:
                                continue
                            b = 6 * (jb - self.ncon)
                            S[a: a + 6, b: b + 6] -= Y @ self.W[ob].T
            if self.add_lambda:
                S[np.arange(n), np.arange(n)] += lam
        elif ph == PH_SOLVE_UPDATE:
            if not self.active():
                return
            S = self._S[: n * n].reshape(n, n).copy()
            rhs = self._S[n * n: n * n + n].copy()
            ok = 1
            dc = np.zeros(n)
            if n > 0:
                try:
                    Lc = np.linalg.cholesky(S)
                    dc = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs))
                except np.linalg.LinAlgError:
                    ok = 0
            st["chol_ok"] = ok
            self.Rn, self.Tn, self.Mn = self.R.copy(), self.T.copy(), self._pts.copy()
            self.cam_step2 = 0.0
            for j in range(self.ncon, self.C):
                d = dc[6 * (j - self.ncon): 6 * (j - self.ncon) + 6]
                self.Rn[j] = self.R[j] @ so3_exp(d[:3])
                self.Tn[j] = self.T[j] + d[3:]
                self.cam_step2 += d @ d
            s2 = 0.0
            for i in range(max(self.lo, 0), self.hi):
                if i < self.npcon:
                    continue
                b = self.gp[i].copy()
                for o in range(self.ptr[i], self.ptr[i + 1]):
                    j = self.cam[o] - self.ncon
                    if j < 0 or self._out[o]:
                        continue
                    b -= self.W[o].T @ dc[6 * j: 6 * j + 6]
                d = self.Vinv[i] @ b
                self.Mn[3 * i: 3 * i + 3] = self._pts[3 * i: 3 * i + 3] + d
                s2 += d @ d
            self._scal[0], self._scal[1] = self.cost_own(self.Rn, self.Tn, self.Mn), s2
        elif ph == PH_CONTROL1:
            if not self.active():
                return
            step2 = self._scal[1] + self.cam_step2
            cost_new = self._scal[0] if st["chol_ok"] else 1e300
            acc = bool(st["chol_ok"]) and cost_new <= st["cost"]
            done = 0
            st["nIter"] += 1
            st["inner_it"] += 1
            if acc:
                dec = st["cost"] - cost_new
                st["cost"] = cost_new
                st["lam"] /= 10
                if dec < 1e-9 * cost_new + 1e-15 or step2 < 1e-20:
                    done = 1
                self.R, self.T = self.Rn.copy(), self.Tn.copy()
                self._pts[3 * self.lo: 3 * self.hi] = self.Mn[3 * self.lo: 3 * self.hi]
            else:
                st["lam"] *= 10
                if st["lam"] > 1e12:
                    done = 1
            if st["inner_it"] >= self.inner:
                done = 1
            st["inner_done"] = done
        elif ph == PH_FLAG:
            if st["all_done"]:
                return
            changed, nout = 0, 0
            for o in range(self.nObs):
                i = self.obs_pt[o]
                if not self.own(i):
                    continue
                j = self.cam[o]
                _, e, _, _ = oracle.ba_residual(self.Ks[j], self.R[j], self.T[j], self.M(self._pts, i), self.xy[o], jac=False)
                out = 1 if e @ e > self.maxErr ** 2 else 0
                changed |= int(out != self._out[o])
                self._out[o] = out
                nout += out
            self._scal[2], self._scal[3] = changed, nout
        elif ph == PH_OUTER_END:
            if st["all_done"]:
                return
            st["changed"], st["nOut"] = int(self._scal[2] > 0), int(self._scal[3] + 0.5)
            st["nOuter"] += 1
            st["inner_done"] = 0
            if not st["changed"]:
                st["all_done"] = 1
        elif ph == PH_FINAL_PREP:
            for i in range(self.P):
                if not self.own(i):
                    self._pts[3 * i: 3 * i + 3] = 0
            for o in range(self.nObs):
                if not self.own(self.obs_pt[o]):
                    self._out[o] = 0
        elif ph == PH_FINISH:
            lo, hi = self.lo, self.hi
            self.lo, self.hi = 0, self.P
            self.final_cost = self.cost_own(self.R, self.T, self._pts)
            self.lo, self.hi = lo, hi
        else:
            raise ValueError(ph)
