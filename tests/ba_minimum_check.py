"""An independent check of where a robust BA ended: the cost written again from scratch in numpy (pinhole projection,
Rodrigues; no code shared with oracle/ba_oracle.c or the kernels), over the measurements the solve left as inliers, with the gauge
(first cameras, first points) held as the solve held it.  At the result (a) the central-difference gradient over every free parameter is
zero against the cost's scale, and (b) scipy's trust-region least squares, started there, finds nothing better."""
import numpy as np


def _rot(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def assert_is_a_minimum(Ks, ptr, cam, xy, ncon, npcon, R, T, M, out, reported_cost, grad_tol=2e-3, gain_tol=1e-6):
    from scipy.optimize import least_squares

    R, T, M = np.asarray(R, float).reshape(-1, 3, 3), np.asarray(T, float).reshape(-1, 3), np.asarray(M, float).reshape(-1, 3)
    Cn, P = len(R), len(M)
    pt_of = np.repeat(np.arange(P), np.diff(ptr))
    keep = np.asarray(out) == 0
    Ks = np.asarray(Ks, float).reshape(Cn, 3, 3)
    nfc, nfp = Cn - ncon, P - npcon

    def residuals(x):
        Rs, Ts, Ms = R.copy(), T.copy(), M.copy()
        for c in range(nfc):
            Rs[ncon + c] = R[ncon + c] @ _rot(x[6 * c:6 * c + 3])
            Ts[ncon + c] = T[ncon + c] + x[6 * c + 3:6 * c + 6]
        Ms[npcon:] = M[npcon:] + x[6 * nfc:].reshape(nfp, 3)
        Xc = np.einsum("nij,nj->ni", Rs[cam], Ms[pt_of]) + Ts[cam]
        uvw = np.einsum("nij,nj->ni", Ks[cam], Xc)
        r = np.asarray(xy, float).reshape(-1, 2) - uvw[:, :2] / uvw[:, 2:3]
        return r[keep].ravel()

    x0 = np.zeros(6 * nfc + 3 * nfp)
    r0 = residuals(x0)
    cost = float(r0 @ r0)
    assert abs(cost - reported_cost) < 1e-6 * max(cost, 1.0), (cost, reported_cost)   # the number the solve reports for its inliers
    g = np.zeros_like(x0)
    for k in range(len(x0)):
        d = np.zeros_like(x0)
        d[k] = 1e-6
        rp, rm = residuals(d), residuals(-d)
        g[k] = (rp @ rp - rm @ rm) / 2e-6
    assert np.max(np.abs(g)) < grad_tol * max(cost, 1.0), np.max(np.abs(g))
    sol = least_squares(residuals, x0, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=50)
    assert cost - 2 * sol.cost < gain_tol * max(cost, 1.0), (cost, 2 * sol.cost)
    assert np.max(np.abs(sol.x)) < 1e-3
    return cost
