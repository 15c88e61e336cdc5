"""Test helper: the trust-region control of the device solvers (csrc/dev_cost.h lm_begin / lm_propose / lm_consume — Ceres'
TrustRegionMinimizer + LevenbergMarquardtStrategy on the normal equations) in numpy, driven by a callback that returns the 28
summed scalars (upper triangle of J^T J, J^T r, cost) of an evaluation.  Used by the world_size-2 gloo test of the sharded
registration, where the callback all-reduces the two ranks' partial sums."""
import numpy as np


def _tri(H21):
    H = np.zeros((6, 6))
    t = 0
    for i in range(6):
        for j in range(i, 6):
            H[i, j] = H[j, i] = H21[t]
            t += 1
    return H


def solve(evaluate, x0, max_iter):
    """returns (x, dict(iterations, successful, termination, initial_cost, final_cost))"""
    x = np.array(x0, np.float64)
    acc = evaluate(x)
    H, g, x_cost = _tri(acc[:21]), acc[21:27].copy(), float(acc[27])
    info = dict(iterations=0, successful=0, termination=0, initial_cost=x_cost, final_cost=x_cost)
    if not np.isfinite(x_cost):
        info["termination"] = 4
        return x, info
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))          # jacobi scaling, iteration 0 only
    x_norm = float(np.linalg.norm(x))
    gmax = float(np.max(np.abs(x - (x - g))))
    radius, dec, it, num_invalid, step_ok = 1e4, 2.0, 0, 0, True
    while True:
        if it >= max_iter:
            info["termination"] = 0
            break
        if step_ok and gmax <= 1e-10:
            info["termination"] = 1
            break
        if radius <= 1e-32:
            info["termination"] = 5
            break
        it += 1
        Hs = H * np.outer(scale, scale)
        gs = g * scale
        A = Hs + np.diag(np.clip(np.diag(Hs), 1e-6, 1e32) / radius)
        try:
            Lc = np.linalg.cholesky(A)
            y = np.linalg.solve(Lc.T, np.linalg.solve(Lc, gs))
            step = -y
            ok = bool(np.all(np.isfinite(step)))
        except np.linalg.LinAlgError:
            ok, step = False, np.zeros(6)
        mcc = -(step @ gs) - 0.5 * (step @ Hs @ step) if ok else 0.0
        if not ok or not (mcc > 0.0):
            num_invalid += 1
            if num_invalid >= 5:
                info["termination"] = 4
                break
            radius /= dec; dec *= 2.0; step_ok = False
            continue
        num_invalid = 0
        cand = x + step * scale
        acc = evaluate(cand)
        cand_cost = float(acc[27]) if np.isfinite(acc[27]) else 1.7976931348623157e308
        if np.linalg.norm(x - cand) <= 1e-8 * (x_norm + 1e-8):
            info["termination"] = 2
            break
        change = x_cost - cand_cost
        if abs(change) <= 1e-6 * x_cost:
            info["termination"] = 3
            break
        rd = change / mcc
        if rd > 1e-3:
            x, x_cost, x_norm = cand, cand_cost, float(np.linalg.norm(cand))
            H, g = _tri(acc[:21]), acc[21:27].copy()
            gmax = float(np.max(np.abs(x - (x - g))))
            step_ok = True
            info["successful"] += 1
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rd - 1.0) ** 3))
            dec = 2.0
        else:
            step_ok = False
            radius /= dec; dec *= 2.0
    info["iterations"], info["final_cost"] = it, x_cost
    return x, info
