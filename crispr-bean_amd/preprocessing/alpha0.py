"""One-off Dirichlet-Multinomial precision estimates (``a0``, ``pi_a0``).

Method-of-moments estimate of the DirMult concentration sum per guide followed
by a straight-line fit of ``log a0 ~ log n`` — the preprocessing the reference
does in ``bean/preprocessing/get_alpha0.py:21-128`` (guide counts) and
``bean/preprocessing/get_pi_alpha0.py:20-152`` (control allele counts).  Runs
once per screen on the host in float64 numpy; not part of the per-step path.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
from scipy.optimize import curve_fit

# fallback trends used when too few guides give a finite estimate
# (get_alpha0.py:103-105, get_pi_alpha0.py:111-112)
FALLBACK_POPT = (-1.510, 0.7861)
FALLBACK_PI_POPT = (-3.214, 0.9873)


def _line(x, b0, b1):
    return b0 + b1 * x


def _finite_pairs(x: np.ndarray, y: np.ndarray):
    ok = np.isfinite(x) & np.isfinite(y)
    return x[ok], y[ok]


def _as64(a) -> np.ndarray:
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


# ------------------------------------------------------------------ guide counts
def depth_normalised_mean(X, size_factor, sample_mask=None) -> np.ndarray:
    """``q[b, g]``: masked mean over replicates of ``X / size_factor``
    (get_alpha0.py:21-39).  X is (R, B, G)."""
    X = _as64(X)
    sf = _as64(size_factor).copy()
    R, B, _ = X.shape
    if sample_mask is None:
        m = np.ones((R, B))
    else:
        m = _as64(sample_mask)
        sf[(m == 0) & (sf == 0)] = 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        return ((X / sf[:, :, None]) * m[:, :, None]).sum(0) / m.sum(0)[:, None]


def depth_normalised_var(X, size_factor, sample_mask=None) -> np.ndarray:
    """``w[b, g]``: masked mean squared deviation from ``q`` (get_alpha0.py:42-53)."""
    X = _as64(X)
    sf = _as64(size_factor).copy()
    R, B, _ = X.shape
    m = np.ones((R, B)) if sample_mask is None else _as64(sample_mask)
    q = depth_normalised_mean(X, sf, None if sample_mask is None else m)
    if sample_mask is not None:
        sf[(m == 0) & (sf == 0)] = 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        se = (X / sf[:, :, None] - q) ** 2
        return (se * m[:, :, None]).sum(0) / m.sum(0)[:, None]


def _shrink(y, y_est, prior_var):
    var = ((y - y_est) ** 2).sum() / (len(y) - 1)
    wt = var / (var + prior_var)
    return wt * y_est + (1 - wt) * y


def fitted_alpha0(
    X,
    size_factor,
    sample_mask=None,
    shrink: bool = False,
    shrink_prior_var: float = 1.0,
    popt: Optional[Sequence[float]] = None,
    verbose: bool = False,
) -> Tuple[np.ndarray, Tuple[float, float]]:
    """Trend-fitted ``a0 (G,)`` and the fitted ``(b0, b1)``
    (get_alpha0.py:70-119)."""
    if sample_mask is not None and (_as64(sample_mask).sum(0) == 0).any():
        raise ValueError("Some bins have no data.")
    X1 = _as64(X) + 1.0
    w = depth_normalised_var(X1, size_factor, sample_mask)
    q = depth_normalised_mean(X1, size_factor, sample_mask)
    with np.errstate(divide="ignore", invalid="ignore"):
        n = np.nanmean(q, axis=0) * q.shape[0]
        p = q / n[None, :]
        r = (w - q) / (n[None, :] * p * (1 - p))
        a0 = ((n - 1) / (r - 1 + 1 / (1 - p)) - 1).mean(0)
        log_n, log_a0 = np.log(n), np.log(a0)
    x, y = _finite_pairs(log_n, log_a0)
    if len(y) < 5:
        if popt is None:
            popt = FALLBACK_POPT
        if verbose:
            print(f"Cannot fit log(a0) ~ log(q): {len(y)} valid values; using {popt}")
    else:
        popt, _ = curve_fit(_line, x, y)
    est = _line(log_n, *popt)
    if shrink:
        yy = np.where(np.isnan(log_a0), est, log_a0)
        est = _shrink(yy, est, shrink_prior_var)
    return np.exp(est), (float(popt[0]), float(popt[1]))


def pred_alpha0(X, size_factor, popt, sample_mask=None) -> np.ndarray:
    """``a0`` predicted from an existing trend (get_alpha0.py:122-128)."""
    q = depth_normalised_mean(_as64(X) + 1.0, size_factor, sample_mask)
    with np.errstate(divide="ignore"):
        return np.exp(_line(np.log(q.sum(0)), *popt))


# --------------------------------------------------------- control allele counts
def _pi_q_w(X, size_factor, sample_mask=None):
    """First-control-condition mean / variance over replicates of
    ``X / size_factor`` for allele counts X (R, C, G, A)
    (get_pi_alpha0.py:20-51; only condition 0 is used)."""
    X = _as64(X)
    sf = _as64(size_factor)
    R, C = X.shape[:2]
    m = np.ones((R, C)) if sample_mask is None else _as64(sample_mask)
    norm = X / sf[:, :, None, None]
    msum = m.sum(0)[:, None, None]
    q = (norm * m[:, :, None, None]).sum(0) / msum
    q0 = q[0]
    se = (norm - q0) ** 2
    w = (se * m[:, :, None, None]).sum(0) / msum
    return q0, w[0]


def fitted_pi_alpha0(
    X,
    size_factor,
    sample_mask=None,
    fit: bool = True,
    fit_quantile: Optional[float] = None,
    shrink: bool = False,
    shrink_prior_var: float = 1.0,
    verbose: bool = False,
):
    """``pi_a0 (G,)`` from control allele counts (get_pi_alpha0.py:78-143)."""
    q, w = _pi_q_w(_as64(X) + 1.0, size_factor, sample_mask)
    n = q.sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = q / n[:, None]
        r = (w - q) / (n[:, None] * p * (1 - p))
        a0 = np.nanmean((n[:, None] - 1) / (r - 1 + 1 / (1 - p)) - 1, axis=-1)
        log_n, log_a0 = np.log(n), np.log(a0)
    if not fit:
        return a0, np.nan
    x, y = _finite_pairs(log_n, log_a0)
    if len(y) < 10:
        popt = list(FALLBACK_PI_POPT)
        if verbose:
            print(f"Cannot fit log(pi_a0) ~ log(q): data too sparse; using {popt}")
    else:
        popt, _ = curve_fit(_line, x, y)
    if fit_quantile is not None:
        resid = y - _line(x, *popt)
        sel = np.where(resid < np.quantile(resid, fit_quantile))[0]
        if len(sel) >= 5:
            popt, _ = curve_fit(_line, x[sel], y[sel])
    est = _line(log_n, *popt)
    if shrink:
        yy = np.where(np.isnan(log_a0), est, log_a0)
        # get_pi_alpha0.py:68-75: element-wise variance (no sum), divided by len(y)
        var = (yy - est) ** 2 / len(yy)
        wt = var / (var + shrink_prior_var)
        est = wt * est + (1 - wt) * yy
    return np.exp(est), (float(popt[0]), float(popt[1]))


def pred_pi_alpha0(X, size_factor, popt, sample_mask=None) -> np.ndarray:
    """``pi_a0`` predicted from an existing trend (get_pi_alpha0.py:146-152)."""
    q, _ = _pi_q_w(_as64(X) + 1.0, size_factor, sample_mask)
    with np.errstate(divide="ignore"):
        return np.exp(_line(np.log(q.sum(-1)), *popt))
