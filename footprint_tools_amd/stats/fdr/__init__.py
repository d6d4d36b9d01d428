"""Empirical FDR: the call surface of footprint_tools/stats/fdr/__init__.py:12-33 (v1.3.7).
For whole batches on the GPU use `FootprintScanner.fdr` / `fpt_fdr_dev` instead."""
import numpy as np

from ..utils import bisect


def emperical_fdr(pvals_null, pvals):
    """Fraction of the pooled null p-values that are <= each observed p-value, capped at 1;
    NaN observations rank above everything (-> 1)."""
    null = np.sort(np.asarray(pvals_null, dtype=np.float64), axis=None)
    p = np.asarray(pvals, dtype=np.float64)
    order = np.argsort(p)
    rate = bisect(null, p[order]) / null.size
    out = np.empty(p.shape, dtype=np.float64)
    out[order] = np.minimum(rate, 1.0)
    return out


def pi0est(pvals, lamb=None):
    """Storey's estimate of the share of true nulls (stats/fdr/__init__.py:39-64): the share of p-values at or
    above each lambda over (1 - lambda), the lambda chosen where the bootstrap mean squared error against the
    10th percentile of those estimates is smallest; capped at 1.  Returned as the reference returns it: the
    one-element array of the chosen estimate (or 1)."""
    p = np.asarray(pvals, dtype=np.float64)
    n = p.size
    lam = np.arange(0.05, 1, 0.05) if lamb is None else np.asarray(lamb, dtype=np.float64)
    above = (p[None, :] >= lam[:, None]).sum(axis=1)          # W(lambda)
    pi0 = (above / n) / (1.0 - lam)
    floor = np.percentile(pi0, q=10)
    mse = (above / (n ** 2 * (1.0 - lam) ** 2)) * (1.0 - above / n) + (pi0 - floor) ** 2
    best = pi0[mse == mse.min()]
    if best.size != 1:   # (the reference's `min(array, 1)` cannot decide then; the first minimum is taken here)
        best = best[:1]
    return best if best[0] <= 1 else 1   # the reference's min(array, 1): the array unless it exceeds 1


def qvalue(pvals):
    """Storey q-values as stats/fdr/__init__.py:67-95 computes them: pi0 n p / (rank (1 - (1 - p)^n)) with the
    rank counting ties at their highest place, made monotone from the largest p-value down and capped at 1
    there."""
    p = np.asarray(pvals, dtype=np.float64)
    n = p.size
    pi0 = pi0est(p)
    order = np.argsort(p)
    ps = p[order]
    rank = np.searchsorted(ps, ps, side="right")               # rankdata(method="max") of the sorted values
    q = (pi0 * n * ps) / (rank * (1.0 - (1.0 - ps) ** n))
    q[-1] = min(q[-1], 1.0)
    q = np.minimum.accumulate(q[::-1])[::-1]
    out = np.empty(n, dtype=np.float64)
    out[order] = q
    return out


def bh_qvalue(pvals):
    """Benjamini-Hochberg adjusted p-values (stats/fdr/__init__.py:98-131): m p_(j) / j made monotone from the
    largest p-value down (which is kept as it is).  The reference's own body fails under Python 3 -- `sorted`
    with a positional comparison argument -- so this is what that body computes under Python 2.  ValueError for
    p-values outside [0, 1]: EVERY value is checked here (the reference looks at the first and the last of the
    unsorted input only); NaN values are ordered last by the sort rather than rejected."""
    p = np.asarray(pvals, dtype=np.float64)
    m = p.size
    if m == 0:
        return np.zeros(0)
    if np.nanmin(p) < 0 or np.nanmax(p) > 1:
        raise ValueError("P-values must be between 0 and 1")
    order = np.argsort(p, kind="stable")
    coeff = m * p[order] / np.arange(1, m + 1)
    coeff[-1] = p[order][-1]
    out = np.empty(m, dtype=np.float64)
    out[order] = np.minimum.accumulate(coeff[::-1])[::-1]
    return out
