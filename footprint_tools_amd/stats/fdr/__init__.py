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
