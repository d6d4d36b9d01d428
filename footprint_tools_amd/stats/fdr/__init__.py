"""Mirrors footprint_tools/stats/fdr/__init__.py:12-33 (v1.3.7)."""
import numpy as np

from ..utils import bisect


def emperical_fdr(pvals_null, pvals):
    sorted_pvals_null = np.sort(np.ravel(pvals_null))
    sorted_pvals_idx = np.argsort(pvals)
    counts = bisect(sorted_pvals_null, np.asarray(pvals)[sorted_pvals_idx])
    false_positive_rates = counts / len(sorted_pvals_null)
    false_positive_rates[false_positive_rates > 1] = 1
    return false_positive_rates[np.argsort(sorted_pvals_idx)]
