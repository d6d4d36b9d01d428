"""Mirrors footprint_tools/stats/utils.pyx (v1.3.7): host-side index bookkeeping around the
scan (no floating-point arithmetic of the path happens here)."""
import numpy as np


def segment(x, threshold, w=1, decreasing=0):
    """utils.pyx:15-50: runs of elements passing `threshold`, widened by w-1, merged."""
    x = np.asarray(x, dtype=np.float64)
    d = -1.0 if decreasing else 1.0
    passing = d * x >= d * threshold
    # a run opens at the first passing element; it closes at the first element that is
    # strictly failing (NaN neither opens nor closes... it fails `>=` and fails `<`)
    failing = d * x < d * threshold
    ret = []
    curr_start = -1
    idx = np.flatnonzero(passing | failing)
    for i in idx:
        if curr_start < 0:
            if passing[i]:
                curr_start = int(i) - w + 1
        elif failing[i]:
            if len(ret) > 0 and curr_start <= ret[-1][1]:
                ret[-1][1] = int(i) - 1 + w
            else:
                ret.append([curr_start, int(i) - 1 + w])
            curr_start = -1
    return ret


def bisect(a, b):
    """utils.pyx:52-79: two-pointer count of leading a[lo] that are not greater than b[i]
    (`lo` persists across i; b[i] < a[lo] stops the advance, so NaNs never stop it)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    out = np.zeros(b.shape[0], dtype=np.float64)
    # position of the first a[lo] with b[i] < a[lo], scanning from the previous lo
    stop = np.empty(b.shape[0], dtype=np.int64)
    if a.size and not np.isnan(a).any() and np.all(a[1:] >= a[:-1]):
        stop = np.searchsorted(a, b, side="right")
        stop[np.isnan(b)] = a.size  # NaN < a[lo] is never true
        out[:] = np.maximum.accumulate(stop)
        return out
    lo, hi = 0, a.shape[0]
    for i in range(b.shape[0]):
        while lo < hi:
            if b[i] < a[lo]:
                break
            lo += 1
        out[i] = lo
    return out
