"""Negative binomial scalars.  Mirrors footprint_tools/stats/distributions/nbinom.pyx:82-172
(v1.3.7); k, p, r may also be arrays (evaluated element-wise on the GPU)."""
import numpy as np

from ... import _lib


def _eval(what, k, p, r):
    ctx = _lib.get_ctx()
    k, p, r = np.broadcast_arrays(np.asarray(k), np.asarray(p, dtype=np.float64),
                                  np.asarray(r, dtype=np.float64))
    scalar = k.ndim == 0
    # `int k` argument: Python int -> C int
    k32 = np.ascontiguousarray(np.asarray(k, dtype=np.int64).astype(np.int32)).ravel()
    pp, rr = _lib.f64(p).ravel(), _lib.f64(r).ravel()
    out = np.empty(k32.size)
    _lib.check(ctx.L.fpt_nb_scalar(ctx.h, what, _lib.ptr(k32), _lib.ptr(pp), _lib.ptr(rr), k32.size,
                                   _lib.ptr(out)))
    return float(out[0]) if scalar else out.reshape(k.shape)


def logpmf(k, p, r):
    return _eval(_lib.NB_LOGPMF, k, p, r)


def pmf(k, p, r):
    return _eval(_lib.NB_PMF, k, p, r)


def cdf(k, p, r):
    return _eval(_lib.NB_CDF, k, p, r)


def mean(p, r):
    return p * r / (1 - p)


def var(p, r):
    return (p * r) / ((1 - p) * (1 - p))
