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


# ---- maximum-likelihood fit (host side, like the reference: nbinom.pyx:25-80) ----------------

def mle(par, data, sm):
    """Score equations of the NB likelihood in (p, r) for `data` with mean `sm`; both are zero at
    the maximum-likelihood estimate."""
    import scipy.special
    p, r = par[0], par[1]
    data = np.asarray(data, dtype=np.float64)
    n = data.shape[0]
    dig = scipy.special.psi(data + r)
    return np.array([sm / (r + sm) - p,
                     np.sum(dig) - n * scipy.special.psi(r) + n * np.log(r / (r + sm))])


def fit(data, p=None, r=None):
    """Maximum-likelihood (p, r) of a negative binomial for the 1-D array `data`; p, r are the
    starting point of the root finder (moment estimates when not given)."""
    import warnings

    import scipy.optimize
    data = np.asarray(data, dtype=np.float64)
    if p is None or r is None:
        m1, m2 = np.average(data), np.var(data)
        r = (m1 * m1) / (m2 - m1)
        p = (m2 - m1) / m2
    sm = np.sum(data) / len(data)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        root = scipy.optimize.fsolve(mle, np.array([p, r]), args=(data, sm))
    return (root[0], root[1])


def rvs(p, r):
    """nbinom.pyx:174-189: declared and not implemented in the reference either (null counts are drawn by
    `dispersion_model.sample`)."""
    raise NotImplementedError
