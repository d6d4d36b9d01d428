"""Dispersion model.  Mirrors footprint_tools/modeling/dispersion.pyx (v1.3.7) for everything
on the scan path; the per-base negative-binomial evaluations (hcephes incbet / lgam) run on
the GPU.  `learn_dispersion_model` (a one-off host fit needing pwlf) is out of scope."""
import base64

import numpy as np

from .. import _lib


def _masked_sum(x, par, nseg):
    # dispersion.pyx:26-57: sum_s mask_s * (y_s + k_s * x) in Python float arithmetic
    x = float(x)
    brk, icpt, slope = par[:nseg], par[nseg:2 * nseg], par[2 * nseg:3 * nseg]
    acc = None
    for s in range(nseg):
        if s == 0:
            m = x < brk[0]
        elif s == nseg - 1:
            m = x >= brk[s - 1]
        else:
            m = (x >= brk[s - 1]) & (x < brk[s])
        term = float(m) * (float(icpt[s]) + float(slope[s]) * x)
        acc = term if acc is None else acc + term
    return acc


def piecewise_three(x, *par):
    return _masked_sum(x, par, 3)


def piecewise_four(x, *par):
    return _masked_sum(x, par, 4)


def piecewise_five(x, *par):
    return _masked_sum(x, par, 5)


class dispersion_model(object):
    """dispersion.pyx:59-355"""

    def __init__(self):
        self._h = self._p = self._r = None
        self._mu_params = self._r_params = None
        self._metadata = ''

    def __reduce__(self):
        return (dispersion_model, (), {'mu_params': self.mu_params, 'r_params': self.r_params})

    def __setstate__(self, x):
        self.mu_params = x['mu_params']
        self.r_params = x['r_params']

    h = property(lambda s: s._h, lambda s, x: setattr(s, '_h', x))
    p = property(lambda s: s._p, lambda s, x: setattr(s, '_p', x))
    r = property(lambda s: s._r, lambda s, x: setattr(s, '_r', x))
    metadata = property(lambda s: s._metadata, lambda s, x: setattr(s, '_metadata', x))

    @property
    def mu_params(self):
        return self._mu_params

    @mu_params.setter
    def mu_params(self, x):
        self._mu_params = np.array(x, order='c')

    @property
    def r_params(self):
        return self._r_params

    @r_params.setter
    def r_params(self, x):
        self._r_params = np.array(x, order='c')

    def fit_mu(self, x):
        """dispersion.pyx:127-144"""
        res = piecewise_three(x, *np.asarray(self._mu_params, dtype=np.float64))
        return res if res > 0.0 else 0.1

    def fit_r(self, x):
        """dispersion.pyx:146-163 (raises ZeroDivisionError when the fit is exactly 0)"""
        v = piecewise_five(x, *np.asarray(self._r_params, dtype=np.float64))
        if v == 0.0:
            raise ZeroDivisionError("float division")
        res = 1.0 / v
        return res if res > 0.0 else 1e-6

    def __str__(self):
        raise NotImplementedError

    # ---- per-base NB values on the GPU ---------------------------------------------------
    def _slot(self, ctx):
        return ctx.dispersion_slot(self._mu_params, self._r_params)

    def _values(self, what, exp, obs, res=None, ctx=None):
        ctx = ctx or _lib.get_ctx()
        exp, obs = _lib.f64(exp), _lib.f64(obs)
        if exp.shape != obs.shape:
            raise ValueError("exp and obs differ in shape")
        out = res if res is not None else np.empty(exp.shape)
        buf = out if (isinstance(out, np.ndarray) and out.dtype == np.float64
                      and out.flags.c_contiguous) else np.empty(exp.shape)
        _lib.check(ctx.L.fpt_nb_values(ctx.h, what, self._slot(ctx), _lib.ptr(exp), _lib.ptr(obs),
                                       exp.size, _lib.ptr(buf)))
        if buf is not out:
            out[...] = buf
        return out

    def log_pmf_values(self, exp, obs):
        """dispersion.pyx:170-196"""
        return self._values(_lib.NB_LOGPMF, exp, obs)

    def pmf_values(self, exp, obs):
        """dispersion.pyx:199-225"""
        return self._values(_lib.NB_PMF, exp, obs)

    def log_pmf_values_0(self, exp, obs, res):
        """dispersion.pyx:228-258: writes into caller-owned `res` and returns it"""
        return self._values(_lib.NB_LOGPMF, exp, obs, res)

    def pmf_values_0(self, exp, obs, res):
        """dispersion.pyx:260-289"""
        return self._values(_lib.NB_PMF, exp, obs, res)

    def p_values(self, exp, obs):
        """dispersion.pyx:291-316"""
        return self._values(_lib.NB_CDF, exp, obs)

    def sample(self, x, times):
        """dispersion.pyx:318-355.  Draws come from numpy's global legacy RNG exactly as the
        reference draws them (np.random.negative_binomial per base, in order), so a seeded run
        reproduces the reference's samples; the p-values of the draws are evaluated on the GPU."""
        x = _lib.f64(x)
        n = x.shape[0]
        r = np.array([self.fit_r(v) for v in x])
        mu = np.array([self.fit_mu(v) for v in x])
        p = r / (r + mu)
        vals = np.zeros((n, times), dtype=np.int_)
        for i in range(n):
            vals[i, :] = np.random.negative_binomial(r[i], p[i], times)
        ctx = _lib.get_ctx()
        k = np.ascontiguousarray(vals, dtype=np.int64)
        # <int>vals[j]: C truncation of a long to int (dispersion.pyx:353)
        k32 = np.ascontiguousarray(k.astype(np.int32))
        pp = np.ascontiguousarray(np.repeat(p, times))
        rr = np.ascontiguousarray(np.repeat(r, times))
        pv = np.ones((n, times), dtype=np.float64)
        _lib.check(ctx.L.fpt_nb_scalar(ctx.h, _lib.NB_CDF, _lib.ptr(k32), _lib.ptr(pp), _lib.ptr(rr),
                                       k32.size, _lib.ptr(pv)))
        return vals, pv


def learn_dispersion_model(h, cutoff=250, trim=(2.5, 97.5)):
    """Dispersion model from the (expected, observed) histogram `h` (rows = expected count),
    following modeling/dispersion.pyx:357-469: a maximum-likelihood NB fit of every row with at
    least `cutoff` observations (after trimming `trim` percent off both ends), then continuous
    piecewise-linear fits of mu(x) (3 segments, forced through the first fitted row) and of
    1/r(x) (5 segments whose breakpoints are optimised, forced through row 1).

    The histogram comes from `FootprintScanner.histogram` (device) or the reference's loop
    (cli/learn_dm.py:276-287).  The piecewise fits use `modeling.piecewise` in place of pwlf."""
    from scipy import optimize

    from ..stats.distributions import nbinom
    from .piecewise import PiecewiseLinFit
    h = np.asarray(h)
    size = int(h.shape[0])
    p, r = np.zeros(size), np.zeros(size)
    values = np.arange(h.shape[1], dtype=np.float64)
    for i in range(size):
        counts = h[i, :].astype(np.int64)
        x = np.repeat(values, counts)  # the row unpacked, ascending
        if len(x) > 1e5:  # downsample to keep the root finding tractable (global numpy RNG)
            x = np.sort(np.random.choice(x, size=int(1e5)))
        if len(x) < cutoff:
            p[i] = r[i] = np.nan  # too few points: left out of the curve fits
            continue
        lower = int(np.floor(x.shape[0] * (trim[0] / 100.0)))
        upper = int(np.ceil(x.shape[0] * (trim[1] / 100.0)))
        core = x[lower:upper]
        m, v = np.mean(core), np.var(core)
        r0 = (m * m) / (v - m) if v != m else np.inf
        if not (r0 > 0.0) or not np.isfinite(r0):
            r0 = 10.0
        p[i], r[i] = nbinom.fit(core, p=r0 / (r0 + m), r=r0)
    with np.errstate(all="ignore"):
        mus = p * r / (1 - p)
    r[r > 200] = 200.0  # the reference's guard against runaway fits

    x = np.arange(size)
    ok = np.isfinite(mus)
    if ok.sum() < 2:
        raise ValueError("not enough rows with >= %d observations to fit a dispersion model" % cutoff)
    first_x, last_x = np.min(x[ok]), np.max(x[ok]) * 0.75

    fit_mu = PiecewiseLinFit(x[ok], mus[ok])
    fit_mu.fit_with_breaks_force_points(np.linspace(first_x, last_x, 4), [x[ok][0]], [mus[ok][0]])

    fit_r = PiecewiseLinFit(x[ok], 1.0 / r[ok])
    best = optimize.minimize(fit_r.fit_with_breaks_opt, [3.0, 7.0, 15.0, 25.0])
    breaks = np.zeros(6)
    breaks[0], breaks[-1] = first_x, last_x
    breaks[1:-1] = best.x
    fit_r.fit_with_breaks_force_points(breaks, [1], [1.0 / r[1]])

    model = dispersion_model()
    model.h, model.p, model.r = h, p, r
    model.mu_params = list(fit_mu.fit_breaks[1:]) + list(fit_mu.intercepts) + list(fit_mu.slopes)
    model.r_params = list(fit_r.fit_breaks[1:]) + list(fit_r.intercepts) + list(fit_r.slopes)
    return model


def base64encode(x):
    return [str(x.dtype), base64.b64encode(x), x.shape]


def base64decode(x):
    dtype = np.dtype(x[0])
    arr = np.frombuffer(base64.b64decode(x[1]), dtype)
    if len(x) > 2:
        return arr.reshape(x[2])
    return arr


def load_dispersion_model(filename):
    """dispersion.pyx:483-521 (stdlib json instead of simplejson)"""
    import json
    import urllib.request as request

    file = request.urlopen(filename) if filename.startswith('http') else open(filename, 'r')
    params = json.load(file)
    file.close()
    model = dispersion_model()
    model.mu_params = base64decode(params['mu_params'])
    model.r_params = base64decode(params['r_params'])
    if 'h' in params:
        model.h = base64decode(params['h'])
    if 'p' in params:
        model.p = base64decode(params['p'])
    if 'r' in params:
        model.r = base64decode(params['r'])
    if 'metadata' in params:
        model.metadata = params['metadata']
    return model


def write_dispersion_model(model, extra=None):
    """dispersion.pyx:523-549"""
    import json
    from datetime import datetime

    from .. import __version__

    def enc(a):
        a = np.asarray(a if a is not None else np.zeros(0), order='C')
        return [str(a.dtype), base64.b64encode(a).decode('ascii'), list(a.shape)]

    out = {'mu_params': enc(model.mu_params), 'r_params': enc(model.r_params), 'h': enc(model.h),
           'p': enc(model.p), 'r': enc(model.r),
           'version': "footprint_tools_amd %s" % __version__,
           'date': "on %s" % datetime.now().strftime('%Y-%m-%d %H:%M:%S'),
           'metadata': extra if extra else ""}
    return json.dumps(out, indent=4)
