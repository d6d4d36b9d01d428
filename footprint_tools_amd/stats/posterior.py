"""Posterior footprint probabilities: the call surface of footprint_tools/stats/posterior.py
(v1.3.7) -- the four functions, one reference call each (the windowed negative-binomial
log-likelihoods on the GPU, the two prior builders as numpy reductions over datasets) -- and the
batched form the drivers use: `posterior_batch` / `posterior_dev` run the whole sequence of
cli/post.py:98-124 for every interval and dataset of a batch in ONE kernel (fpt_posterior_dev of
include/fpt.h)."""
import ctypes as C

import numpy as np

from .. import _lib
from . import windowing


def compute_prior_weighted(fdr, w, cutoff=0.05, pseudocount=0.5):
    """Per-nucleotide occupancy prior (reference: posterior.py:12-42).

    fdr, w : (datasets, bases).  With k = #datasets calling a footprint (fdr <= cutoff) and
    n = #datasets in which the base lies in a hotspot, the prior is (n-k+c)/(n+2c); bases
    outside a dataset's hotspots get 1."""
    fdr = np.asarray(fdr)
    w = np.asarray(w)
    called = np.count_nonzero(fdr <= cutoff, axis=0)
    covered = np.sum(w, axis=0)
    unocc = covered - called + pseudocount
    occ = called + pseudocount
    row = unocc / (unocc + occ)
    return np.where(w == 0, 1.0, np.ones(fdr.shape) * row[np.newaxis, :])


def compute_delta_prior(obs, exp, fdr, beta_prior, cutoff=0.05):
    """Point estimate of the expected depletion at footprinted bases (posterior.py:45-90):
    precision-weighted mean over datasets of the Beta posterior mean of obs / max(exp, obs),
    using only datasets whose fdr passes `cutoff`; bases no dataset supports get 1."""
    import scipy.stats

    obs = np.asarray(obs, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    prior = np.asarray(beta_prior, dtype=np.float64)
    trials = np.maximum(exp, obs)
    mean, var = scipy.stats.beta.stats(obs + prior[:, 0:1], trials - obs + prior[:, 1:2], moments="mv")
    with np.errstate(all="ignore"):
        weight = 1.0 / np.sqrt(var)
        weight[np.asarray(fdr) > cutoff] = 0
        delta = (weight * mean).sum(axis=0) / weight.sum(axis=0)
    delta[np.isnan(delta)] = 1
    return delta


def log_likelihood(obs, exp, dm, delta=1, w=3):
    """Windowed NB log-likelihood per dataset and base (posterior.py:93-121): row i is
    windowing.sum(dm[i].log_pmf_values(exp[i] * delta, obs[i]), w), edges 1.0."""
    obs = np.asarray(obs, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    out = np.ones(obs.shape, order="c")
    for i, model in enumerate(dm[:obs.shape[0]] if hasattr(dm, "__getitem__") else dm):
        scaled = np.ascontiguousarray(exp[i] * delta)
        out[i] = windowing.sum(model.log_pmf_values(scaled, np.ascontiguousarray(obs[i])), w)
    return out


def posterior(prior, ll_on, ll_off):
    """log P(unoccupied | data) (posterior.py:124-149)."""
    with np.errstate(all="ignore"):
        occupied = np.log(1 - prior) + ll_on
        unoccupied = np.log(prior) + ll_off
        return unoccupied - np.logaddexp(occupied, unoccupied)


# ---- the batched form: cli/post.py:98-124 for a whole batch in one launch ----------------------
def pack_models(models):
    """(datasets, 24) array -- mu_params (9) then r_params (15) per dataset -- of dispersion models or
    (mu_params, r_params) pairs: what fpt_posterior_desc.models takes"""
    rows = []
    for m in models:
        mu, r = (m.mu_params, m.r_params) if hasattr(m, "mu_params") else m
        mu, r = _lib.f64(mu).ravel(), _lib.f64(r).ravel()
        if mu.size != 9 or r.size != 15:
            raise ValueError("mu_params needs 9 and r_params 15 values")
        rows.append(np.concatenate([mu, r]))
    return np.ascontiguousarray(np.stack(rows))


def posterior_dev(ctx, n_intervals, total_bases, n_datasets, dm_slot, betas, obs, exp, fdr, w, post_out,
                  interval_len=None, interval_off_dev=None, max_interval_len=0, fdr_cutoff=0.05,
                  half_win_width=3, pseudocount=0.5, prior_out=None, delta_out=None, ll_on_out=None,
                  ll_off_out=None, status_out=None, models=None):
    """Enqueue fpt_posterior_dev on device pointers (ints); does not synchronise.  Tracks are
    (n_datasets, total_bases) row-major, post_out (total_bases, n_datasets); dataset d uses betas[d]
    and row d of `models` (pack_models: any number of datasets, nothing kept on the context) or,
    without it, dispersion slot dm_slot + d (at most 64 datasets)."""
    d = _lib.PosteriorDesc()
    d.n_intervals, d.interval_len, d.interval_off = int(n_intervals), int(interval_len or 0), interval_off_dev
    d.total_bases, d.max_interval_len = int(total_bases), int(max_interval_len)
    d.n_datasets, d.dm_id, d.half_win_width = int(n_datasets), int(dm_slot), int(half_win_width)
    d.fdr_cutoff, d.pseudocount = float(fdr_cutoff), float(pseudocount)
    keep = _lib.f64(betas).reshape(int(n_datasets), 2)
    d.betas = keep.ctypes.data
    d.obs, d.exp, d.fdr, d.w, d.post_out = obs, exp, fdr, w, post_out
    d.prior_out, d.delta_out, d.ll_on_out, d.ll_off_out, d.status_out = prior_out, delta_out, ll_on_out, ll_off_out, status_out
    keep_m = None
    if models is not None:
        keep_m = _lib.f64(models).reshape(int(n_datasets), 24)
        d.models = keep_m.ctypes.data
    _lib.check(ctx.L.fpt_posterior_dev(ctx.h, C.byref(d)))


def posterior_batch(obs, exp, fdr, w, betas, dm, fdr_cutoff=0.05, half_win_width=3, interval_off=None,
                    pieces=False, ctx=None):
    """-log P(unoccupied | data), clipped at 0, for every base and dataset of a batch.

    obs, exp, fdr, w : (datasets, bases) host arrays -- what cli/post.py's `_load_data` builds for
    one interval, for many intervals back to back (`interval_off`: their offsets; None = the
    arrays are one interval).  betas: (datasets, 2); dm: one dispersion model per dataset.
    Returns the (bases, datasets) array whose slice per interval is the reference's `stats`
    record; with pieces=True also a dict of prior, delta, ll_on, ll_off as the reference's
    functions return them.  Raises ZeroDivisionError where dm.log_pmf_values would."""
    from ..scan import DeviceArray
    ctx = ctx or _lib.get_ctx()
    obs, exp, fdr, w = (_lib.f64(a) for a in (obs, exp, fdr, w))
    if obs.ndim != 2 or exp.shape != obs.shape or fdr.shape != obs.shape or w.shape != obs.shape:
        raise ValueError("obs, exp, fdr and w must be (datasets, bases) arrays of one shape")
    D, total = obs.shape
    models = list(dm)[:D]
    if len(models) != D:
        raise ValueError("one dispersion model per dataset")
    if interval_off is None:
        off = np.array([0, total], dtype=np.int64)
    else:
        off = np.ascontiguousarray(interval_off, dtype=np.int64)
        if off[0] != 0 or off[-1] != total or np.any(np.diff(off) < 0):
            raise ValueError("interval_off does not partition the tracks")
    n_iv = off.size - 1
    if total == 0:
        empty = np.zeros((0, D))
        return (empty, dict(prior=np.zeros((D, 0)), delta=np.zeros(0), ll_on=np.zeros((D, 0)), ll_off=np.zeros((D, 0)))) if pieces else empty
    packed = pack_models(models)  # handed over with the call: any number of datasets, no slot of the context touched
    slot = 0
    bufs = []
    try:
        def dev(a):
            b = DeviceArray(ctx, max(a.nbytes, 16)).upload(a)
            bufs.append(b)
            return b
        d_in = dev(np.concatenate([obs.ravel(), exp.ravel(), fdr.ravel(), w.ravel()]))
        d_off = dev(off)
        d_st = dev(np.zeros(max(n_iv, 1), np.int32))
        n = D * total
        d_out = DeviceArray(ctx, (n * (4 if pieces else 1) + (total if pieces else 0)) * 8)
        bufs.append(d_out)
        base = d_out.ptr
        posterior_dev(ctx, n_iv, total, D, slot, betas, d_in.ptr, d_in.ptr + n * 8, d_in.ptr + 2 * n * 8,
                      d_in.ptr + 3 * n * 8, base, interval_off_dev=d_off.ptr,
                      max_interval_len=int(np.diff(off).max()), fdr_cutoff=fdr_cutoff,
                      half_win_width=half_win_width, status_out=d_st.ptr,
                      prior_out=base + n * 8 if pieces else None, ll_on_out=base + 2 * n * 8 if pieces else None,
                      ll_off_out=base + 3 * n * 8 if pieces else None, delta_out=base + 4 * n * 8 if pieces else None,
                      models=packed)
        ctx.synchronize()
        if d_st.download(np.int32, n_iv).any():
            raise ZeroDivisionError("float division")  # dispersion.pyx:160-161 through log_pmf_values
        post = d_out.download(np.float64, n).reshape(total, D)
        if not pieces:
            return post
        rest = d_out.download(np.float64, 3 * n + total, n * 8)
        return post, dict(prior=rest[:n].reshape(D, total), ll_on=rest[n:2 * n].reshape(D, total),
                          ll_off=rest[2 * n:3 * n].reshape(D, total), delta=rest[3 * n:])
    finally:
        for b in bufs:
            b.free()
