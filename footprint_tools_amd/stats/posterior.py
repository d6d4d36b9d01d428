"""Posterior footprint probabilities: the call surface of footprint_tools/stats/posterior.py
(v1.3.7).  The windowed negative-binomial log-likelihoods run on the GPU (log-pmf kernel +
window-sum kernel); the two prior builders are small numpy reductions over datasets, as in the
reference."""
import numpy as np

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
