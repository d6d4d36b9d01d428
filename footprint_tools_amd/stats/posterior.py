"""Posterior footprint probabilities.  Mirrors footprint_tools/stats/posterior.py (v1.3.7);
the windowed NB log-likelihoods run on the GPU."""
import numpy as np

from . import windowing


def compute_prior_weighted(fdr, w, cutoff=0.05, pseudocount=0.5):
    """posterior.py:12-42"""
    k = np.sum(fdr <= cutoff, axis=0)
    n = np.sum(w, axis=0)
    a = n - k + pseudocount
    b = k + pseudocount
    pr = a / (a + b)
    res = np.ones(fdr.shape)
    res *= pr[np.newaxis, :]
    res[w == 0] = 1
    return res


def compute_delta_prior(obs, exp, fdr, beta_prior, cutoff=0.05):
    """posterior.py:45-90"""
    import scipy.stats

    (n, w) = obs.shape
    mus = np.ones((n, w))
    ws = np.ones((n, w))
    for i in range(n):
        k = obs[i, :]
        nn = np.max(np.vstack([exp[i, :], obs[i, :]]), axis=0)
        mu, v = scipy.stats.beta.stats(k + beta_prior[i][0], nn - k + beta_prior[i][1], loc=0,
                                       scale=1, moments="mv")
        mus[i, :] = mu
        ws[i, :] = 1 / np.sqrt(v)
    ws[fdr > cutoff] = 0
    with np.errstate(all="ignore"):
        delta = np.sum(ws * mus, axis=0) / np.sum(ws, axis=0)
    delta[np.isnan(delta)] = 1
    return delta


def log_likelihood(obs, exp, dm, delta=1, w=3):
    """posterior.py:93-121"""
    res = np.ones((obs.shape[0], obs.shape[1]), order="c")
    n = obs.shape[0]
    for i in range(n):
        res[i, :] = windowing.sum(dm[i].log_pmf_values(exp[i, :] * delta, obs[i, :]), w)
    return res


def posterior(prior, ll_on, ll_off):
    """posterior.py:124-149"""
    with np.errstate(all="ignore"):
        prior_on = np.log(1 - prior)
        prior_off = np.log(prior)
        p_off = prior_off + ll_off
        p_on = prior_on + ll_on
        denom = np.logaddexp(p_on, p_off)
        return p_off - denom
