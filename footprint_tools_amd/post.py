"""Posterior footprint probabilities over many datasets: the per-interval driver of
cli/post.py:40-124 (`posterior_stats`) on this package's modules.

For one interval, every dataset's per-nucleotide track (`ftd detect` output) gives obs, exp and
fdr rows (`_load_data`, post.py:57-87: column 3 = exp, 4 = obs, 7 = fdr, w = 1 where the dataset
has a row); the priors are numpy reductions over datasets (stats/posterior.py:12-90) and the
windowed NB log-likelihoods run on the GPU (stats.posterior.log_likelihood).  The record is
`{"interval", "stats"}` with stats = -log P(unoccupied) clipped at 0, bases x datasets."""
import numpy as np

from .modeling import dispersion
from .stats import posterior
from .tabix import TabixFile


class _interval(object):
    def __init__(self, chrom, start, end):
        self.chrom, self.start, self.end = str(chrom), int(start), int(end)

    def __len__(self):
        return self.end - self.start

    def __str__(self):
        return "%s:%d-%d" % (self.chrom, self.start, self.end)


class posterior_stats(object):
    def __init__(self, intervals, samples_data, fdr_cutoff):
        """intervals: sequence of (chrom, start, end) or objects with those attributes (the
        reference reads a BED file); samples_data: one record per dataset with `tabix_file`,
        `dm_file`, `beta_a`, `beta_b` (a pandas DataFrame with those columns, as in the reference,
        or a list of dicts)."""
        self.intervals = [iv if hasattr(iv, "chrom") else _interval(*iv[:3]) for iv in intervals]
        rows = samples_data.to_dict("records") if hasattr(samples_data, "to_dict") else list(samples_data)
        self.samples_data = rows
        self.fdr_cutoff = fdr_cutoff
        self.tabix_files = []  # opened on the first __getitem__, as in the reference
        self.disp_models = [dispersion.load_dispersion_model(r["dm_file"]) for r in rows]
        self.betas = np.array([[r["beta_a"], r["beta_b"]] for r in rows], dtype=np.float64)

    def _open_tabix_files(self):
        self.tabix_files = [TabixFile(r["tabix_file"]) for r in self.samples_data]

    def _load_data(self, interval):
        """post.py:57-87"""
        n, m = len(self.tabix_files), len(interval)
        obs, exp = np.zeros((n, m)), np.zeros((n, m))
        fdr, w = np.ones((n, m)), np.zeros((n, m))
        for i, tbf in enumerate(self.tabix_files):
            try:
                pos, vals = tbf.fetch_columns(interval.chrom, interval.start, interval.end)
                j = pos - interval.start
                exp[i, j] = vals[:, 2]   # file column 3
                obs[i, j] = vals[:, 3]   # 4
                fdr[i, j] = vals[:, 6]   # 7
                w[i, j] = 1.0
            except Exception:  # the reference logs and carries on with what it has (post.py:84-85)
                pass
        return obs, exp, fdr, w

    def cleanup(self):
        for tbf in self.tabix_files:
            tbf.close()
        self.tabix_files = []

    def __len__(self):
        return len(self.intervals)

    def __getitem__(self, index):
        """post.py:98-124"""
        if len(self.tabix_files) == 0:
            self._open_tabix_files()
        interval = self.intervals[index]
        obs, exp, fdr, w = self._load_data(interval)
        prior = posterior.compute_prior_weighted(fdr, w, cutoff=self.fdr_cutoff)
        delta = posterior.compute_delta_prior(obs, exp, fdr, self.betas, cutoff=self.fdr_cutoff)
        ll_on = posterior.log_likelihood(obs, exp, self.disp_models, delta=delta, w=3)
        ll_off = posterior.log_likelihood(obs, exp, self.disp_models, w=3)
        post = -posterior.posterior(prior, ll_on, ll_off)
        post[post <= 0] = 0.0
        return {"interval": interval, "stats": post.T}
