"""Posterior footprint probabilities over many datasets: the driver of cli/post.py:40-124
(`posterior_stats`) in batched form.

Every dataset's per-nucleotide track (`ftd detect` output) gives obs, exp and fdr rows over an
interval (cli/post.py:57-87: file column 3 = exp, 4 = obs, 7 = fdr, w = 1 where the dataset has a
row; 0 / 0 / 1 / 0 elsewhere).  `load_batch` builds those rows for any number of intervals back to
back, and ONE kernel launch (stats.posterior.posterior_batch -> fpt_posterior_dev) turns them into
the records: priors over datasets, windowed NB log-likelihoods with and without the expected
depletion, log-sum-exp, -log P(unoccupied) clipped at 0.  `ps[i]` is the reference's per-interval
record `{"interval", "stats"}` (a batch of one); `ps.batch(indices)` is how the drivers call it."""
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
    def __init__(self, intervals, samples_data, fdr_cutoff, ctx=None):
        """intervals: sequence of (chrom, start, end) or objects with those attributes (the
        reference reads a BED file); samples_data: one record per dataset with `tabix_file`,
        `dm_file`, `beta_a`, `beta_b` (a pandas DataFrame with those columns, as in the reference,
        or a list of dicts)."""
        self.intervals = [iv if hasattr(iv, "chrom") else _interval(*iv[:3]) for iv in intervals]
        rows = samples_data.to_dict("records") if hasattr(samples_data, "to_dict") else list(samples_data)
        self.samples_data = rows
        self.fdr_cutoff = fdr_cutoff
        self.ctx = ctx
        self.tabix_files = []  # opened on first use, as in the reference
        self.disp_models = [dispersion.load_dispersion_model(r["dm_file"]) for r in rows]
        self.betas = np.array([[r["beta_a"], r["beta_b"]] for r in rows], dtype=np.float64)

    def _open_tabix_files(self):
        self.tabix_files = [TabixFile(r["tabix_file"]) for r in self.samples_data]

    def load_batch(self, intervals):
        """The (datasets, bases) arrays obs, exp, fdr, w of cli/post.py:57-87 for `intervals` back
        to back, and the offsets of the intervals in them.  One batched, threaded fetch per dataset
        (tabix.TabixFile.fetch_batch): column 3 = exp, 4 = obs, 7 = fdr, w = 1 where the track has a row."""
        if len(self.tabix_files) == 0:
            self._open_tabix_files()
        off = np.concatenate([[0], np.cumsum([len(iv) for iv in intervals])]).astype(np.int64)
        n, total = len(self.tabix_files), int(off[-1])
        obs, exp = np.zeros((n, total)), np.zeros((n, total))
        fdr, w = np.ones((n, total)), np.zeros((n, total))
        chroms = [iv.chrom for iv in intervals]
        starts = np.array([iv.start for iv in intervals], dtype=np.int64)
        ends = np.array([iv.end for iv in intervals], dtype=np.int64)
        for i, tbf in enumerate(self.tabix_files):
            try:
                tbf.fetch_batch(chroms, starts, ends, [3, 4, 7], out_off=off, out=[exp[i], obs[i], fdr[i]], present=w[i])
            except Exception:  # the reference logs and carries on with what it has (post.py:84-85)
                continue
        return obs, exp, fdr, w, off

    def _load_data(self, interval):
        """post.py:57-87 for one interval"""
        return self.load_batch([interval])[:4]

    def cleanup(self):
        for tbf in self.tabix_files:
            tbf.close()
        self.tabix_files = []

    def __len__(self):
        return len(self.intervals)

    def batch(self, indices):
        """Records of many intervals from one launch (the order of `indices`)."""
        ivs = [self.intervals[i] for i in indices]
        obs, exp, fdr, w, off = self.load_batch(ivs)
        stats = posterior.posterior_batch(obs, exp, fdr, w, self.betas, self.disp_models, fdr_cutoff=self.fdr_cutoff,
                                          half_win_width=3, interval_off=off, ctx=self.ctx)
        return [{"interval": iv, "stats": stats[a:b]} for iv, a, b in zip(ivs, off[:-1], off[1:])]

    def __getitem__(self, index):
        return self.batch([index])[0]

    def batch_iter(self, batch_size=4096):
        """{"interval": [...], "stats": [...]} per step, like the reference's loader (cli/post.py:160-175),
        plus the step's rows as one matrix ("table", "row_off") for detect.write_batch_to_output"""
        for a in range(0, len(self.intervals), int(batch_size)):
            ivs = self.intervals[a:a + int(batch_size)]
            obs, exp, fdr, w, off = self.load_batch(ivs)
            table = posterior.posterior_batch(obs, exp, fdr, w, self.betas, self.disp_models, fdr_cutoff=self.fdr_cutoff,
                                              half_win_width=3, interval_off=off, ctx=self.ctx)
            from .detect import _row_blocks
            yield {"interval": ivs, "stats": _row_blocks(table, off), "table": table, "row_off": off}
