#!/usr/bin/env python3
"""End-to-end walk through the path on synthetic data (needs an MI355X):

    learn a dispersion model (`ftd learn_dm`)  ->  per-base statistics with empirical FDR
    (`ftd detect`)  ->  footprint calls (BED)  ->  the same columns the reference writes.

Everything heavy runs on the GPU through libfpt_hip.so; the "genome", reads and intervals are
made up here so that the script has no inputs.  Usage: python examples/synthetic_detect.py
"""
import io
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from footprint_tools_amd import detect, learn  # noqa: E402
from footprint_tools_amd.modeling import bias, dispersion  # noqa: E402


class Interval(object):  # chrom / start / end / widen(n), like genome_tools.genomic_interval
    def __init__(self, chrom, start, end):
        self.chrom, self.start, self.end = chrom, start, end

    def widen(self, w):
        return Interval(self.chrom, self.start - w, self.end + w)


def main():
    rs = np.random.RandomState(7)
    genome_len = 400000
    seq = "".join(rs.choice(list("ACGT"), genome_len))
    # a 6-mer bias model with propensities spread over two orders of magnitude
    bm = bias.bias_model()
    for kmer in itertools.product("ACGT", repeat=6):
        bm["".join(kmer)] = float(np.exp(rs.normal(-4.0, 1.0)))
    # cut counts follow the model the path assumes: rate = accessibility x protection x 6-mer
    # propensity of the strand (predict.pyx:150-153 alignment), negative-binomial noise
    access = 6.0 + 4.0 * np.sin(np.arange(genome_len) / 3000.0) ** 2
    protected = np.zeros(genome_len, bool)
    for s in rs.randint(1000, genome_len - 1000, 400):
        protected[s:s + rs.randint(8, 20)] = True
    fwd, rev = bm.probs_both(seq)
    prop_p, prop_m = np.ones(genome_len), np.ones(genome_len)
    prop_p[3:genome_len - 3], prop_m[3:genome_len - 3] = fwd / fwd.mean(), rev / rev.mean()
    lam = np.where(protected, 0.1, 1.0) * access
    r_true = 6.0
    plus = rs.negative_binomial(r_true, r_true / (r_true + lam * prop_p)).astype(np.float64)
    minus = rs.negative_binomial(r_true, r_true / (r_true + lam * prop_m)).astype(np.float64)

    class Reads(object):
        def __getitem__(self, iv):
            return {"+": plus[iv.start:iv.end], "-": minus[iv.start:iv.end]}

    class Fasta(object):
        def fetch(self, chrom, s, e):
            return seq[s:e]

    starts = np.arange(2000, genome_len - 3000, 1500)
    intervals = [Interval("chrS", int(s), int(s) + int(rs.randint(300, 1200))) for s in starts]

    # 1. learn_dm: histogram of (expected, observed) on the device, model fit on the host.
    # learn.learn_dm(...) is the reference's flow (no smoothing, 2.5 % trimming); here the
    # histogram is taken with the smoothing `detect` uses and fitted without trimming, which
    # recovers the simulated dispersion on clean NB data.
    ds_learn = learn.expected_counts(intervals, Reads(), Fasta(), bm, half_win_width=5,
                                     smoothing_half_win_width=50, smoothing_clip=0.01)
    dm = dispersion.learn_dispersion_model(ds_learn.histogram(), trim=(0, 100))
    print("dispersion model: mu(10) = %.2f  r(10) = %.2f" % (dm.fit_mu(10.0), dm.fit_r(10.0)))
    model_json = dispersion.write_dispersion_model(dm)

    # 2. detect: expected / observed / -log p / -log window p / empirical FDR per base
    ds = detect.deviation_stats(intervals, Reads(), Fasta(), bm, dm, half_win_width=5,
                                smoothing_half_win_width=50, smoothing_clip=0.01, fdr_shuffle_n=100, seed=1)
    bedgraph, bed = io.StringIO(), io.StringIO()
    detect.write_output_header(["exp", "obs", "neglog_pval", "neglog_winpval", "fdr"], file=bedgraph,
                               include_name=False)
    n_bases = n_calls = hits = 0
    for batch in ds.batch_iter(batch_size=128):
        for iv, stats in zip(batch["interval"], batch["stats"]):
            detect.write_stats_to_output(iv, stats, file=bedgraph)
            detect.write_segments_to_output(iv, stats[:, -1], 0.001, file=bed, decreasing=True)
            n_bases += stats.shape[0]
    for line in bed.getvalue().splitlines():
        _, s, e = line.split("\t")[:3]
        n_calls += 1
        hits += bool(protected[int(s):int(e)].any())
    print("%d intervals, %d bases; %d footprints called at empirical FDR <= 0.001, %d overlap a planted site"
          % (len(intervals), n_bases, n_calls, hits))
    print("dispersion model JSON: %d bytes; bedGraph: %d lines" % (len(model_json), bedgraph.getvalue().count("\n")))

    # 3. the statistics as a bgzip-compressed, tabix-indexed track (what `bgzip` + `tabix -p bed` make of
    #    the bedGraph in the reference's workflow), read back by region the way the posterior caller does
    import os
    import tempfile
    from footprint_tools_amd.tabix import TabixFile, TrackWriter
    path = os.path.join(tempfile.mkdtemp(), "stats.bed.gz")
    with TrackWriter(path) as w:
        w.write(bedgraph.getvalue())
    # (a run that only wants the track skips the text: `detect.write_track(ds, path)` formats and
    #  compresses every batch inside the library while the next one is on the GPU)
    tb = TabixFile(path)
    iv = intervals[len(intervals) // 2]
    rows = list(tb.fetch(iv.chrom, iv.start, iv.end))
    print("track %s (%d bytes + %d of index): %d rows of %s:%d-%d read back through the index"
          % (os.path.basename(path), os.path.getsize(path), os.path.getsize(path + ".tbi"), len(rows), iv.chrom,
             iv.start, iv.end))


if __name__ == "__main__":
    main()
