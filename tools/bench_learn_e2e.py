"""End-to-end rate of the learn_dm flow from files (diagnostic): synthetic BAM + FASTA -> expected_counts.histogram
(the (exp, obs) histogram on the device) -> dispersion.learn_dispersion_model (host fit).  One line per stage."""
import itertools
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import cutcounts, learn  # noqa: E402
from footprint_tools_amd.fasta import FastaFile  # noqa: E402
from footprint_tools_amd.modeling import bias, dispersion  # noqa: E402
from tests.bamwriter import write_bam  # noqa: E402

n_reads, n_iv, glen = int(float(os.environ.get("N_READS", "2e6"))), int(os.environ.get("N_IV", "50000")), 12000000
rs = np.random.RandomState(1)
tmp = tempfile.mkdtemp()
pos = np.sort(rs.randint(0, glen - 100, n_reads))
flags = rs.choice([0, 16], n_reads)
write_bam(os.path.join(tmp, "r.bam"), [("chr1", glen)],
          [dict(ref=0, pos=int(p), cigar="36M", flag=int(f), mapq=30) for p, f in zip(pos, flags)], block_bytes=60000)
seq = "".join(rs.choice(list("ACGT"), glen))
with open(os.path.join(tmp, "g.fa"), "w") as f:
    f.write(">chr1\n" + "\n".join(seq[a:a + 60] for a in range(0, glen, 60)) + "\n")
g = np.load("tests/golden/kmer_probs.npz")
bm = bias.bias_model()
for j, kk in enumerate(itertools.product("ACGT", repeat=6)):
    bm["".join(kk)] = float(g["table"][j])


class Interval(object):
    def __init__(self, c, s, e):
        self.chrom, self.start, self.end = c, s, e

    def widen(self, w):
        return Interval(self.chrom, self.start - w, self.end + w)


lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 1500).astype(int)
starts = 1000 + np.cumsum(lens + rs.randint(10, 60, n_iv)) - lens
ivs = [Interval("chr1", int(s), int(s + l)) for s, l in zip(starts, lens)]
n = int(lens.sum())
bf, fa = cutcounts.bamfile(os.path.join(tmp, "r.bam")), FastaFile(os.path.join(tmp, "g.fa"))
ec = learn.expected_counts(ivs, bf, fa, bm, half_win_width=5, batch_size=int(os.environ.get("BATCH", "8192")))
ec.compute(range(64))
t0 = time.perf_counter()
h = ec.histogram()
t1 = time.perf_counter() - t0
print("histogram: %d bases in %.2f s -> %.3g bases/s (%d counted)" % (n, t1, n / t1, int(h.sum())))
t0 = time.perf_counter()
dm = dispersion.learn_dispersion_model(h)
t2 = time.perf_counter() - t0
print("model fit: %.2f s; mu(5) = %.3f r(5) = %.3f" % (t2, dm.fit_mu(5.0), dm.fit_r(5.0)))
