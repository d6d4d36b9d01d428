"""End-to-end rate of the posterior caller from tracks (diagnostic): one statistics track (made by
detect.write_track from a synthetic BAM + FASTA) read as N_DS datasets through tabix.TabixFile.fetch_batch
-> posterior_stats.batch (one fpt_posterior_dev launch per batch).  One line per stage."""
import itertools
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import cutcounts, detect, post  # noqa: E402
from footprint_tools_amd.fasta import FastaFile  # noqa: E402
from footprint_tools_amd.modeling import bias, dispersion  # noqa: E402
from tests.bamwriter import write_bam  # noqa: E402

n_reads, n_iv, glen, n_ds = int(float(os.environ.get("N_READS", "2e6"))), int(os.environ.get("N_IV", "50000")), 12000000, int(os.environ.get("N_DS", "8"))
rs = np.random.RandomState(1)
tmp = tempfile.mkdtemp()
pos = np.sort(rs.randint(0, glen - 100, n_reads))
flags = rs.choice([0, 16], n_reads)
write_bam(os.path.join(tmp, "r.bam"), [("chr1", glen)],
          [dict(ref=0, pos=int(p), cigar="36M", flag=int(f), mapq=30) for p, f in zip(pos, flags)], block_bytes=60000)
seq = "".join(rs.choice(list("ACGT"), glen))
with open(os.path.join(tmp, "g.fa"), "w") as f:
    f.write(">chr1\n" + "\n".join(seq[a:a + 60] for a in range(0, glen, 60)) + "\n")
g, lat = np.load("tests/golden/kmer_probs.npz"), np.load("tests/golden/nb_lattice.npz")
bm = bias.bias_model()
for j, kk in enumerate(itertools.product("ACGT", repeat=6)):
    bm["".join(kk)] = float(g["table"][j])
dm = dispersion.dispersion_model()
dm.mu_params, dm.r_params = lat["mu_A"], lat["r_A"]
dm_file = os.path.join(tmp, "dm.json")
open(dm_file, "w").write(dispersion.write_dispersion_model(dm))


class Interval(object):
    def __init__(self, c, s, e):
        self.chrom, self.start, self.end = c, s, e

    def __len__(self):
        return self.end - self.start

    def widen(self, w):
        return Interval(self.chrom, self.start - w, self.end + w)


lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 1500).astype(int)
starts = 1000 + np.cumsum(lens + rs.randint(10, 60, n_iv)) - lens
ivs = [Interval("chr1", int(s), int(s + l)) for s, l in zip(starts, lens)]
bf, fa = cutcounts.bamfile(os.path.join(tmp, "r.bam")), FastaFile(os.path.join(tmp, "g.fa"))
ds = detect.deviation_stats(ivs, bf, fa, bm, dm, fdr_shuffle_n=50, seed=1, batch_size=8192)
track = os.path.join(tmp, "stats.bed.gz")
t0 = time.perf_counter()
n = detect.write_track(ds, track)
print("track written: %d bases in %.2f s" % (n, time.perf_counter() - t0))
samples = [dict(tabix_file=track, dm_file=dm_file, beta_a=1.0 + 0.1 * d, beta_b=20.0) for d in range(n_ds)]
ps = post.posterior_stats(ivs, samples, 0.05)
ps.batch(range(64))
bs = int(os.environ.get("BATCH", "8192"))
t0 = time.perf_counter()
t_load = 0.0
for a in range(0, n_iv, bs):
    t1 = time.perf_counter()
    ps.load_batch(ivs[a:a + bs])
    t_load += time.perf_counter() - t1
print("load_batch alone:     %.2f s -> %.3g bases/s (%d datasets)" % (t_load, n / t_load, n_ds))
t0 = time.perf_counter()
m = 0
for a in range(0, n_iv, bs):
    m += sum(r["stats"].shape[0] for r in ps.batch(range(a, min(a + bs, n_iv))))
dt = time.perf_counter() - t0
print("posterior_stats.batch: %d bases x %d datasets in %.2f s -> %.3g bases/s" % (m, n_ds, dt, m / dt))
