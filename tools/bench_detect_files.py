"""End-to-end rate of the detect driver on files (diagnostic): a synthetic BAM + FASTA, 20,000 intervals,
`detect.deviation_stats` with the readers of this package -- the batch handed over on the device --
and, on a subset, through the per-interval reader interface the reference uses.  One line per mode."""
import itertools
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import cutcounts, detect  # noqa: E402
from footprint_tools_amd.fasta import FastaFile  # noqa: E402
from footprint_tools_amd.modeling import bias, dispersion  # noqa: E402
from tests.bamwriter import write_bam  # noqa: E402

n_reads, n_iv, glen = int(float(os.environ.get("N_READS", "2e6"))), int(os.environ.get("N_IV", "20000")), 6000000
rs = np.random.RandomState(1)
tmp = tempfile.mkdtemp()
refs = [("chr1", glen)]
pos = np.sort(rs.randint(0, glen - 100, n_reads))
flags = rs.choice([0, 16], n_reads)
t0 = time.perf_counter()
write_bam(os.path.join(tmp, "r.bam"), refs, [dict(ref=0, pos=int(p), cigar="36M", flag=int(f), mapq=30) for p, f in zip(pos, flags)],
          block_bytes=60000, index=True)
seq = "".join(rs.choice(list("ACGT"), glen))
with open(os.path.join(tmp, "g.fa"), "w") as f:
    f.write(">chr1\n" + "\n".join(seq[a:a + 60] for a in range(0, glen, 60)) + "\n")
print("files written in %.1f s (%d alignments)" % (time.perf_counter() - t0, n_reads))
g, lat = np.load("tests/golden/kmer_probs.npz"), np.load("tests/golden/nb_lattice.npz")
bm = bias.bias_model()
for j, kk in enumerate(itertools.product("ACGT", repeat=6)):
    bm["".join(kk)] = float(g["table"][j])
dm = dispersion.dispersion_model()
dm.mu_params, dm.r_params = lat["mu_A"], lat["r_A"]


class Interval(object):
    def __init__(self, c, s, e):
        self.chrom, self.start, self.end = c, s, e

    def widen(self, w):
        return Interval(self.chrom, self.start - w, self.end + w)


starts = np.sort(rs.randint(1000, glen - 2000, n_iv))
lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 1500).astype(int)
ivs = [Interval("chr1", int(s), int(s + l)) for s, l in zip(starts, lens)]
t0 = time.perf_counter()
bf, fa = cutcounts.bamfile(os.path.join(tmp, "r.bam")), FastaFile(os.path.join(tmp, "g.fa"))
print("BAM read: %.2f s" % (time.perf_counter() - t0))
# the alignments of a tenth of the intervals only, through the BAI index
t0 = time.perf_counter()
sub10 = ivs[:len(ivs) // 10]
bf10 = cutcounts.bamfile(os.path.join(tmp, "r.bam"), regions=[(iv.chrom, iv.start, iv.end) for iv in sub10])
print("BAM read through the index, regions of %d intervals: %.2f s (%d of %d alignments)"
      % (len(sub10), time.perf_counter() - t0, bf10.n_reads, bf.n_reads))
for mode in ("device", "per-interval"):
    sub = ivs if mode == "device" else ivs[:1000]
    rf = bf if mode == "device" else type("R", (), {"__getitem__": lambda self, iv: bf[iv]})()
    ff = fa if mode == "device" else type("F", (), {"fetch": lambda self, c, s, e: fa.fetch(c, s, e)})()
    ds = detect.deviation_stats(sub, rf, ff, bm, dm, fdr_shuffle_n=50, seed=1, batch_size=int(os.environ.get("BATCH", "4096")))
    ds.compute(range(min(64, len(sub))))  # warm-up
    prof = None
    if os.environ.get("PROFILE") and mode == "device":
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    n = sum(sum(s.shape[0] for s in b["stats"]) for b in ds.batch_iter())
    dt = time.perf_counter() - t0
    if prof:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(22)
    print("%-12s readers: %d intervals, %d bases in %.2f s -> %.3g bases/s" % (mode, len(sub), n, dt, n / dt))
