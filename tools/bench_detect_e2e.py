"""End-to-end rate of `detect` from files to an indexed track (diagnostic): synthetic BAM + FASTA ->
detect.deviation_stats (device readers) -> write_stats_to_output -> tabix.TrackWriter (BGZF + .tbi).
One line per stage arrangement."""
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
from footprint_tools_amd.tabix import TrackWriter  # noqa: E402
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


# non-overlapping sorted intervals (a track has to be sorted)
lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 1500).astype(int)
gaps = rs.randint(10, 60, n_iv)
starts = 1000 + np.cumsum(lens + gaps) - lens
assert starts[-1] + lens[-1] < glen - 2000
ivs = [Interval("chr1", int(s), int(s + l)) for s, l in zip(starts, lens)]
bf, fa = cutcounts.bamfile(os.path.join(tmp, "r.bam")), FastaFile(os.path.join(tmp, "g.fa"))
ds = detect.deviation_stats(ivs, bf, fa, bm, dm, fdr_shuffle_n=50, seed=1, batch_size=int(os.environ.get("BATCH", "8192")))
ds.compute(range(64))
t0 = time.perf_counter()
n = sum(sum(s.shape[0] for s in b["stats"]) for b in ds.batch_iter())
t_stats = time.perf_counter() - t0
print("statistics only:        %d bases in %.2f s -> %.3g bases/s" % (n, t_stats, n / t_stats))


class Sink(object):
    n = 0

    def write(self, text):
        self.n += len(text)


t0 = time.perf_counter()
sink = Sink()
for b in ds.batch_iter():
    for iv, st in zip(b["interval"], b["stats"]):
        detect.write_stats_to_output(iv, st, file=sink)
t_text = time.perf_counter() - t0
print("... + text (discarded): %.2f s -> %.3g bases/s (%d bytes)" % (t_text, n / t_text, sink.n))
path = os.path.join(tmp, "out.bed.gz")
t0 = time.perf_counter()
with TrackWriter(path) as w:
    for b in ds.batch_iter():
        for iv, st in zip(b["interval"], b["stats"]):
            detect.write_stats_to_output(iv, st, file=w)
t_track = time.perf_counter() - t0
print("... + indexed track:    %.2f s -> %.3g bases/s (%d bytes + %d of index)"
      % (t_track, n / t_track, os.path.getsize(path), os.path.getsize(path + ".tbi")))
if hasattr(detect, "write_batch_to_output"):
    t0 = time.perf_counter()
    sink = Sink()
    for b in ds.batch_iter():
        detect.write_batch_to_output(b, file=sink)
    t1 = time.perf_counter() - t0
    print("batch -> text (disc.):  %.2f s -> %.3g bases/s (%d bytes)" % (t1, n / t1, sink.n))
    t0 = time.perf_counter()
    with open(os.path.join(tmp, "out.bedgraph"), "w") as f:
        for b in ds.batch_iter():
            detect.write_batch_to_output(b, file=f)
    t1 = time.perf_counter() - t0
    print("batch -> text file:     %.2f s -> %.3g bases/s" % (t1, n / t1))
    path2 = os.path.join(tmp, "out2.bed.gz")
    t0 = time.perf_counter()
    with TrackWriter(path2) as w:
        for b in ds.batch_iter():
            detect.write_batch_to_output(b, file=w)
    t2 = time.perf_counter() - t0
    same = open(path, "rb").read() == open(path2, "rb").read()
    print("batch -> track:         %.2f s -> %.3g bases/s (same bytes: %s)" % (t2, n / t2, same))
if hasattr(detect, "write_track"):
    path3 = os.path.join(tmp, "out3.bed.gz")
    t0 = time.perf_counter()
    detect.write_track(ds, path3)
    t3 = time.perf_counter() - t0
    same = open(path, "rb").read() == open(path3, "rb").read()
    print("write_track (overlap):  %.2f s -> %.3g bases/s (same bytes: %s)" % (t3, n / t3, same))
    t0 = time.perf_counter()
    detect.write_track(ds, path3, level=1)
    t3 = time.perf_counter() - t0
    print("write_track, level 1:   %.2f s -> %.3g bases/s (%d bytes)" % (t3, n / t3, os.path.getsize(path3)))
