"""Timing of the device cut-count ingestion (fpt_cut_counts_dev) on synthetic alignments: reads with
uniform random starts over a set of intervals shaped like BASELINE config 4 (lognormal lengths,
mean ~170 bp), the kernel alone (reads and interval table already on the device).  Diagnostic,
prints one line."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import _lib  # noqa: E402
from footprint_tools_amd.cutcounts import CutCountDesc, _bind  # noqa: E402
from footprint_tools_amd.scan import DeviceArray  # noqa: E402

n_iv = int(sys.argv[1]) if len(sys.argv) > 1 else 437500
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50000000
pad = 55
rs = np.random.RandomState(0)
lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 2000).astype(np.int64)
gaps = rs.randint(200, 2000, n_iv)
starts = np.cumsum(lens + gaps) - lens  # one chromosome, sorted, non-overlapping cores
plen = (lens + 2 * pad + 1).astype(np.int32)
skey = (starts - pad - 1).astype(np.int64)
ekey = np.maximum.accumulate(skey + plen)
coff = np.concatenate([[0], np.cumsum(plen)[:-1]]).astype(np.int64)
total = int(plen.sum())
span = int(starts[-1] + lens[-1] + 1000)
# 60 % of the reads inside intervals (DHS-like enrichment), the rest anywhere
inside = rs.rand(n_reads) < 0.6
iv = rs.randint(0, n_iv, n_reads)
pos = np.where(inside, starts[iv] + (rs.rand(n_reads) * lens[iv]).astype(np.int64), rs.randint(0, span, n_reads)).astype(np.int32)
flag = np.where(rs.rand(n_reads) < 0.5, 16, 0).astype(np.uint16)
ctx = _lib.get_ctx()
L = _bind(ctx.L)
arrs = [np.zeros(n_reads, np.int32), pos, pos + 36, flag, np.full(n_reads, 30, np.uint8)]
dev = [DeviceArray(ctx, a.nbytes).upload(a) for a in arrs]
tmp = [DeviceArray(ctx, a.nbytes).upload(a) for a in (skey, ekey, plen, coff)]
cp, cm = DeviceArray(ctx, total * 8), DeviceArray(ctx, total * 8)
cp.upload(np.zeros(total)); cm.upload(np.zeros(total))
d = CutCountDesc()
d.n_reads = n_reads
d.ref_id, d.ref_start, d.ref_end, d.flag, d.mapq = (x.ptr for x in dev)
d.offset_plus, d.offset_minus, d.min_qual, d.remove_dups, d.remove_qcfail = 0, -1, 1, 0, 1
d.n_intervals = n_iv
d.start_key, d.maxend_key, d.padded_len, d.counts_off = (x.ptr for x in tmp)
d.counts_plus, d.counts_minus = cp.ptr, cm.ptr
_lib.check(L.fpt_cut_counts_dev(ctx.h, C.byref(d)))
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    _lib.check(L.fpt_cut_counts_dev(ctx.h, C.byref(d)))
ctx.synchronize()
dt = (time.perf_counter() - t0) / 5
s = cp.download(np.float64, total).sum() + cm.download(np.float64, total).sum()
print("cut counts: %d reads into %d intervals (%d padded positions) in %.3f ms -> %.3g reads/s; "
      "%.1f B read per alignment -> %.0f GB/s; counts added per pass %.4g"
      % (n_reads, n_iv, total, dt * 1e3, n_reads / dt, 15.0, 15.0 * n_reads / dt / 1e9, s / 6))
