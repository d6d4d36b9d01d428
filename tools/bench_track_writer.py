"""Timing of the batch writer alone (diagnostic, host only): a made-up (bases, 5) table of N_IV intervals
through TrackWriter.write_stats; FPT_TRACK_TIMES=1 prints where each call spends its time."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd.tabix import TrackWriter  # noqa: E402

n_iv = int(os.environ.get("N_IV", "8192"))
rs = np.random.RandomState(1)
lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 1500).astype(np.int64)
off = np.concatenate([[0], np.cumsum(lens)])
starts = 1000 + np.cumsum(lens + 20) - lens
table = np.abs(rs.standard_normal((int(off[-1]), 5))) * 10.0 ** rs.randint(-3, 2, (int(off[-1]), 5))
chroms = ["chr1"] * n_iv
path = os.path.join(tempfile.mkdtemp(), "t.bed.gz")
for rep in range(3):
    t0 = time.perf_counter()
    with TrackWriter(path) as w:
        w.write_stats(chroms, starts, off, table)
        t1 = time.perf_counter()
    t2 = time.perf_counter()
    print("%d intervals, %d rows: write_stats %.1f ms, close %.1f ms -> %.3g rows/s" % (n_iv, off[-1], (t1 - t0) * 1e3, (t2 - t1) * 1e3, off[-1] / (t2 - t0)))
