"""Batched stand-in of `ftd learn_dm` (cli/learn_dm.py): expected vs observed cut counts of a set
of intervals, their 2-D histogram, and the dispersion model learned from it.

The reference computes `column_stack((exp, obs))` one interval at a time in a worker pool
(`expected_counts.__getitem__`, cli/learn_dm.py:75-107) and increments the histogram in a Python
loop (:276-287).  Here a whole batch of intervals goes through ONE fused launch that stops after
the expected counts (fpt_nb_mode FPT_NB_NONE: there is no dispersion model yet) and ONE histogram
launch; the tracks never leave the GPU, only the 200 x 1000 histogram does.  The model fit itself
(`dispersion.learn_dispersion_model`) is host code, as in the reference.
"""
import numpy as np

from . import _lib
from .modeling import dispersion
from .scan import DeviceArray, FootprintScanner


class expected_counts(object):
    def __init__(self, intervals, read_func, fasta_func, bm, half_win_width=5, smoothing_half_win_width=0,
                 smoothing_clip=0.01, batch_size=4096, ctx=None):
        """intervals / read_func / fasta_func as for detect.deviation_stats.  The reference builds
        its predictor with the class defaults (no smoothing), cli/learn_dm.py:62-67."""
        self.intervals = list(intervals)
        self.read_func, self.fasta_func, self.bm = read_func, fasta_func, bm
        self.padding = half_win_width + smoothing_half_win_width
        self.batch_size = int(batch_size)
        self._sc = FootprintScanner(bm.table(), None, half_win_width, smoothing_half_win_width, smoothing_clip,
                                    scales=(), default_propensity=bm.default, ctx=ctx, nb_mode="none")

    def __len__(self):
        return len(self.intervals)

    def _fetch(self, interval):
        pad_interval = interval.widen(self.padding)
        pad_interval.start -= 1  # predict.pyx:133
        raw = self.read_func[pad_interval]
        seq = self.fasta_func.fetch(pad_interval.chrom, pad_interval.start - self.bm.offset(),
                                    pad_interval.end + self.bm.offset())
        if isinstance(seq, str):
            seq = seq.encode("ascii", "replace")
        return (np.ascontiguousarray(raw['+'], dtype=np.float64), np.ascontiguousarray(raw['-'], dtype=np.float64),
                np.frombuffer(bytes(seq), dtype=np.uint8))

    def _upload(self, indices):
        from .cutcounts import bamfile
        from .fasta import FastaFile
        from .intervals import interval_columns
        if isinstance(indices, range) and indices.step == 1 and 0 <= indices.start <= indices.stop <= len(self.intervals):
            ivs = self.intervals[indices.start:indices.stop]
        else:
            indices = list(indices)
            ivs = [self.intervals[i] for i in indices]
        if getattr(self, "_cols", None) is None:  # the interval list as columns, read off the objects once
            self._cols = interval_columns.of(self.intervals)
        cols = self._cols.take(indices)
        lens = cols.end - cols.start
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ctx = self._sc.ctx
        if (hasattr(self.read_func, "cut_counts_dev") and hasattr(self.fasta_func, "fetch_batch")
                and self.bm.offset() == 3 and getattr(self.read_func, "_ctx", None) in (None, ctx)):
            # a cutcounts.bamfile and a fasta.FastaFile: the count arrays are filled on the device
            # from the alignments held there, no round trip per interval (see detect.deviation_stats)
            if getattr(self.read_func, "_ctx", None) is None:
                self.read_func._ctx = ctx
            d_cp, d_cm = self.read_func.cut_counts_dev(cols if isinstance(self.read_func, bamfile) else ivs, self.padding)
            want = int(off[-1]) + len(ivs) * (2 * self.padding + 7)
            if isinstance(self.fasta_func, FastaFile):  # the FASTA bytes live on the device too
                d_sq, n_sq = self.fasta_func.fetch_batch_dev(ctx, cols, self.padding)
                if n_sq != want:
                    raise ValueError("fasta_func returned sequence of the wrong length")
                return off, [d_cp, d_cm, d_sq, DeviceArray(ctx, off.nbytes).upload(off),
                             DeviceArray(ctx, max(2 * int(off[-1]) * 8, 16))]
            sq = self.fasta_func.fetch_batch(ivs, self.padding)
            if sq.size != want:
                raise ValueError("fasta_func returned sequence of the wrong length")
        else:
            cps, cms, sqs = zip(*(self._fetch(iv) for iv in ivs))
            for L, cp, sq in zip(lens, cps, sqs):
                if cp.size != L + 2 * self.padding + 1 or sq.size != cp.size + 6:
                    raise ValueError("read_func / fasta_func returned arrays of the wrong length")
            cp, cm, sq = np.concatenate(cps), np.concatenate(cms), np.concatenate(sqs)
            d_cp, d_cm = DeviceArray(ctx, max(cp.nbytes, 16)).upload(cp), DeviceArray(ctx, max(cm.nbytes, 16)).upload(cm)
        bufs = [d_cp, d_cm, DeviceArray(ctx, max(sq.nbytes, 16)).upload(sq), DeviceArray(ctx, off.nbytes).upload(off),
                DeviceArray(ctx, max(2 * int(off[-1]) * 8, 16))]
        return off, bufs

    def _scan(self, off, bufs):
        total = int(off[-1])
        self._sc.scan_dev(off.size - 1, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, exp_out=bufs[4].ptr,
                          obs_out=bufs[4].ptr + total * 8, interval_off_dev=bufs[3].ptr, interval_off_host=off)

    def compute(self, indices):
        """[column_stack((exp, obs)) for each interval] -- what the reference's dataset yields"""
        if not isinstance(indices, range):
            indices = list(indices)
        if len(indices) == 0:
            return []
        off, bufs = self._upload(indices)
        try:
            self._scan(off, bufs)
            self._sc.ctx.synchronize()
            total = int(off[-1])
            flat = bufs[4].download(np.float64, 2 * total)
        finally:
            for b in bufs:
                b.free()
        return [np.column_stack((flat[a:b], flat[total + a:total + b])) for a, b in zip(off[:-1], off[1:])]

    def __getitem__(self, index):
        return self.compute([index])[0]

    def histogram(self, dims=(200, 1000)):
        """hist[int(exp), int(obs)] over all intervals (cli/learn_dm.py:273-287), accumulated on
        the device batch after batch"""
        ctx = self._sc.ctx
        rows, cols = int(dims[0]), int(dims[1])
        d_h = DeviceArray(ctx, rows * cols * 8).upload(np.zeros(rows * cols, np.uint64))
        try:
            for a in range(0, len(self.intervals), self.batch_size):
                off, bufs = self._upload(range(a, min(a + self.batch_size, len(self.intervals))))
                try:
                    self._scan(off, bufs)
                    total = int(off[-1])
                    _lib.check(ctx.L.fpt_hist2d_dev(ctx.h, bufs[4].ptr, bufs[4].ptr + total * 8, total, rows,
                                                    cols, d_h.ptr))
                    ctx.synchronize()
                finally:
                    for b in bufs:
                        b.free()
            return d_h.download(np.uint64, rows * cols).reshape(rows, cols).astype(np.int64)
        finally:
            d_h.free()


def learn_dm(intervals, read_func, fasta_func, bm, half_win_width=5, hist_dims=(200, 1000), seed=None,
             batch_size=4096, ctx=None, **fit_kwargs):
    """`ftd learn_dm` end to end: histogram on the GPU, model fit on the host.  seed: numpy's
    global RNG (used only when a histogram row holds more than 1e5 points), cli/learn_dm.py:262-268."""
    np.random.seed(seed)
    ds = expected_counts(intervals, read_func, fasta_func, bm, half_win_width=half_win_width,
                         batch_size=batch_size, ctx=ctx)
    hist = ds.histogram(hist_dims)
    return dispersion.learn_dispersion_model(hist, **fit_kwargs)
