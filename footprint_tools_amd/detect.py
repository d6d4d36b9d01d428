"""Batched driver of the per-interval deviation statistics.

The reference computes them one interval at a time inside a worker pool
(`deviation_stats.__getitem__`, cli/detect.py:93-148).  `deviation_stats` here gathers a whole
batch of intervals' padded cut counts and sequence (what `prediction.compute` fetches,
modeling/predict.pyx:132-140), runs ONE fused scan launch and ONE empirical-FDR launch on the
GPU and hands back the same per-interval `{"interval", "stats"}` records:

    stats = column_stack(exp, obs, -log(pvals), -log(win_pvals), efdr)        detect.py:142-144

An interval for which the reference would have raised (`ZeroDivisionError` out of `fit_r`) gets
the reference's fallback row pvals = win_pvals = efdr = 1 (detect.py:136-140).

`read_func[interval]` must return {'+','-'} float arrays of len(interval) and
`fasta_func.fetch(chrom, start, end)` a string; `interval` needs chrom/start/end/widen(n), as
genome_tools.genomic_interval has.  The output writers produce the reference's bedGraph / BED
text (cli/utils.py:86-210).
"""
import ctypes as C
import re
import sys
from collections.abc import Sequence

import numpy as np

from . import __version__, _lib
from .scan import FootprintScanner
from .stats import utils


class _row_blocks(Sequence):
    """the per-interval row blocks of a step's matrix, as the list the reference's loader yields"""

    def __init__(self, table, off):
        self._table, self._off = table, off

    def __len__(self):
        return len(self._off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._table[int(self._off[i]):int(self._off[i + 1])]

    def __iter__(self):
        o = self._off.tolist() if hasattr(self._off, "tolist") else list(self._off)
        t = self._table
        return (t[a:b] for a, b in zip(o[:-1], o[1:]))


class deviation_stats(object):
    def __init__(self, intervals, read_func, fasta_func, bm, dm, half_win_width=5,
                 smoothing_half_win_width=50, smoothing_clip=0.01, fdr_shuffle_n=100, seed=0,
                 batch_size=4096, ctx=None):
        """intervals: sequence of interval objects (the reference reads them from a BED file,
        detect.py:50).  seed: key of the device null sampler (the reference seeds numpy's global
        RNG, detect.py:348-352; the draws here are reproducible but not numpy's)."""
        self.intervals = list(intervals)
        self.read_func, self.fasta_func, self.bm, self.dm = read_func, fasta_func, bm, dm
        self.half_win_width = half_win_width
        self.smoothing_half_win_width = smoothing_half_win_width
        self.smoothing_clip = smoothing_clip
        self.fdr_shuffle_n = fdr_shuffle_n
        self.seed = int(seed)
        self.batch_size = int(batch_size)
        self.padding = half_win_width + smoothing_half_win_width
        self._sc = None
        self._ctx = ctx
        self._bases_before = None
        self._cols = None

    def __len__(self):
        return len(self.intervals)

    def _scanner(self):
        if self._sc is None:
            if self.dm:
                self._sc = FootprintScanner(self.bm.table(), self.dm, self.half_win_width,
                                            self.smoothing_half_win_width, self.smoothing_clip,
                                            scales=(3,), default_propensity=self.bm.default, ctx=self._ctx)
            else:  # no dispersion model: expected and observed counts only (detect.py:145-146)
                self._sc = FootprintScanner(self.bm.table(), None, self.half_win_width,
                                            self.smoothing_half_win_width, self.smoothing_clip, scales=(),
                                            default_propensity=self.bm.default, ctx=self._ctx, nb_mode="none")
        return self._sc

    def _fetch(self, interval):
        """padded counts and sequence of one interval, exactly as predict.pyx:132-140 fetches them"""
        pad_interval = interval.widen(self.padding)
        pad_interval.start -= 1
        raw = self.read_func[pad_interval]
        seq = self.fasta_func.fetch(pad_interval.chrom, pad_interval.start - self.bm.offset(),
                                    pad_interval.end + self.bm.offset())
        cp = np.ascontiguousarray(raw['+'], dtype=np.float64)
        cm = np.ascontiguousarray(raw['-'], dtype=np.float64)
        if isinstance(seq, str):
            seq = seq.encode("ascii", "replace")
        return cp, cm, np.frombuffer(bytes(seq), dtype=np.uint8)

    def _device_inputs(self):
        """True when the two readers can hand a whole batch over without a round trip per interval:
        a `cutcounts.bamfile` (its alignments live on the device, `cut_counts_dev` fills the padded
        count arrays there in one launch) and a `fasta.FastaFile` (`fetch_batch`)."""
        return (hasattr(self.read_func, "cut_counts_dev") and hasattr(self.fasta_func, "fetch_batch")
                and self.bm.offset() == 3 and getattr(self.read_func, "_ctx", None) in (None, self._scanner().ctx))

    def _runs(self, indices):
        """maximal runs of consecutive indices: one FDR call each keeps the RNG counters global"""
        idx = np.asarray(indices, dtype=np.int64)
        cuts = np.nonzero(np.diff(idx) != 1)[0] + 1
        bounds = [0] + cuts.tolist() + [int(idx.size)]
        runs = list(zip(bounds[:-1], bounds[1:]))
        if self._bases_before is None:
            # global base index of an interval = bases of all intervals before it in the full list,
            # so the null draws do not depend on how the list is batched or sharded
            cols = self._columns()
            self._bases_before = np.concatenate([[0], np.cumsum(cols.end - cols.start)])
        return runs

    def _columns(self):
        """the interval list as columns (read off the objects once)"""
        if self._cols is None:
            from .intervals import interval_columns
            self._cols = interval_columns.of(self.intervals)
        return self._cols

    def _compute_on_device(self, indices, ivs, lens, off):
        """The batch without leaving the GPU between the steps: cut counts from the alignments on
        the device, one upload of the sequence bytes, scan, FDR, one download of the tracks."""
        from .scan import DeviceArray
        sc = self._scanner()
        ctx = sc.ctx
        if getattr(self.read_func, "_ctx", None) is None:
            self.read_func._ctx = ctx
        total, n_iv = int(off[-1]), len(ivs)
        S = len(sc.scales)
        n_tracks = 3 + S + (1 if self.dm else 0)
        from .cutcounts import bamfile
        from .fasta import FastaFile
        cols = self._columns().take(indices)  # (this package's readers take the batch as columns)
        d_cp, d_cm = self.read_func.cut_counts_dev(cols if isinstance(self.read_func, bamfile) else ivs, self.padding)
        bufs = [d_cp, d_cm]
        try:
            if hasattr(self.fasta_func, "fetch_batch_dev"):  # the FASTA bytes live on the device too
                d_sq, n_sq = self.fasta_func.fetch_batch_dev(ctx, cols if isinstance(self.fasta_func, FastaFile) else ivs,
                                                             self.padding)
            else:
                sq = self.fasta_func.fetch_batch(ivs, self.padding)
                d_sq, n_sq = DeviceArray(ctx, max(sq.nbytes, 16)).upload(sq), sq.size
            bufs.append(d_sq)
            if n_sq != total + n_iv * (2 * self.padding + 7):
                raise ValueError("fasta_func returned sequence of the wrong length")
            d_off = DeviceArray(ctx, off.nbytes).upload(off); bufs.append(d_off)
            d_out = DeviceArray(ctx, max(n_tracks * total * 8, 16)); bufs.append(d_out)
            d_st = DeviceArray(ctx, max(n_iv * 4, 16)).upload(np.zeros(max(n_iv, 1), np.int32)); bufs.append(d_st)
            t8 = total * 8
            sc.scan_dev(n_iv, d_cp.ptr, d_cm.ptr, d_sq.ptr, exp_out=d_out.ptr, obs_out=d_out.ptr + t8,
                        pval_out=d_out.ptr + 2 * t8, winp_out=(d_out.ptr + 3 * t8) if S else None,
                        interval_off_dev=d_off.ptr, interval_off_host=off, status_out=d_st.ptr)
            if self.dm:
                for a, b in self._runs(indices):
                    roff = np.ascontiguousarray(off[a:b + 1] - off[a])
                    d_roff = DeviceArray(ctx, roff.nbytes).upload(roff); bufs.append(d_roff)
                    sc.fdr_dev(b - a, d_out.ptr + int(off[a]) * 8, d_out.ptr + 3 * t8 + int(off[a]) * 8,
                               d_out.ptr + (3 + S) * t8 + int(off[a]) * 8, times=self.fdr_shuffle_n, seed=self.seed,
                               half_win_width=3, interval_off_dev=d_roff.ptr,
                               base_index0=int(self._bases_before[indices[a]]), obs=d_out.ptr + t8 + int(off[a]) * 8,
                               interval_off_host=roff)
            if self.dm:
                # the five record columns (detect.py:142-144, and the fallback rows of :136-140)
                # assembled on the device: one download of the (bases, 5) matrix
                d_tab = DeviceArray(ctx, max(5 * t8, 16)); bufs.append(d_tab)
                _lib.check(ctx.L.fpt_detect_columns_dev(ctx.h, n_iv, 0, d_off.ptr, total, d_st.ptr, d_out.ptr,
                                                        d_out.ptr + t8, d_out.ptr + 2 * t8, d_out.ptr + 3 * t8,
                                                        d_out.ptr + (3 + S) * t8, d_tab.ptr))
                ctx.synchronize()
                return d_tab.download(np.float64, 5 * total).reshape(total, 5), None
            ctx.synchronize()
            flat = d_out.download(np.float64, 2 * total).reshape(2, total)
        finally:
            for x in bufs:
                x.free()
        return dict(exp=flat[0], obs=flat[1]), None

    def compute(self, indices):
        """statistics of intervals `indices` (one GPU batch); list of {"interval", "stats"}"""
        return self._compute(indices)[0]

    def _compute(self, indices, records=True):
        """(records, the (bases, columns) matrix their `stats` are row blocks of, row offsets);
        records=False: the intervals instead of the records (batch_iter makes the blocks on demand)"""
        if isinstance(indices, range) and indices.step == 1 and 0 <= indices.start <= indices.stop <= len(self.intervals):
            ivs = self.intervals[indices.start:indices.stop]
        else:
            indices = list(indices)
            ivs = [self.intervals[i] for i in indices]
        if not ivs:
            return [], None, None
        cols = self._columns().take(indices)
        lens = cols.end - cols.start
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        sc = self._scanner()
        if self._device_inputs():  # (strand '-' intervals included: cut_counts_dev mirrors and swaps their counts)
            res, efdr = self._compute_on_device(indices, ivs, lens, off)
            if self.dm:  # the (bases, 5) table came back assembled
                if not records:
                    return ivs, res, off
                o = off.tolist()
                return [{"interval": iv, "stats": res[a:b]} for iv, a, b in zip(ivs, o[:-1], o[1:])], res, off
        else:
            cps, cms, sqs = zip(*(self._fetch(iv) for iv in ivs))
            for L, cp, sq in zip(lens, cps, sqs):
                if cp.size != L + 2 * self.padding + 1 or sq.size != cp.size + 6:
                    raise ValueError("read_func / fasta_func returned arrays of the wrong length")
            res = sc.scan(np.concatenate(cps), np.concatenate(cms), np.concatenate(sqs), interval_off=off)
            efdr = None
        if not self.dm:
            table = np.column_stack((res["exp"], res["obs"]))
            if not records:
                return ivs, table, off
            return [{"interval": iv, "stats": table[a:b]} for iv, a, b in zip(ivs, off[:-1], off[1:])], table, off
        if efdr is None:
            efdr = np.empty(off[-1])
            for a, b in self._runs(indices):
                sl = slice(off[a], off[b])
                efdr[sl] = sc.fdr(res["exp"][sl], res["winp"][0][sl], times=self.fdr_shuffle_n, seed=self.seed,
                                  half_win_width=3, interval_off=off[a:b + 1] - off[a],
                                  base_index0=int(self._bases_before[indices[a]]), obs=res["obs"][sl])
        # the five columns for the whole batch at once; a record's `stats` is its block of rows
        pv, wp, ef = np.array(res["pval"], copy=True), np.array(res["winp"][0], copy=True), np.array(efdr, copy=True)
        for j in np.nonzero(res["status"])[0]:  # the reference's `except Exception` branch (detect.py:136-140)
            pv[off[j]:off[j + 1]] = wp[off[j]:off[j + 1]] = ef[off[j]:off[j + 1]] = 1.0
        with np.errstate(all="ignore"):  # detect.py:41 np.seterr(all="ignore")
            table = np.column_stack((res["exp"], res["obs"], -np.log(pv), -np.log(wp), ef))
        if not records:
            return ivs, table, off
        return [{"interval": iv, "stats": table[a:b]} for iv, a, b in zip(ivs, off[:-1], off[1:])], table, off

    def __getitem__(self, index):
        return self.compute([index])[0]

    def batch_iter(self, batch_size=None):
        bs = int(batch_size or self.batch_size)
        for a in range(0, len(self.intervals), bs):
            ivs, table, off = self._compute(range(a, min(a + bs, len(self.intervals))), records=False)
            # the step's rows as one matrix (for write_batch_to_output); "stats" cuts an interval's block out
            # of it when asked (200,000 slices made up front were 40 % of a step's host time)
            yield {"interval": ivs, "stats": _row_blocks(table, off), "table": table, "row_off": off}


# ---- output writers: same text as cli/utils.py:86-210 ---------------------------------------

def write_output_header(columns, file=sys.stdout, delim="\t", include_name=True, extra=None):
    lines = ["# generated by footprint_tools_amd version %s" % __version__]
    if extra:
        lines += ["# " + e for e in (extra if isinstance(extra, list) else [extra])]
    cols = ["chrom", "start", "end"] + (["name"] if include_name else []) + list(columns)
    lines.append("# " + delim.join(cols))
    file.write("\n".join(lines) + "\n")
    file.flush()


_FIXED = re.compile(r"^0?\.(\d{1,2})f$")


def _native_stats_text(chrom, start, stats, rows, delim, precision):
    """the lines through the library's formatter (fpt_format_stats: the same correctly rounded
    decimals, ~50x the speed of formatting value by value in Python); None if it does not apply"""
    if len(delim) != 1 or not delim.isascii() or not str(chrom).isascii() or stats.ndim != 2:
        return None
    m = np.ascontiguousarray(stats, dtype=np.float64)
    n, k = m.shape
    sel = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
    count = n if sel is None else sel.size
    if count == 0:
        return ""
    with np.errstate(all="ignore"):
        largest = float(np.abs(m).max()) if m.size else 0.0
        if not np.isfinite(largest):  # nan / inf somewhere: the largest finite magnitude
            finite = np.abs(m[np.isfinite(m)])
            largest = float(finite.max()) if finite.size else 0.0
    digits = int(np.floor(np.log10(max(largest, 1.0)))) + 1
    cap = count * (len(str(chrom)) + 1 + 2 * (len(str(int(start) + n)) + 1) + k * (digits + precision + 4) + 1) + 64
    buf = np.empty(cap, dtype=np.uint8)
    used = C.c_int64()
    L = _lib.load()
    _lib.check(L.fpt_format_stats(str(chrom).encode(), int(start), m.ctypes.data, n, k,
                                  sel.ctypes.data if sel is not None else None, count, delim.encode(), precision,
                                  buf.ctypes.data, cap, C.byref(used)))
    return buf[:used.value].tobytes().decode("ascii")


def write_stats_to_output(interval, stats, file=sys.stdout, delim="\t", filter_fn=None, fmt_string="0.4f"):
    rows = np.nonzero(filter_fn(stats))[0] if filter_fn else None
    chrom, start = interval.chrom, interval.start
    fixed = _FIXED.match(fmt_string)
    # the library's formatter takes up to 30 decimals; anything else goes value by value below
    text = (_native_stats_text(chrom, start, stats, rows, delim, int(fixed.group(1)))
            if fixed and int(fixed.group(1)) <= 30 else None)
    if text is None:  # any other format string: value by value, as the reference does
        fmt = "{0:" + fmt_string + "}"
        text = "".join(
            delim.join([str(chrom), str(start + i), str(start + i + 1)] + [fmt.format(v) for v in stats[i, :]])
            + "\n" for i in (range(stats.shape[0]) if rows is None else rows))
    file.write(text)


def write_batch_to_output(batch, file=sys.stdout, delim="\t", fmt_string="0.4f"):
    """`write_stats_to_output` for every interval of a `batch_iter` step, in one call of the
    library: the step's rows are one matrix (`batch["table"]`, `batch["row_off"]`), formatted on a
    team of threads -- and compressed there too when `file` is a `tabix.TrackWriter`.  The same
    bytes as the loop over `zip(batch["interval"], batch["stats"])` (cli/detect.py:399-411)."""
    ivs = batch["interval"]
    fixed = _FIXED.match(fmt_string)
    table = batch.get("table")
    if (table is None or not fixed or int(fixed.group(1)) > 30 or len(delim) != 1 or not delim.isascii()
            or not all(str(iv.chrom).isascii() for iv in ivs)):
        for iv, st in zip(ivs, batch["stats"]):
            write_stats_to_output(iv, st, file=file, delim=delim, fmt_string=fmt_string)
        return
    precision = int(fixed.group(1))
    chroms, starts = [iv.chrom for iv in ivs], [iv.start for iv in ivs]
    if hasattr(file, "write_stats") and delim == "\t":
        file.write_stats(chroms, starts, batch["row_off"], table, precision)
        return
    names, ids, st, off, m = _lib.batch_text_args(chroms, starts, batch["row_off"], table)
    L = _lib.load()
    args = (len(st), names, len(names), ids.ctypes.data, st.ctypes.data, off.ctypes.data, m.ctypes.data, m.shape[1],
            delim.encode(), precision)
    used = C.c_int64()
    buf = np.empty(int(off[-1] - off[0]) * (max(len(str(c)) for c in chroms) + 24 + m.shape[1] * (precision + 8)) + 64
                   if len(st) else 0, dtype=np.uint8)
    rc = L.fpt_format_stats_batch(*args, buf.ctypes.data, buf.size, C.byref(used))
    if rc != 0 and used.value > buf.size:  # long values: the library says how much it needs
        buf = np.empty(used.value, dtype=np.uint8)
        rc = L.fpt_format_stats_batch(*args, buf.ctypes.data, buf.size, C.byref(used))
    _lib.check(rc)
    view = memoryview(buf)[:used.value]
    raw = getattr(file, "buffer", None)  # a text file over a binary one: the bytes go underneath (no str copy)
    if raw is not None and str(getattr(file, "encoding", "")).lower().replace("_", "-") in (
            "utf-8", "utf8", "ascii", "us-ascii", "latin-1", "iso-8859-1", "iso8859-1", "cp1252"):
        file.flush()
        raw.write(view)
        return
    try:
        file.write(str(view, "ascii"))
    except TypeError:  # a binary file
        file.write(view)


def write_track(ds, filename, batch_size=None, header_columns=None, fmt_string="0.4f", level=None):
    """A whole `detect` run into a bgzip-compressed, tabix-indexed track: every step of
    `ds.batch_iter()` through `write_batch_to_output` into a `tabix.TrackWriter`, the writing of one
    step (threads inside the library, no interpreter lock held) overlapped with the statistics of
    the next -- the role of the reference's writer thread behind its queue (cli/detect.py:364-411).
    Intervals must come sorted the way a track is; `level` is the zlib level of the track (None: 6,
    bgzip's).  Returns the number of bases written."""
    import queue
    import threading
    from .tabix import TrackWriter
    q, err, n = queue.Queue(maxsize=2), [], 0
    w = TrackWriter(filename, level=level)

    def drain():
        while True:
            batch = q.get()
            if batch is None:
                return
            if not err:
                try:
                    write_batch_to_output(batch, file=w, fmt_string=fmt_string)
                except Exception as e:  # noqa: BLE001 -- handed to the caller below
                    err.append(e)

    t = threading.Thread(target=drain, name="fpt-track-writer", daemon=True)
    t.start()
    try:
        if header_columns is not None:
            write_output_header(header_columns, file=w, include_name=False)
        for batch in ds.batch_iter(batch_size):
            if err:
                break
            n += int(batch["row_off"][-1]) if "row_off" in batch else sum(s.shape[0] for s in batch["stats"])
            q.put(batch)
    finally:
        q.put(None)
        t.join()
        try:
            w.close()
        except Exception as e:  # noqa: BLE001
            err.append(e)
    if err:
        raise err[0]
    return n


def write_segment_batch_to_output(intervals, segments, name=".", file=sys.stdout, delim="\t", fmt_string="0.4f"):
    """BED lines of `FootprintScanner.segment` output (one device pass over the whole FDR track)
    in the format of write_segments_to_output; `intervals` are the batch's interval objects."""
    fmt = "{0:" + fmt_string + "}"
    file.write("".join(
        delim.join([str(intervals[i].chrom), str(intervals[i].start + s), str(intervals[i].start + e), name,
                    fmt.format(sc)]) + "\n"
        for i, s, e, sc in zip(segments["interval"], segments["start"], segments["end"], segments["score"])))


def write_segments_to_output(interval, stats, threshold, name=".", file=sys.stdout, delim="\t",
                             score_fn=np.min, decreasing=False, fmt_string="0.4f"):
    assert stats.ndim == 1
    fmt = "{0:" + fmt_string + "}"
    for s, e in utils.segment(stats, threshold, 3, decreasing=decreasing):
        file.write(delim.join([str(interval.chrom), str(interval.start + s), str(interval.start + e), name,
                               fmt.format(score_fn(stats[s:e]))]) + "\n")
