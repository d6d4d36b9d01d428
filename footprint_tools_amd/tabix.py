"""Per-nucleotide statistics tracks written by `ftd detect` (bgzip-compressed bedGraph, indexed
with tabix in the reference's workflow) -- indexed region access without pysam / htslib.

The reference opens them with `pysam.TabixFile(fn)` and walks `fetch(chrom, start, end,
parser=pysam.asTuple())` per interval (cli/post.py:52-87).  `TabixFile` here keeps that call
surface over the library's own reader (fpt_track_* of include/fpt.h, fpt_track.cpp): the file is
mapped, `<file>.tbi` is used when present (else an index of the same shape is built by one pass at
open), a query inflates only the BGZF members it needs, and `fetch_batch` serves a whole interval
list on a team of threads, scattering the wanted columns straight into (bases) arrays -- the loop
of `_load_data` (cli/post.py:70-83) for all intervals at once.  Parity of the reader is unpinned
(no pysam here); the column meaning is the reference's writer's (cli/utils.py:119-144: chrom,
start, start+1, exp, obs, -log p, -log win-p, fdr).
"""
import ctypes as C

import numpy as np

from . import _lib


def _bind(L):
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.fpt_track_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fpt_track_close.argtypes = [vp]
    L.fpt_track_n_refs.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.fpt_track_ref.argtypes = [vp, i32, C.c_char_p, i32]
    L.fpt_track_fetch.argtypes = [vp, i64, vp, vp, vp, vp, i32, vp, vp, vp]
    L.fpt_track_fetch_rows.argtypes = [vp, C.c_char_p, i64, i64, i32, vp, i64, vp, vp, C.POINTER(i64)]
    L.fpt_track_writer_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fpt_track_writer_write.argtypes = [vp, C.c_char_p, i64]
    L.fpt_track_writer_set_level.argtypes = [vp, i32]
    L.fpt_track_writer_close.argtypes = [vp]
    return L


class TrackWriter(object):
    """bedGraph text -> `filename` (bgzip-compressed) + `filename`.tbi: what `bgzip` and `tabix -p bed`
    make of `ftd detect`'s output in the reference's workflow, so that a track written here can be
    read back by `TabixFile` (and by pysam / tabix).  A file-like object for the writers of
    detect.py (`write_stats_to_output(..., file=TrackWriter(path))`): `write` takes str or bytes in
    pieces of any size; lines must be sorted by position within a chromosome, chromosomes
    contiguous.  `close` finishes both files (and raises on the first error met)."""

    def __init__(self, filename, level=None):
        self.filename = filename
        self.L = _bind(_lib.load())
        h = C.c_void_p()
        try:
            _lib.check(self.L.fpt_track_writer_open(str(filename).encode(), C.byref(h)))
        except ValueError as e:
            raise IOError(str(e))
        self.h = h
        if level is not None:  # zlib level of the members: 6 is bgzip's (the default), 1 trades ~20 % of size for ~3x the speed
            _lib.check(self.L.fpt_track_writer_set_level(h, int(level)))

    def write(self, text):
        if self.h is None:
            raise ValueError("write to a closed TrackWriter")
        data = text.encode() if isinstance(text, str) else bytes(text)
        _lib.check(self.L.fpt_track_writer_write(self.h, data, len(data)))
        return len(data)

    def write_stats(self, chroms, starts, row_off, table, precision=4):
        """the lines of a whole batch (detect.write_batch_to_output) formatted and compressed inside
        the library: `table` is the (rows, columns) matrix, interval j owns rows
        row_off[j]:row_off[j+1] and starts at starts[j] of chroms[j]"""
        if self.h is None:
            raise ValueError("write to a closed TrackWriter")
        names, ids, st, off, m = _lib.batch_text_args(chroms, starts, row_off, table)
        _lib.check(self.L.fpt_track_writer_write_stats(self.h, len(st), names, len(names), ids.ctypes.data, st.ctypes.data,
                                                       off.ctypes.data, m.ctypes.data, m.shape[1], int(precision)))

    def flush(self):
        pass

    def close(self):
        if getattr(self, "h", None):
            h, self.h = self.h, None
            _lib.check(self.L.fpt_track_writer_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TabixFile(object):
    def __init__(self, filename):
        self.filename = filename
        self.L = _bind(_lib.load())
        h = C.c_void_p()
        try:
            _lib.check(self.L.fpt_track_open(str(filename).encode(), C.byref(h)))
        except ValueError as e:  # like pysam: a file that is not there (or not a track) fails at open
            raise IOError(str(e))
        self.h = h
        n, idx = C.c_int32(), C.c_int32()
        _lib.check(self.L.fpt_track_n_refs(h, C.byref(n), C.byref(idx)))
        self.has_tbi = bool(idx.value)
        buf = C.create_string_buffer(4096)
        self.contigs = []
        for i in range(n.value):
            _lib.check(self.L.fpt_track_ref(h, i, buf, 4096))
            self.contigs.append(buf.value.decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.fpt_track_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fetch_columns(self, chrom, start, end, cols=None, n_cols=8):
        """(positions, values): rows with start <= position < end.  values[:, k] is file column
        cols[k] (0-based), by default columns 1 .. n_cols-1 (so values[:, k] is file column k+1)."""
        cols = np.arange(1, n_cols, dtype=np.int32) if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
        cap = max(int(end) - int(start), 0) + 16
        while True:
            pos = np.empty(cap, np.int64)
            vals = np.empty((cap, cols.size), np.float64)
            n = C.c_int64()
            _lib.check(self.L.fpt_track_fetch_rows(self.h, str(chrom).encode(), int(start), int(end), cols.size,
                                                   cols.ctypes.data, cap, pos.ctypes.data, vals.ctypes.data, C.byref(n)))
            if n.value <= cap:
                return pos[:n.value], vals[:n.value]
            cap = n.value  # rows wider or denser than one per base: again with room

    def fetch(self, chrom, start, end, parser=None):
        """rows as tuples of strings, like pysam's asTuple parser: (chrom, start, end, column 3, ...)"""
        pos, vals = self.fetch_columns(chrom, start, end, cols=np.arange(2, 8))
        for p, row in zip(pos, vals):
            yield (chrom, repr(int(p))) + tuple(repr(int(v)) if k == 0 else repr(float(v)) for k, v in enumerate(row))

    def fetch_batch(self, chroms, starts, ends, cols, out_off=None, out=None, present=None):
        """The rows of many intervals scattered into (bases) arrays: for a row at position x of
        interval i, out[k][out_off[i] + x - starts[i]] = file column cols[k] and present[...] = 1.
        Returns (out, present, out_off); arrays not given are made (nan / 0-filled)."""
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        ends = np.ascontiguousarray(ends, dtype=np.int64)
        n = starts.size
        if out_off is None:
            out_off = np.concatenate([[0], np.cumsum(np.maximum(ends - starts, 0))]).astype(np.int64)
        out_off = np.ascontiguousarray(out_off, dtype=np.int64)
        total = int(out_off[-1]) if out_off.size > n else int((out_off + np.maximum(ends - starts, 0)).max(initial=0))
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        if out is None:
            out = [np.full(total, np.nan) for _ in range(cols.size)]
        if present is None:
            present = np.zeros(total)
        names = (C.c_char_p * max(n, 1))(*[str(c).encode() for c in chroms])
        ptrs = (C.c_void_p * max(cols.size, 1))(*[a.ctypes.data for a in out])
        _lib.check(self.L.fpt_track_fetch(self.h, n, names, starts.ctypes.data, ends.ctypes.data, out_off.ctypes.data,
                                          cols.size, cols.ctypes.data, ptrs, present.ctypes.data))
        return out, present, out_off
