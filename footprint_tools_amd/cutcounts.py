"""Cut-count ingestion: alignments -> per-base 5' cleavage counts (footprint_tools/cutcounts.py).

`bamfile` mirrors the reference class of the same name for the part the hot path needs:

    reads = bamfile(path, min_qual=1, remove_dups=False, remove_qcfail=True, offset=(0, -1))
    reads[interval]  ->  {'+': float64[len], '-': float64[len], 'fragments': []}     (lookup, :274-313)

and adds the batched form the scan consumes: `cut_counts_dev(intervals, pad)` fills the padded CSR
count arrays of a whole interval list on the device in one kernel launch per batch of reads.

The rule (cutcounts.py:119-145, 196-205, 231-248), applied on the device by k_cut_counts:
a read counts if it is not QC-fail (remove_qcfail) / duplicate (remove_dups), has MAPQ >= min_qual
and -- when paired -- is a proper pair and neither secondary nor supplementary; forward reads cut
at reference_start + offset[0] on '+', reverse reads at reference_end + offset[1] on '-'.

Differences, on purpose: the file is read once by the library's own BGZF/BAM reader (htslib /
pysam are not available here; no .bai, no CRAM; blocks inflated on a team of threads), all
alignments are kept ON THE DEVICE (13 bytes per alignment; the host keeps one start key per 256
alignments as a coarse index), and 'fragments' (the fragment intervals the reference also returns;
unused on this path) is always empty.  A query launches the counting kernel over the alignments
that can reach it only -- those whose start lies between the query's first base minus the longest
alignment span and its last base -- found in the coarse index, so a single-interval lookup costs
what its neighbourhood holds, not the file (files that are not coordinate-sorted are sorted once
at load).  Unmapped reads carrying a position are skipped.  Parity of the
file reader is unpinned (no pysam to compare with); the counting rule is pinned: tests compare
`lookup` with what the reference's own `bamfile.lookup` returned for 6,000 alignments under five
filter / offset settings (tests/golden/cutcounts.npz), and with hand-derived vectors.  One case of
malformed input is not reproduced: two alignments that share a name and are both flagged read 1 (or
both read 2) -- the reference's pairing dictionary (cutcounts.py:196-215) then drops the earlier one.
"""
import ctypes as C

import numpy as np

from . import _lib
from .scan import DeviceArray


class CutCountDesc(C.Structure):
    """struct fpt_cutcount_desc of include/fpt.h"""
    _fields_ = [
        ("n_reads", C.c_int64), ("ref_id", C.c_void_p), ("ref_start", C.c_void_p), ("ref_end", C.c_void_p),
        ("flag", C.c_void_p), ("mapq", C.c_void_p),
        ("offset_plus", C.c_int32), ("offset_minus", C.c_int32),
        ("min_qual", C.c_int32), ("remove_dups", C.c_int32), ("remove_qcfail", C.c_int32),
        ("n_intervals", C.c_int64), ("start_key", C.c_void_p), ("maxend_key", C.c_void_p),
        ("padded_len", C.c_void_p), ("counts_off", C.c_void_p),
        ("counts_plus", C.c_void_p), ("counts_minus", C.c_void_p), ("flip", C.c_void_p),
    ]


def _bind(L):
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.fpt_bam_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fpt_bam_close.argtypes = [vp]
    L.fpt_bam_n_refs.argtypes = [vp, C.POINTER(i32)]
    L.fpt_bam_ref.argtypes = [vp, i32, C.c_char_p, i32, C.POINTER(i64)]
    L.fpt_bam_read.argtypes = [vp, i64, vp, vp, vp, vp, vp, C.POINTER(i64)]
    L.fpt_bam_has_index.argtypes = [vp, C.POINTER(i32)]
    L.fpt_bam_seek_region.argtypes = [vp, i32, i64, i64]
    L.fpt_cut_counts_dev.argtypes = [vp, C.POINTER(CutCountDesc)]
    return L


def merge_regions(regions, gap=1 << 16):
    """(chrom, start, end) triples -> per chromosome, sorted and merged where less than `gap` bases
    apart (a seek costs more than reading a few blocks on)."""
    by = {}
    for c, a, b in regions:
        by.setdefault(c, []).append((int(a), int(b)))
    out = []
    for c in by:
        cur = None
        for a, b in sorted(by[c]):
            if cur and a <= cur[1] + gap:
                cur[1] = max(cur[1], b)
            else:
                if cur:
                    out.append((c, cur[0], cur[1]))
                cur = [a, b]
        out.append((c, cur[0], cur[1]))
    return out


def read_alignments(filepath, batch=1 << 20, regions=None):
    """(references, ref_id, start, end, flag, mapq) with references = [(name, length), ...] and the rest
    numpy arrays over the alignments: of the whole file in one sequential pass, or -- `regions`, a list
    of (chrom, start, end), and a BAI index beside the file -- of those regions only, through the index
    (what the reference's samfile.fetch does per interval, cutcounts.py:191).  Without an index
    `regions` is ignored.  Needs no GPU."""
    L = _bind(_lib.load())
    h = C.c_void_p()
    try:
        _lib.check(L.fpt_bam_open(str(filepath).encode(), C.byref(h)))
    except ValueError as e:  # the reference raises IOError("Cannot open BAM file: ...") (cutcounts.py:103)
        raise IOError(str(e))
    try:
        n = C.c_int32()
        _lib.check(L.fpt_bam_n_refs(h, C.byref(n)))
        refs = []
        buf = C.create_string_buffer(1024)
        for i in range(n.value):
            ln = C.c_int64()
            _lib.check(L.fpt_bam_ref(h, i, buf, 1024, C.byref(ln)))
            refs.append((buf.value.decode(), ln.value))
        parts = []

        def drain(min_start=None):
            while True:
                rid, st, en = (np.empty(batch, np.int32) for _ in range(3))
                fl, mq = np.empty(batch, np.uint16), np.empty(batch, np.uint8)
                got = C.c_int64()
                _lib.check(L.fpt_bam_read(h, batch, rid.ctypes.data, st.ctypes.data, en.ctypes.data, fl.ctypes.data,
                                          mq.ctypes.data, C.byref(got)))
                if got.value == 0:
                    break
                cols = [a[:got.value].copy() if got.value < batch // 4 else a[:got.value] for a in (rid, st, en, fl, mq)]
                if min_start is not None:  # an alignment that reaches over from the region before: it came with that one
                    keep = cols[1] >= min_start
                    if not keep.all():
                        cols = [a[keep] for a in cols]
                parts.append(cols)

        indexed = C.c_int32(0)
        _lib.check(L.fpt_bam_has_index(h, C.byref(indexed)))
        if regions is not None and indexed.value:
            names = {name: i for i, (name, _) in enumerate(refs)}
            prev = {}  # end of the region read before, per chromosome
            for c, a, b in merge_regions(regions):
                if c in names:  # (a chromosome the file does not have: no alignments)
                    _lib.check(L.fpt_bam_seek_region(h, names[c], max(int(a), 0), int(b)))
                    drain(prev.get(c))
                    prev[c] = int(b)
        else:
            drain()
        cols = [np.concatenate([p[k] for p in parts]) if parts else np.empty(0, dt)
                for k, dt in enumerate((np.int32, np.int32, np.int32, np.uint16, np.uint8))]
        return (refs,) + tuple(cols)
    finally:
        L.fpt_bam_close(h)


class bamfile(object):
    """Class to access a BAM file (and convert tags to cleavage counts); cutcounts.py:40-109."""

    def __init__(self, filepath, min_qual=1, remove_dups=False, remove_qcfail=True, offset=(0, -1),
                 is_cram=False, fasta_reference_filepath=None, ctx=None, regions=None):
        """`regions` (not in the reference's signature): (chrom, start, end) triples that bound what
        will be asked of this object -- with a BAI index beside the file only their alignments are
        read (each widened by 65,536 bases, far beyond any read length plus padding)."""
        if is_cram:
            raise IOError("Cannot open BAM file: %s (CRAM needs htslib, which this build does not have)" % filepath)
        self.filepath = filepath
        self.offset = offset
        self.min_qual = min_qual
        self.remove_dups = remove_dups
        self.remove_qcfail = remove_qcfail
        self._ctx = ctx
        if regions is not None:
            regions = [(c, int(a) - 65536, int(b) + 65536) for c, a, b in regions]
        self.references, rid, st, en, fl, mq = read_alignments(filepath, regions=regions)
        self._ref_index = {name: i for i, (name, _) in enumerate(self.references)}
        key = (rid.astype(np.int64) << 32) | np.clip(st, 0, None).astype(np.int64)
        if key.size and np.any(key[1:] < key[:-1]):  # not coordinate-sorted: sort once
            order = np.argsort(key, kind="stable")
            rid, st, en, fl, mq, key = rid[order], st[order], en[order], fl[order], mq[order], key[order]
        self._n_reads = int(rid.size)
        # coarse index: the start key of every 256th alignment, and how far an alignment's cut
        # can lie from its start (the longest reference span, plus the offsets)
        self._index_step = 256
        self._index = key[::self._index_step].copy()
        span = int((en.astype(np.int64) - st).max()) if rid.size else 0
        self._reach = max(span, 0) + abs(int(offset[0])) + abs(int(offset[1])) + 1
        self._host = (rid, st, en, fl, mq)  # dropped once the arrays are on the device
        self._dev = None

    def close(self):
        """Closes BAM file"""
        if self._dev:
            for d in self._dev:
                d.free()
            self._dev = None
        return True

    @property
    def n_reads(self):
        return self._n_reads

    def _reads_dev(self):
        if self._dev is None:
            ctx = self._ctx or _lib.get_ctx()
            self._ctx = ctx
            if self._host is None:
                raise RuntimeError("the alignments of this bamfile have been released (close())")
            self._dev = [DeviceArray(ctx, max(a.nbytes, 16)).upload(a) if a.size else DeviceArray(ctx, 16)
                         for a in self._host]
            self._host = None  # the device holds them now
        return self._dev

    def _read_range(self, lo_key, hi_key):
        """[a, b): the alignments whose start key lies in [lo_key - reach, hi_key], from the coarse index"""
        a = int(np.searchsorted(self._index, lo_key - self._reach, side="left")) - 1
        b = int(np.searchsorted(self._index, hi_key, side="right"))
        return max(a, 0) * self._index_step, min(b * self._index_step, self._n_reads)

    def cut_counts_ranges_dev(self, chroms, starts, lengths, counts_plus=None, counts_minus=None, flip=None, rid=None):
        """Counts of the ranges [starts[i], starts[i] + lengths[i]) on chroms[i], written back to
        back (CSR) into two device arrays of sum(lengths) doubles: returns (plus, minus, offsets).
        Existing arrays are accumulated into (several files of one dataset).  flip[i] = True gives
        range i the reference's strand '-' form ({'+': rev[::-1], '-': fw[::-1]}, cutcounts.py:307-311)."""
        ctx = self._ctx or _lib.get_ctx()
        self._ctx = ctx
        L = _bind(ctx.L)
        starts = np.asarray(starts, dtype=np.int64)
        lengths = np.asarray(lengths, dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        total = int(off[-1])
        if counts_plus is None:
            counts_plus, counts_minus = DeviceArray(ctx, max(total, 1) * 8).zero(), DeviceArray(ctx, max(total, 1) * 8).zero()
        if rid is None:  # (callers that hold the intervals as columns pass the file's reference ids)
            rid = np.array([self._ref_index.get(c, -1) for c in chroms], dtype=np.int64)
        known = rid >= 0  # a chromosome the file does not have: all zeros, like an empty fetch
        key = (rid << 32) | np.clip(starts, 0, None)
        # ranges starting before 0 keep their true length: shift by the clipped part
        clip = np.clip(-starts, 0, None)
        order = np.argsort(key[known], kind="stable")
        idx = np.nonzero(known)[0][order]
        if idx.size and self.n_reads:
            skey = key[idx]
            plen = (lengths[idx] - clip[idx]).astype(np.int32)
            ekey = np.maximum.accumulate(skey + plen)
            coff = (off[:-1][idx] + clip[idx]).astype(np.int64)
            # only the alignments that can reach the ranges (sorted by start key)
            r0, r1 = self._read_range(int(skey[0]), int(ekey[-1]))
            if r1 > r0:
                dev = self._reads_dev()
                arrs = [skey, ekey, plen, coff]
                if flip is not None:
                    fl = np.ascontiguousarray(np.asarray(flip, dtype=bool)[idx], dtype=np.uint8)
                    if np.any(clip[idx][fl.astype(bool)] > 0):
                        raise ValueError("a strand '-' range may not start before position 0")
                    arrs.append(fl)
                tmp = [DeviceArray(ctx, max(a.nbytes, 16)).upload(a) for a in arrs]
                d = CutCountDesc()
                d.n_reads = r1 - r0
                d.ref_id, d.ref_start, d.ref_end = (dev[k].ptr + 4 * r0 for k in range(3))
                d.flag, d.mapq = dev[3].ptr + 2 * r0, dev[4].ptr + r0
                d.offset_plus, d.offset_minus = int(self.offset[0]), int(self.offset[1])
                d.min_qual, d.remove_dups, d.remove_qcfail = int(self.min_qual), int(bool(self.remove_dups)), int(bool(self.remove_qcfail))
                d.n_intervals = idx.size
                d.start_key, d.maxend_key, d.padded_len, d.counts_off = (x.ptr for x in tmp[:4])
                d.flip = tmp[4].ptr if flip is not None else None
                d.counts_plus, d.counts_minus = counts_plus.ptr, counts_minus.ptr
                _lib.check(L.fpt_cut_counts_dev(ctx.h, C.byref(d)))
                ctx.synchronize()
                for x in tmp:
                    x.free()
        return counts_plus, counts_minus, off

    def cut_counts_dev(self, intervals, pad):
        """The padded count arrays of an interval list exactly as the fused scan reads them
        (`prediction.compute` fetches [start - pad - 1, end + pad), modeling/predict.pyx:132-134):
        (counts_plus, counts_minus) DeviceArrays in FootprintScanner.scan_dev's CSR layout.
        Intervals on strand '-' get the reference's mirrored and swapped arrays."""
        if hasattr(intervals, "cid"):  # intervals.interval_columns
            cols = intervals
            flip = cols.flip
            cp, cm, _ = self.cut_counts_ranges_dev(None, cols.start - (pad + 1), cols.end - cols.start + (2 * pad + 1),
                                                   flip=flip if flip.any() else None, rid=cols.lookup(self._ref_index, -1))
            return cp, cm
        ivs = list(intervals)
        flip = [getattr(iv, "strand", None) == "-" for iv in ivs]
        cp, cm, _ = self.cut_counts_ranges_dev([iv.chrom for iv in ivs], [iv.start - pad - 1 for iv in ivs],
                                               [iv.end - iv.start + 2 * pad + 1 for iv in ivs],
                                               flip=flip if any(flip) else None)
        return cp, cm

    def lookup(self, interval):
        """Lookup reads in a defined genomic region (cutcounts.py:274-313)."""
        n = interval.end - interval.start
        flip = getattr(interval, "strand", None) == "-"
        cp, cm, _ = self.cut_counts_ranges_dev([interval.chrom], [interval.start], [n], flip=[True] if flip else None)
        plus, minus = cp.download(np.float64, n), cm.download(np.float64, n)
        cp.free()
        cm.free()
        return {"+": plus, "-": minus, "fragments": []}

    def __getitem__(self, x):
        """intervals only (the allelically resolved counts of cutcounts.py:315-488 are outside this path: SURVEY.md
        section 8 f-3 names cutcounts.py:231-313 and :119-145)"""
        if not hasattr(x, "chrom") or not hasattr(x, "start") or not hasattr(x, "end"):
            raise TypeError("Query type not supported: %s" % type(x))
        return self.lookup(x)
