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

`bamfile.lookup_allelic(chrom, start, end, pos, ref, alt, flip=False)` (cutcounts.py:315-488) resolves the counts
of a window by the allele its read pairs carry at a variant: host code like the reference's -- it walks a few
hundred reads per variant -- over the records the library's reader hands out whole (`fpt_bam_read_raw`), with the
reference's rules for a usable base call (quality >= 20, more than 3 bases from the read's 5' end, at most one
mismatch for the reference allele and two for the alternate by the XM / NM tag) and for a pair (both mates agree).

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


class ReadError(Exception):
    ERROR_ALIGNMENT = (0, "Read alignment problematic (QC fail, duplicate, or MAPQ < 1)")
    ERROR_5PROXIMITY = (1, "Variant too close to 5' end of tag")
    ERROR_BASEQ = (2, "Base quality < 20")
    ERROR_GENOTYPE = (3, "Base does not match reference or expected alternate allele")
    ERROR_MISMATCH = (4, "Read contains too many mismatches")

    def __init__(self, e):
        self.value = e[0]
        self.message = e[1]


class GenotypeError(Exception):
    pass


class ReadFormatError(Exception):
    pass


_SEQ_CODES = "=ACMGRSVTWYHKDBN"


class aligned_read(object):
    """One alignment record (SAM/BAM specification 4.2) with the members of pysam.AlignedSegment that
    cutcounts.py:119-248, 315-385 read: flags, reference_start / reference_end, mapping_quality, query_name,
    query_sequence, query_qualities, template_length, reference_id, has_tag / get_tag for integer tags."""

    __slots__ = ("reference_id", "reference_start", "reference_end", "mapping_quality", "flag", "query_name",
                 "next_reference_id", "next_reference_start", "template_length", "query_sequence", "query_qualities",
                 "_tags", "reference_name")

    def __init__(self, rec, ref_names=None):
        import struct
        (rid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, nrid, npos, tlen) = struct.unpack_from("<iiBBHHHiiii", rec, 0)
        self.reference_id, self.reference_start, self.mapping_quality, self.flag = rid, pos, mapq, flag
        self.next_reference_id, self.next_reference_start, self.template_length = nrid, npos, tlen
        at = 32
        self.query_name = bytes(rec[at:at + max(l_name - 1, 0)]).decode("ascii", "replace")
        at += l_name
        span = 0
        for k in range(n_cig):
            v = struct.unpack_from("<I", rec, at + 4 * k)[0]
            if (v & 0xf) in (0, 2, 3, 7, 8):  # M D N = X consume the reference
                span += v >> 4
        self.reference_end = pos + span if span > 0 else None  # (pysam: None without such an operation)
        at += 4 * n_cig
        packed = rec[at:at + (l_seq + 1) // 2]
        seq = []
        for b in packed:
            seq.append(_SEQ_CODES[b >> 4])
            seq.append(_SEQ_CODES[b & 0xf])
        self.query_sequence = "".join(seq[:l_seq]) if l_seq else None
        at += (l_seq + 1) // 2
        q = rec[at:at + l_seq]
        self.query_qualities = None if (l_seq == 0 or (len(q) and q[0] == 0xff)) else list(q)
        at += l_seq
        self._tags = {}
        n = len(rec)
        sizes = {"c": ("<b", 1), "C": ("<B", 1), "s": ("<h", 2), "S": ("<H", 2), "i": ("<i", 4), "I": ("<I", 4), "f": ("<f", 4), "A": (None, 1)}
        while at + 3 <= n:  # the optional fields: tag (2), type (1), value
            tag, ty = bytes(rec[at:at + 2]).decode("ascii", "replace"), chr(rec[at + 2])
            at += 3
            if ty in sizes:
                fmt, w = sizes[ty]
                if at + w > n:
                    break
                self._tags[tag] = chr(rec[at]) if fmt is None else struct.unpack_from(fmt, rec, at)[0]
                at += w
            elif ty in "ZH":
                end = at
                while end < n and rec[end] != 0:
                    end += 1
                self._tags[tag] = bytes(rec[at:end]).decode("ascii", "replace")
                at = end + 1
            elif ty == "B":
                if at + 5 > n:
                    break
                sub, cnt = chr(rec[at]), struct.unpack_from("<i", rec, at + 1)[0]
                at += 5 + cnt * sizes.get(sub, (None, 1))[1]
            else:
                break
        self.reference_name = ref_names[rid] if (ref_names is not None and 0 <= rid < len(ref_names)) else None

    # the flag bits pysam names (SAM specification 1.4)
    is_paired = property(lambda self: bool(self.flag & 0x1))
    is_proper_pair = property(lambda self: bool(self.flag & 0x2))
    is_unmapped = property(lambda self: bool(self.flag & 0x4))
    is_reverse = property(lambda self: bool(self.flag & 0x10))
    is_read1 = property(lambda self: bool(self.flag & 0x40))
    is_read2 = property(lambda self: bool(self.flag & 0x80))
    is_secondary = property(lambda self: bool(self.flag & 0x100))
    is_qcfail = property(lambda self: bool(self.flag & 0x200))
    is_duplicate = property(lambda self: bool(self.flag & 0x400))
    is_supplementary = property(lambda self: bool(self.flag & 0x800))

    def has_tag(self, tag):
        return tag in self._tags

    def get_tag(self, tag, with_value_type=False):
        return self._tags[tag]


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
    L.fpt_bam_read_raw.argtypes = [vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)]
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

    # ---- the allelically resolved counts (cutcounts.py:119-248, 315-488): host code, like the reference's
    def fetch(self, chrom, start, end):
        """the alignments that overlap [start, end) of `chrom`, in file order, as `aligned_read`s -- what
        `samfile.fetch(chrom, start, end)` yields (cutcounts.py:191).  Through the BAI index when there is one,
        else in one pass over the file."""
        import struct
        L = _bind(_lib.load())
        rid = self._ref_index.get(chrom, -1)
        if rid < 0:
            return
        names = [n for n, _ in self.references]
        h = C.c_void_p()
        _lib.check(L.fpt_bam_open(str(self.filepath).encode(), C.byref(h)))
        try:
            indexed = C.c_int32(0)
            _lib.check(L.fpt_bam_has_index(h, C.byref(indexed)))
            if indexed.value:
                _lib.check(L.fpt_bam_seek_region(h, rid, max(int(start), 0), int(end)))
            cap = 1 << 20
            buf = (C.c_uint8 * cap)()
            n, used = C.c_int64(), C.c_int64()
            while True:
                _lib.check(L.fpt_bam_read_raw(h, 4096, buf, cap, C.byref(n), C.byref(used)))
                if n.value == 0:
                    break
                raw = C.string_at(buf, used.value)
                at = 0
                for _ in range(n.value):
                    block = struct.unpack_from("<i", raw, at)[0]
                    read = aligned_read(raw[at + 4:at + 4 + block], names)
                    at += 4 + block
                    if read.reference_id != rid or read.is_unmapped:
                        continue
                    if read.reference_start >= end:
                        if indexed.value:
                            return
                        continue
                    r_end = read.reference_end if read.reference_end is not None else read.reference_start + 1
                    if r_end > start:
                        yield read
        finally:
            L.fpt_bam_close(h)

    def validate_read(self, read):
        """cutcounts.py:119-145"""
        if self.remove_qcfail and read.is_qcfail:
            raise ReadError(ReadError.ERROR_ALIGNMENT)
        if self.remove_dups and read.is_duplicate:
            raise ReadError(ReadError.ERROR_ALIGNMENT)
        if read.mapping_quality < self.min_qual:
            raise ReadError(ReadError.ERROR_ALIGNMENT)
        return read

    def read_pair_generator(self, chrom, start, end):
        """(read1, read2) of every usable pair with an alignment in [start - 10, end + 10) -- either may be
        None: single-end data, or a mate outside the window (cutcounts.py:170-229)"""
        waiting = {}
        for read in self.fetch(chrom, max(start - 10, 0), end + 10):
            try:
                self.validate_read(read)
            except ReadError:
                continue
            if not read.is_paired:
                yield read, None
                continue
            if not read.is_proper_pair or read.is_secondary or read.is_supplementary:
                continue
            name = read.query_name
            if name not in waiting:
                waiting[name] = [read, None] if read.is_read1 else [None, read]
            else:
                first, second = waiting.pop(name)
                yield (read, second) if read.is_read1 else (first, read)
        for first, second in waiting.values():  # pairs whose mate lies outside the window
            yield first, second

    def _add_read(self, read, fw, rev):
        """cutcounts.py:231-248"""
        if read.is_reverse:
            a = int(read.reference_end) + self.offset[1]
            rev[a] = rev.get(a, 0.0) + 1.0
        else:
            a = int(read.reference_start) + self.offset[0]
            fw[a] = fw.get(a, 0.0) + 1.0

    def _get_fragment(self, read):
        """the fragment of a mapped read, from the 5' ends of the pair (cutcounts.py:250-272)"""
        from .intervals import genomic_interval
        tlen = read.template_length
        if read.is_reverse:
            end = int(read.reference_end) + self.offset[1]
            start = end + tlen
        else:
            start = int(read.reference_start) + self.offset[0]
            end = start + tlen
        return genomic_interval(read.reference_name, start, end)

    def _validate_genotype(self, read, pos, ref, alt):
        """the allele a read carries at `pos` (0-based): `ref`, `alt`, or None where the read does not cover
        the position; ReadError for a base call that cannot be used (cutcounts.py:315-385)"""
        if not read:
            return None
        try:
            offset = pos - read.reference_start
            if offset < 0:
                raise IndexError
            base_call = read.query_sequence[offset]
            qual = read.query_qualities[offset]
            if qual < 20:
                raise ReadError(ReadError.ERROR_BASEQ)
            offset_5p = read.reference_end - 1 - pos if read.is_reverse else pos - read.reference_start
            if offset_5p <= 3:
                raise ReadError(ReadError.ERROR_5PROXIMITY)
            if read.has_tag("XM"):
                tag = "XM"
            elif read.has_tag("NM"):
                tag = "NM"
            else:
                raise ReadFormatError("No mismatch tag in read! (must contain XM or NM tag)")
            mm = int(read.get_tag(tag, with_value_type=False))
            if base_call == ref:
                if mm > 1:
                    raise ReadError(ReadError.ERROR_MISMATCH)
                return ref
            if base_call == alt:
                if mm > 2:
                    raise ReadError(ReadError.ERROR_MISMATCH)
                return alt
            raise ReadError(ReadError.ERROR_GENOTYPE)
        except IndexError:  # the variant lies outside the read
            return None

    def lookup_allelic(self, chrom, start, end, pos, ref, alt, flip=False):
        """Counts of [start, end) resolved by the allele the read pairs carry at the variant `pos` (0-based):
        {ref: {...}, alt: {...}, 'other': {...}}, each {'+': array, '-': array, 'fragments': [...]} -- 'other'
        holds the pairs that do not cover the variant; a pair with an unusable base call or discordant mates is
        left out (cutcounts.py:387-488)."""
        import logging
        counts = {k: ({}, {}, []) for k in ("ref", "alt", "non")}
        for read1, read2 in self.read_pair_generator(chrom, max(start - 10, 0), end + 10):
            try:
                g1 = self._validate_genotype(read1, pos, ref, alt)
                g2 = self._validate_genotype(read2, pos, ref, alt)
                if (not g1) and (not g2):
                    fw, rev, reads = counts["non"]
                elif (g1 and g2) and g1 != g2:
                    raise ReadError(ReadError.ERROR_GENOTYPE)
                elif g1 == ref or g2 == ref:
                    fw, rev, reads = counts["ref"]
                elif g1 == alt or g2 == alt:
                    fw, rev, reads = counts["alt"]
                else:
                    raise ReadError(ReadError.ERROR_GENOTYPE)
                if read1:
                    self._add_read(read1, fw, rev)
                if read2:
                    self._add_read(read2, fw, rev)
                reads.append(self._get_fragment(read1 if read1 else read2))
            except ReadError as e:
                logging.debug(e)
                continue

        def arrays(fw, rev, reads):
            f = np.array([fw.get(i, 0.0) for i in range(start, end)])
            r = np.array([rev.get(i, 0.0) for i in range(start, end)])
            return {"+": r[::-1] if flip else f, "-": f[::-1] if flip else r, "fragments": reads}

        return {ref: arrays(*counts["ref"]), alt: arrays(*counts["alt"]), "other": arrays(*counts["non"])}

    def __getitem__(self, x):
        """intervals only: the reference's dispatch hands a `pysam.VariantRecord` to `lookup_allelic` as ONE
        argument (cutcounts.py:513-515), which raises TypeError there -- call `lookup_allelic` directly"""
        if not hasattr(x, "chrom") or not hasattr(x, "start") or not hasattr(x, "end"):
            raise TypeError("Query type not supported: %s" % type(x))
        return self.lookup(x)
