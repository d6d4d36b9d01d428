"""Genomic intervals for the drivers: the duck type the path needs (`chrom`, `start`, `end`,
`widen(n)` -> copy, `len()`; modeling/predict.pyx:132-140 and cli/detect.py:118 build it as
`genome_tools.genomic_interval(chrom, start, end)`, a package that is not part of the reference
tree and not in this image) and a reader of interval files the way cli/detect.py:50 reads them
(`pd.read_table(interval_file, header=None)`: tab-separated, no header, columns chrom / start / end
and whatever follows).
"""
import gzip


class genomic_interval(object):
    """chrom:[start, end) with optional name / score / strand (BED columns 4-6).  Strand '-' makes
    cut-count lookups return mirrored and swapped arrays (cutcounts.py:307-311)."""

    __slots__ = ("chrom", "start", "end", "name", "score", "strand")

    def __init__(self, chrom, start, end, name=".", score=None, strand=None):
        self.chrom, self.start, self.end = str(chrom), int(start), int(end)
        self.name, self.score, self.strand = name, score, strand

    def __len__(self):
        return self.end - self.start

    def __str__(self):
        return "\t".join([self.chrom, str(self.start), str(self.end)])

    def __repr__(self):
        return "genomic_interval(%r, %d, %d)" % (self.chrom, self.start, self.end)

    def __eq__(self, other):
        return (self.chrom, self.start, self.end, self.strand) == (other.chrom, other.start, other.end, other.strand)

    def __hash__(self):
        return hash((self.chrom, self.start, self.end, self.strand))

    def widen(self, w):
        """a copy widened by w bases on both sides (the original is left alone: predict.pyx:132)"""
        return genomic_interval(self.chrom, self.start - int(w), self.end + int(w), self.name, self.score, self.strand)

    def shift(self, x):
        return genomic_interval(self.chrom, self.start + int(x), self.end + int(x), self.name, self.score, self.strand)


def read_intervals(path):
    """The intervals of a BED-like file (plain or gzip / bgzip): tab-separated, no header, chrom /
    start / end, then optionally name, score, strand.  Lines starting with '#', 'track' or
    'browser' and empty lines are skipped.  Raises ValueError naming the line that does not parse."""
    out = []
    opener = gzip.open if str(path).endswith((".gz", ".bgz")) else open
    with opener(path, "rt") as f:
        for n, line in enumerate(f, 1):
            line = line.rstrip("\r\n")
            if not line or line[0] == "#" or line.startswith(("track", "browser")):
                continue
            c = line.split("\t")
            try:
                iv = genomic_interval(c[0], int(c[1]), int(c[2]), c[3] if len(c) > 3 else ".",
                                      c[4] if len(c) > 4 else None, c[5] if len(c) > 5 and c[5] in "+-" else None)
            except (IndexError, ValueError):
                raise ValueError("%s line %d: not <chrom> TAB <start> TAB <end> ..." % (path, n))
            if iv.end < iv.start or iv.start < 0:
                raise ValueError("%s line %d: start / end out of order" % (path, n))
            out.append(iv)
    return out


class interval_columns(object):
    """An interval list as columns (distinct chromosome names, an id per interval, starts, ends,
    strand '-' flags): read off the objects once, so that a batch is a slice of arrays instead of
    a dozen passes over Python objects (~2 us per interval and batch otherwise -- more than the
    statistics of a 160-base interval cost on the device)."""

    def __init__(self, names, cid, start, end, flip):
        self.names, self.cid, self.start, self.end, self.flip = names, cid, start, end, flip

    @classmethod
    def of(cls, intervals):
        import numpy as np
        n = len(intervals)
        uniq = {}
        cid = np.fromiter((uniq.setdefault(iv.chrom, len(uniq)) for iv in intervals), dtype=np.int32, count=n)
        start = np.fromiter((iv.start for iv in intervals), dtype=np.int64, count=n)
        end = np.fromiter((iv.end for iv in intervals), dtype=np.int64, count=n)
        flip = np.fromiter((getattr(iv, "strand", None) == "-" for iv in intervals), dtype=bool, count=n)
        return cls(list(uniq), cid, start, end, flip)

    def __len__(self):
        return self.cid.size

    def take(self, indices):
        """the rows `indices` (a range of step 1 is a slice)"""
        import numpy as np
        if isinstance(indices, range) and indices.step == 1:
            sl = slice(indices.start, indices.stop)
        else:
            sl = np.asarray(indices, dtype=np.int64)
        return interval_columns(self.names, self.cid[sl], self.start[sl], self.end[sl], self.flip[sl])

    def chroms(self):
        names = self.names
        return [names[c] for c in self.cid.tolist()]

    def lookup(self, table, missing):
        """per interval: table[chrom] (rows of an int64 array), `missing` for names it does not have"""
        import numpy as np
        rows = np.array([table.get(c, missing) for c in self.names], dtype=np.int64)
        return rows[self.cid] if len(self.names) else rows.reshape((0,) + np.shape(missing))
