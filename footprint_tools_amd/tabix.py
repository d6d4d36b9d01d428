"""Per-nucleotide statistics tracks written by `ftd detect` (bgzip-compressed bedGraph, indexed
with tabix in the reference's workflow) -- read access without pysam / htslib.

The reference opens them with `pysam.TabixFile(fn)` and walks `fetch(chrom, start, end,
parser=pysam.asTuple())` per interval (cli/post.py:52-87).  BGZF is a chain of gzip members, which
the standard library inflates; the track is parsed once into columns and `fetch` answers from
memory by binary search (rows of a track are sorted by position within a chromosome).  This holds
a whole track in memory: meant for the hotspot-restricted tracks the posterior caller reads, not
for genome-wide dumps.  Parity of the reader is unpinned (no pysam here); the column meaning is
the reference's writer's (cli/utils.py:119-144: chrom, start, start+1, exp, obs, -log p,
-log win-p, fdr).
"""
import gzip
import io

import numpy as np


class TabixFile(object):
    def __init__(self, filename):
        self.filename = filename
        self._by_chrom = None
        with open(filename, "rb") as f:  # fail now, like pysam does, if the file is not there
            magic = f.read(2)
        self._gz = magic == b"\x1f\x8b"

    def _load(self):
        import pandas as pd
        opener = gzip.open if self._gz else open
        with opener(self.filename, "rb") as f:
            text = f.read()
        tab = pd.read_csv(io.BytesIO(text), sep="\t", header=None, comment="#", dtype={0: str})
        self._by_chrom = {}
        for chrom, part in tab.groupby(0, sort=False):
            part = part.sort_values(1, kind="stable")
            self._by_chrom[chrom] = (part[1].to_numpy(np.int64), part.iloc[:, 1:].to_numpy(np.float64))

    def fetch_columns(self, chrom, start, end):
        """(positions, values): rows with start <= position < end; values[:, k] is file column k+1."""
        if self._by_chrom is None:
            self._load()
        if chrom not in self._by_chrom:
            return np.zeros(0, np.int64), np.zeros((0, 0))
        pos, vals = self._by_chrom[chrom]
        a, b = np.searchsorted(pos, start, "left"), np.searchsorted(pos, end, "left")
        return pos[a:b], vals[a:b]

    def fetch(self, chrom, start, end, parser=None):
        """rows as tuples of strings, like pysam's asTuple parser"""
        pos, vals = self.fetch_columns(chrom, start, end)
        for p, row in zip(pos, vals):
            yield (chrom,) + tuple(repr(int(v)) if k < 2 else repr(float(v)) for k, v in enumerate(row))

    def close(self):
        self._by_chrom = None
