"""Reference-sequence access for the scan: what `pysam.FastaFile(path).fetch(chrom, start, end)`
gives the reference (cli/detect.py:100-110, modeling/predict.pyx:136-140), without htslib.

`FastaFile.fetch` returns the bases of [start, end) as `str`; `fetch_batch` returns the ASCII bytes
of a whole interval list in the fused scan's CSR layout (L + 2*pad + 7 bases per interval: the
padded interval plus 3 bases either side for the 6-mer context, predict.pyx:136-140);
`fetch_batch_dev` does the same gather ON THE DEVICE from a copy of the file kept there (uploaded
once: a genome is a few GB of 288), so a batch costs one small upload of interval descriptors
instead of a host-side gather and an upload of the bytes.  Positions
outside the chromosome read as 'N' (pysam truncates; the scan then uses the default propensity,
as `kmer_model.__getitem__` does for any 6-mer it does not know, bias.py:16-17).

Uses the .fai index next to the file when there is one, else builds the same table by one pass.
"""
import mmap
import os

import numpy as np


class FastaFile(object):
    def __init__(self, filepath):
        self.filepath = filepath
        self._f = open(filepath, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if os.path.getsize(filepath) else b""
        self.index = {}  # name -> (length, byte offset of the first base, bases per line, bytes per line)
        fai = filepath + ".fai"
        if os.path.exists(fai):
            for line in open(fai):
                f = line.rstrip("\n").split("\t")
                if len(f) >= 5:
                    self.index[f[0]] = (int(f[1]), int(f[2]), int(f[3]), int(f[4]))
        else:
            self._build_index()
        self.references = list(self.index)
        self.lengths = [self.index[r][0] for r in self.references]

    def _build_index(self):
        name, length, offset, lb, lw, pos = None, 0, 0, 0, 0, 0
        for raw in iter(self._f.readline, b""):
            if raw.startswith(b">"):
                if name is not None:
                    self.index[name] = (length, offset, lb, lw)
                name = raw[1:].split()[0].decode() if raw[1:].split() else ""
                length, lb, lw, offset = 0, 0, 0, pos + len(raw)
            elif name is not None:
                body = raw.rstrip(b"\r\n")
                if lb == 0 and body:
                    lb, lw = len(body), len(raw)
                length += len(body)
            pos += len(raw)
        if name is not None:
            self.index[name] = (length, offset, lb, lw)
        self._f.seek(0)

    def close(self):
        if self._mm:
            self._mm.close()
        self._f.close()

    def _bytes(self, chrom, start, end):
        """uint8 array of [start, end), 'N' outside the chromosome (or for an unknown chromosome)"""
        n = max(end - start, 0)
        out = np.full(n, ord("N"), dtype=np.uint8)
        if chrom not in self.index or n == 0:
            return out
        length, offset, lb, lw = self.index[chrom]
        a, b = max(start, 0), min(end, length)
        if a >= b or lb == 0:
            return out
        first = offset + (a // lb) * lw + a % lb
        last = offset + ((b - 1) // lb) * lw + (b - 1) % lb + 1
        raw = np.frombuffer(self._mm, dtype=np.uint8, count=last - first, offset=first)
        col = (np.arange(last - first) + (first - offset)) % lw  # position within the line, newline bytes >= lb
        out[a - start:b - start] = raw[col < lb]
        return out

    def fetch(self, chrom, start, end):
        return self._bytes(chrom, int(start), int(end)).tobytes().decode("ascii", "replace")

    def fetch_batch(self, intervals, pad, context=3):
        """ASCII bytes of every interval's [start - pad - 1 - context, end + pad + context), back to
        back: the `seq` array of FootprintScanner.scan / scan_dev.  One vectorised gather from the
        mapped file for the whole list (no work per interval in Python)."""
        ivs = list(intervals)
        if not ivs:
            return np.zeros(0, np.uint8)
        starts = np.array([iv.start for iv in ivs], dtype=np.int64) - (pad + 1 + context)
        ends = np.array([iv.end for iv in ivs], dtype=np.int64) + (pad + context)
        par = np.array([self.index.get(iv.chrom, (0, 0, 0, 1)) for iv in ivs], dtype=np.int64)  # length, offset, lb, lw
        n = np.maximum(ends - starts, 0)
        off = np.concatenate([[0], np.cumsum(n)])
        total = int(off[-1])
        which = np.repeat(np.arange(len(ivs)), n)
        g = np.arange(total, dtype=np.int64) - off[:-1][which] + starts[which]  # genomic position of every byte
        length, offset, lb, lw = (par[which, k] for k in range(4))
        inside = (g >= 0) & (g < length) & (lb > 0)
        gi = np.where(inside, g, 0)
        lbs = np.where(lb > 0, lb, 1)
        pos = offset + (gi // lbs) * lw + gi % lbs
        out = np.full(total, ord("N"), dtype=np.uint8)
        if inside.any():
            data = np.frombuffer(self._mm, dtype=np.uint8)
            out[inside] = data[pos[inside]]
        return out

    # ---- the same gather on the device -------------------------------------------------------------
    def to_device(self, ctx):
        """Keep the file's bytes on the device of `ctx` (once)."""
        from . import _lib  # noqa: F401
        from .scan import DeviceArray
        if getattr(self, "_dev", None) is None or self._dev_ctx is not ctx:
            data = np.frombuffer(self._mm, dtype=np.uint8) if self._mm else np.zeros(0, np.uint8)
            self._dev = DeviceArray(ctx, max(data.size, 16))
            if data.size:
                self._dev.upload(data)
            self._dev_bytes, self._dev_ctx = int(data.size), ctx
        return self._dev

    def fetch_batch_dev(self, ctx, intervals, pad, context=3):
        """`fetch_batch` without the host gather: a DeviceArray holding the `seq` bytes of the
        interval list in the scan's layout (fpt_seq_gather_dev)."""
        from . import _lib
        from .scan import DeviceArray
        fa = self.to_device(ctx)
        if hasattr(intervals, "cid"):  # intervals.interval_columns
            ivs = intervals
            starts, n = ivs.start - (pad + 1 + context), ivs.end - ivs.start + (2 * pad + 1 + 2 * context)
            where = ivs.lookup(self.index, (0, 0, 0, 1))
        else:
            ivs = list(intervals)
            starts = np.array([iv.start for iv in ivs], dtype=np.int64) - (pad + 1 + context)
            n = np.array([iv.end - iv.start for iv in ivs], dtype=np.int64) + (2 * pad + 1 + 2 * context)
            where = np.array([self.index.get(iv.chrom, (0, 0, 0, 1)) for iv in ivs], dtype=np.int64).reshape(len(ivs), 4)
        off = np.concatenate([[0], np.cumsum(n)])
        desc = np.empty((len(ivs), 7), dtype=np.int64)
        desc[:, 0], desc[:, 1], desc[:, 2] = starts, n, off[:-1]
        desc[:, 3:] = where
        out = DeviceArray(ctx, max(int(off[-1]), 16))
        if len(ivs):
            d_desc = DeviceArray(ctx, desc.nbytes).upload(desc)
            _lib.check(ctx.L.fpt_seq_gather_dev(ctx.h, fa.ptr, self._dev_bytes, d_desc.ptr, len(ivs), out.ptr))
            ctx.synchronize()
            d_desc.free()
        return out, int(off[-1])
