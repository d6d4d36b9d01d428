"""Test helper: bgzip a bedGraph text into BGZF blocks cut at arbitrary byte positions (lines straddle
blocks) and write a tabix index (.tbi) for it -- the TBI layout of the tabix manual: header (format
0x10000 = zero-based half-open / the BED preset, columns 1-2-3, meta '#'), names, and per reference the
binning index plus the linear index of 16 kb windows (virtual offset of the first overlapping record).
So that the library's reader can be tested on indexed files without htslib."""
import struct

from .bamwriter import _bgzf_block


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def write_bgzf_with_tbi(path, text, block_bytes=3000, tbi=True):
    """text: bytes of a position-sorted bedGraph (chrom, start, end, ...).  Writes `path` and `path.tbi`."""
    cuts = list(range(0, len(text), block_bytes)) + [len(text)]
    blocks = [_bgzf_block(text[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    coff = [0]
    for blk in blocks:
        coff.append(coff[-1] + len(blk))
    with open(path, "wb") as f:
        for blk in blocks:
            f.write(blk)
        f.write(_bgzf_block(b""))
    if not tbi:
        return

    def voff(byte_pos):  # virtual offset of an uncompressed byte position
        k = min(byte_pos // block_bytes, len(blocks) - 1) if blocks else 0
        return (coff[k] << 16) | (byte_pos - k * block_bytes)
    names, bins, lin = [], {}, {}
    pos = 0
    for line in text.split(b"\n"):
        if line and not line.startswith(b"#"):
            f_ = line.split(b"\t")
            chrom, beg, end = f_[0].decode(), int(f_[1]), int(f_[2])
            if chrom not in bins:
                names.append(chrom)
                bins[chrom], lin[chrom] = {}, []
            v0, v1 = voff(pos), voff(pos + len(line) + 1)
            b = bins[chrom].setdefault(reg2bin(beg, end), [])
            if b and b[-1][1] == v0:
                b[-1][1] = v1
            else:
                b.append([v0, v1])
            for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                while len(lin[chrom]) <= w:
                    lin[chrom].append(None)
                if lin[chrom][w] is None:
                    lin[chrom][w] = v0
        pos += len(line) + 1
    out = bytearray(b"TBI\1")
    nm = b"".join(n.encode() + b"\0" for n in names)
    out += struct.pack("<8i", len(names), 0x10000, 1, 2, 3, ord("#"), 0, len(nm)) + nm
    for n in names:
        out += struct.pack("<i", len(bins[n]))
        for b, chunks in sorted(bins[n].items()):
            out += struct.pack("<Ii", b, len(chunks))
            for v0, v1 in chunks:
                out += struct.pack("<QQ", v0, v1)
        # htslib fills the windows without a record with the previous window's offset (0 before the first)
        filled, prev = [], 0
        for v in lin[n]:
            prev = v if v is not None else prev
            filled.append(prev)
        out += struct.pack("<i", len(filled)) + b"".join(struct.pack("<Q", v) for v in filled)
    with open(path + ".tbi", "wb") as f:
        for a in range(0, len(out), 40000):
            f.write(_bgzf_block(bytes(out[a:a + 40000])))
        f.write(_bgzf_block(b""))
