"""Test helper: write a small BAM file (BGZF blocks with the BC extra field and the EOF block, SAM/BAM
specification section 4) from a list of reads, so that the library's reader can be tested without
pysam / htslib."""
import struct
import zlib

CIGAR_OPS = "MIDNSHP=X"


def _bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = comp.compress(data) + comp.flush()
    bsize = len(body) + 25  # header 18 + body + crc/isize 8 - 1
    head = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def write_bam(path, references, reads, block_bytes=3000):
    """references: [(name, length)]; reads: dicts with ref (index), pos, cigar ("50M2D10M"), flag, mapq
    and optionally name."""
    out = bytearray(b"BAM\1")
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in references)
    out += struct.pack("<i", len(text)) + text.encode()
    out += struct.pack("<i", len(references))
    for name, length in references:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length)
    for k, r in enumerate(reads):
        name = (r.get("name") or "r%d" % k).encode() + b"\0"
        ops, num = [], ""
        for ch in r["cigar"]:
            if ch.isdigit():
                num += ch
            else:
                ops.append((int(num) << 4) | CIGAR_OPS.index(ch))
                num = ""
        l_seq = sum(v >> 4 for v in ops if (v & 0xf) in (0, 1, 4, 7, 8))
        rec = struct.pack("<iiBBHHHiiii", r["ref"], r["pos"], len(name), r["mapq"], 4680, len(ops), r["flag"], l_seq,
                          r.get("next_ref", -1), r.get("next_pos", -1), r.get("tlen", 0))
        rec += name + b"".join(struct.pack("<I", v) for v in ops)
        rec += b"\x11" * ((l_seq + 1) // 2) + b"\x28" * l_seq
        out += struct.pack("<i", len(rec)) + rec
    with open(path, "wb") as f:
        for a in range(0, len(out), block_bytes):  # records deliberately straddle block boundaries
            f.write(_bgzf_block(bytes(out[a:a + block_bytes])))
        f.write(_bgzf_block(b""))  # EOF marker block
