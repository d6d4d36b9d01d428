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


def write_bam(path, references, reads, block_bytes=3000, index=False):
    """references: [(name, length)]; reads: dicts with ref (index), pos, cigar ("50M2D10M"), flag, mapq
    and optionally name, next_ref / next_pos / tlen, seq (bases), qual (one value or a list) and tags ({"NM": 1}).  index=True also writes <path>.bai (reads must be sorted by (ref, pos)): the
    linear index of the SAM specification 5.2, and per reference one bin holding one chunk that covers
    all of its alignments -- a reader that walks the bins finds every alignment, just not quickly."""
    rec_at = []  # (uncompressed offset of the record, ref, pos, end)
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
        if "seq" in r:  # bases (4 bits each: "=ACMGRSVTWYHKDBN"), qualities, integer tags
            seq = r["seq"]
            assert len(seq) == l_seq, "seq does not match the CIGAR"
            codes = ["=ACMGRSVTWYHKDBN".index(ch) for ch in seq] + [0]
            rec += bytes((codes[2 * i] << 4) | codes[2 * i + 1] for i in range((l_seq + 1) // 2))
            q = r.get("qual", 40)
            rec += bytes([q] * l_seq) if isinstance(q, int) else bytes(q)
        else:
            rec += b"\x11" * ((l_seq + 1) // 2) + b"\x28" * l_seq
        for tag, val in (r.get("tags") or {}).items():
            rec += tag.encode() + (b"C" + struct.pack("<B", val) if 0 <= val < 256 else b"i" + struct.pack("<i", val))
        span = sum(v >> 4 for v in ops if (v & 0xf) in (0, 2, 3, 7, 8))
        rec_at.append((len(out), r["ref"], r["pos"], r["pos"] + max(span, 1)))
        out += struct.pack("<i", len(rec)) + rec
    block_at = []  # compressed offset of every block
    with open(path, "wb") as f:
        for a in range(0, len(out), block_bytes):  # records deliberately straddle block boundaries
            block_at.append(f.tell())
            f.write(_bgzf_block(bytes(out[a:a + block_bytes])))
        end_at = f.tell()
        f.write(_bgzf_block(b""))  # EOF marker block
    if index:
        def voff(u):
            return (block_at[u // block_bytes] << 16) | (u % block_bytes)
        bai = bytearray(b"BAI\1") + struct.pack("<i", len(references))
        for ri in range(len(references)):
            mine = [(voff(u), pos, end) for u, ref, pos, end in rec_at if ref == ri and pos >= 0]
            if not mine:
                bai += struct.pack("<ii", 0, 0)
                continue
            lin = [0] * ((max(e for _, _, e in mine) - 1 >> 14) + 1)
            for v, pos, end in mine:
                for w in range(pos >> 14, (end - 1 >> 14) + 1):
                    lin[w] = v if lin[w] == 0 else min(lin[w], v)
            last = max(u for u, ref, _, _ in rec_at if ref == ri)
            nxt = min([u for u, _, _, _ in rec_at if u > last] or [len(out)])
            chunk_end = voff(nxt) if nxt < len(out) else (end_at << 16)
            bai += struct.pack("<i", 1) + struct.pack("<Ii", 0, 1) + struct.pack("<QQ", min(v for v, _, _ in mine), chunk_end)
            bai += struct.pack("<i", len(lin)) + b"".join(struct.pack("<Q", v) for v in lin)
        bai += struct.pack("<Q", 0)  # alignments without coordinates
        with open(path + ".bai", "wb") as f:
            f.write(bytes(bai))
