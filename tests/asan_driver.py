"""Runs under LD_PRELOAD=libasan.so (tests/test_ingest_cpu.py::test_sanitizer_build starts it): the
host-only sanitizer builds of the BGZF / BAM reader and the text formatter
(footprint_tools_amd/libfpt_host_asan.so, `make -C footprint_tools_amd/csrc asan`) and of the CPU
checker (oracle/libfpt_oracle_asan.so) on good input, and the reader on a corpus of damaged files:
every damaged file must end in an error code, never in a sanitizer report.  Prints "ASAN-DRIVER OK"."""
import ctypes as C
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.bamwriter import _bgzf_block, write_bam  # noqa: E402
from tests.tbiwriter import write_bgzf_with_tbi  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "footprint_tools_amd", "libfpt_host_asan.so"))
vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
L.fpt_last_error.restype = C.c_char_p
L.fpt_bam_open.argtypes = [C.c_char_p, C.POINTER(vp)]
L.fpt_bam_close.argtypes = [vp]
L.fpt_bam_read.argtypes = [vp, i64, vp, vp, vp, vp, vp, C.POINTER(i64)]
L.fpt_bam_read_raw.argtypes = [vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)]
L.fpt_format_stats.argtypes = [C.c_char_p, i64, vp, i64, i32, vp, i64, C.c_char, i32, vp, i64, C.POINTER(i64)]
L.fpt_track_open.argtypes = [C.c_char_p, C.POINTER(vp)]
L.fpt_track_close.argtypes = [vp]
L.fpt_track_fetch_rows.argtypes = [vp, C.c_char_p, i64, i64, i32, vp, i64, vp, vp, C.POINTER(i64)]
L.fpt_track_fetch.argtypes = [vp, i64, vp, vp, vp, vp, i32, vp, vp, vp]


def track_rows(path, chrom, a, b):
    """(rc_open, rc_fetch, n_rows) of one region query"""
    h = vp()
    rc = L.fpt_track_open(path.encode(), C.byref(h))
    if rc:
        return rc, 0, 0
    cols = np.array([3, 4, 7], np.int32)
    cap = max(b - a, 1)
    pos, vals, n = np.empty(cap, np.int64), np.empty((cap, 3)), i64()
    rc2 = L.fpt_track_fetch_rows(h, chrom.encode(), a, b, 3, cols.ctypes.data, cap, pos.ctypes.data, vals.ctypes.data, C.byref(n))
    L.fpt_track_close(h)
    return 0, rc2, n.value


def read_all(path, batch=500):
    """(rc_open, rc_read, n_reads): reads until the end or the first error"""
    h = vp()
    rc = L.fpt_bam_open(path.encode(), C.byref(h))
    if rc:
        return rc, 0, 0
    total, rc2 = 0, 0
    while True:
        a = [np.empty(batch, np.int32) for _ in range(3)]
        fl, mq = np.empty(batch, np.uint16), np.empty(batch, np.uint8)
        got = i64()
        rc2 = L.fpt_bam_read(h, batch, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, fl.ctypes.data,
                             mq.ctypes.data, C.byref(got))
        total += got.value
        if rc2 or got.value == 0:
            break
    L.fpt_bam_close(h)
    # the same file once more through the whole-record reader, into a buffer that holds a few records at a time
    # (every record must come out whole: its block_size says how many bytes follow)
    h = vp()
    if L.fpt_bam_open(path.encode(), C.byref(h)) == 0:
        buf = (C.c_uint8 * 700)()
        raw_total = 0
        while True:
            got, used = i64(), i64()
            rc3 = L.fpt_bam_read_raw(h, batch, buf, 700, C.byref(got), C.byref(used))
            if rc3 or got.value == 0:
                break
            at = 0
            for _ in range(got.value):
                block = struct.unpack_from("<i", buf, at)[0]
                assert block >= 32 and at + 4 + block <= used.value
                at += 4 + block
            assert at == used.value
            raw_total += got.value
        if rc2 == 0 and rc3 == 0:
            assert raw_total == total, (raw_total, total)
        L.fpt_bam_close(h)
    return 0, rc2, total


def main(tmp):
    rs = np.random.RandomState(5)
    refs = [("chr1", 100000), ("chr2", 50000)]
    reads = [dict(ref=int(rs.randint(0, 2)), pos=int(rs.randint(0, 40000)), cigar=str(rs.choice(["36M", "20M2D16M", "5S31M"])),
                  flag=int(rs.choice([0, 16, 99, 147])), mapq=int(rs.choice([0, 30, 60]))) for _ in range(4000)]
    good = os.path.join(tmp, "good.bam")
    write_bam(good, refs, reads, block_bytes=2500)
    for threads in ("1", "4"):
        os.environ["FPT_BAM_THREADS"] = threads
        assert read_all(good) == (0, 0, len(reads)), read_all(good)
    raw = open(good, "rb").read()

    def case(name, data, expect_reads=None):
        p = os.path.join(tmp, name)
        with open(p, "wb") as f:
            f.write(data)
        rc_open, rc_read, n = read_all(p)
        failed = rc_open != 0 or rc_read != 0
        if expect_reads is None:
            assert failed, (name, rc_open, rc_read, n)
        else:
            assert not failed and n == expect_reads, (name, rc_open, rc_read, n)
        return L.fpt_last_error().decode()

    # ---- container damage
    case("empty.bam", b"")
    case("garbage.bam", b"not a bam file at all" * 10)
    for cut in (5, 17, 40, len(raw) // 3, len(raw) // 2, len(raw) - 30, len(raw) - 5):
        case("cut%d.bam" % cut, raw[:cut])
    for k in range(60):  # single bit flips all over the file: header fields, deflate payload, CRC, ISIZE
        pos = int(rs.randint(0, len(raw) - 28))  # (the empty EOF block at the end may be damaged harmlessly)
        b = bytearray(raw)
        b[pos] ^= 1 << int(rs.randint(0, 8))
        p = os.path.join(tmp, "flip.bam")
        with open(p, "wb") as f:
            f.write(bytes(b))
        read_all(p)  # error or not (a flip inside a read name changes nothing we look at): no sanitizer report
    b = bytearray(raw)
    b[16:18] = struct.pack("<H", 10)        # BSIZE smaller than the header it sits in
    case("bsize_small.bam", bytes(b))
    b = bytearray(raw)
    b[10:12] = struct.pack("<H", 60000)     # XLEN far beyond the member
    case("xlen_big.bam", bytes(b))
    case("no_bc.bam", zlib.compress(b"BAM\1" + b"\0" * 100))  # a gzip-less zlib stream / no BC subfield

    # ---- record damage inside well-formed BGZF blocks
    def bam_bytes(records, l_text=None, n_ref=None, l_name_ref=None):
        text = b"@HD\tVN:1.6\n"
        out = bytearray(b"BAM\1") + struct.pack("<i", len(text) if l_text is None else l_text) + text
        out += struct.pack("<i", 1 if n_ref is None else n_ref)
        out += struct.pack("<i", 5 if l_name_ref is None else l_name_ref) + b"chr1\0" + struct.pack("<i", 1000)
        for r in records:
            out += r
        return _bgzf_block(bytes(out)) + _bgzf_block(b"")

    def record(block_size=None, l_name=3, n_cig=1, cig=((36 << 4) | 0,), tail=b""):
        body = struct.pack("<iiBBHHHiiii", 0, 100, l_name, 30, 0, n_cig, 0, 36, -1, -1, 0) + b"r1\0"
        body += b"".join(struct.pack("<I", v) for v in cig) + tail
        return struct.pack("<i", len(body) if block_size is None else block_size) + body

    case("ok_one.bam", bam_bytes([record()]), expect_reads=1)
    case("neg_l_text.bam", bam_bytes([record()], l_text=-5))
    case("huge_l_text.bam", bam_bytes([record()], l_text=1 << 30))
    case("neg_n_ref.bam", bam_bytes([record()], n_ref=-1))
    case("many_refs.bam", bam_bytes([record()], n_ref=1000))
    case("bad_l_name_ref.bam", bam_bytes([record()], l_name_ref=1 << 28))
    case("zero_l_name_ref.bam", bam_bytes([record()], l_name_ref=0))
    case("small_block.bam", bam_bytes([record(block_size=12)]))
    case("neg_block.bam", bam_bytes([record(block_size=-1)]))
    case("huge_block.bam", bam_bytes([record(block_size=1 << 29)]))
    case("block_past_end.bam", bam_bytes([record(block_size=500)]))
    # a CIGAR count / name length that reaches past the record: the record is still walked, its end unknown
    case("n_cig_past.bam", bam_bytes([record(n_cig=60000)]), expect_reads=1)
    case("l_name_past.bam", bam_bytes([record(l_name=255)]), expect_reads=1)
    case("no_cigar.bam", bam_bytes([record(n_cig=0, cig=())]), expect_reads=1)

    # ---- region access through a BAI index: good index, then bit flips and cuts in it
    L.fpt_bam_seek_region.argtypes = [vp, i32, i64, i64]
    sreads = sorted(reads, key=lambda r: (r["ref"], r["pos"]))
    ibam = os.path.join(tmp, "idx.bam")
    write_bam(ibam, refs, sreads, block_bytes=2500, index=True)

    def region(path, ref_id, a, b):
        h = vp()
        rc = L.fpt_bam_open(path.encode(), C.byref(h))
        if rc:
            return rc, 0
        n = 0
        if L.fpt_bam_seek_region(h, ref_id, a, b) == 0:
            while True:
                arr = [np.empty(300, np.int32) for _ in range(3)]
                fl, mq = np.empty(300, np.uint16), np.empty(300, np.uint8)
                got = i64()
                if L.fpt_bam_read(h, 300, arr[0].ctypes.data, arr[1].ctypes.data, arr[2].ctypes.data, fl.ctypes.data,
                                  mq.ctypes.data, C.byref(got)) or got.value == 0:
                    break
                n += got.value
        L.fpt_bam_close(h)
        return 0, n

    for ref_id, a, b in ((0, 0, 40000), (1, 16384, 20000), (0, 39000, 90000), (1, 100, 101)):
        span = {"36M": 36, "20M2D16M": 38, "5S31M": 31}
        want = sum(1 for r in sreads if r["ref"] == ref_id and r["pos"] < b and r["pos"] + span[r["cigar"]] > a)
        rc, n = region(ibam, ref_id, a, b)
        assert rc == 0 and n >= want, (ref_id, a, b, n, want)
    braw = open(ibam + ".bai", "rb").read()
    import shutil
    dbam = os.path.join(tmp, "idx_damaged.bam")
    shutil.copy(ibam, dbam)
    for k in range(60):
        bb = bytearray(braw)
        bb[int(rs.randint(0, len(braw)))] ^= 1 << int(rs.randint(0, 8))
        with open(dbam + ".bai", "wb") as f:
            f.write(bytes(bb))
        region(dbam, int(rs.randint(0, 2)), int(rs.randint(0, 30000)), int(rs.randint(30000, 60000)))  # error or not
    for cut in (3, 9, 30, len(braw) // 2, len(braw) - 3):
        with open(dbam + ".bai", "wb") as f:
            f.write(braw[:cut])
        region(dbam, 0, 0, 1000)

    # ---- statistics tracks: region access with and without a tabix index, then damaged files
    lines = [b"#header"]
    tpos = np.sort(rs.choice(60000, 9000, replace=False))
    for p_ in tpos:
        lines.append(("chr1\t%d\t%d\t%.4f\t%.4f\t0.5\t0.5\t%.4f" % (p_, p_ + 1, rs.rand() * 9, rs.rand() * 9, rs.rand())).encode())
    ttext = b"\n".join(lines) + b"\n"
    for with_tbi in (True, False):
        tp = os.path.join(tmp, "track%d.gz" % with_tbi)
        write_bgzf_with_tbi(tp, ttext, block_bytes=2100, tbi=with_tbi)
        for a, b in ((0, 60000), (100, 101), (16384, 40000), (59990, 70000)):
            assert track_rows(tp, "chr1", a, b) == (0, 0, int(((tpos >= a) & (tpos < b)).sum())), (with_tbi, a, b)
        assert track_rows(tp, "chr9", 0, 100) == (0, 0, 0)
        traw = open(tp, "rb").read()
        for k in range(40):
            bb = bytearray(traw)
            bb[int(rs.randint(0, len(traw) - 28))] ^= 1 << int(rs.randint(0, 8))
            dp = os.path.join(tmp, "track_flip.gz")
            with open(dp, "wb") as f:
                f.write(bytes(bb))
            if with_tbi:
                import shutil
                shutil.copy(tp + ".tbi", dp + ".tbi")
            elif os.path.exists(dp + ".tbi"):
                os.remove(dp + ".tbi")
            track_rows(dp, "chr1", 0, 60000)  # an error or not: no sanitizer report
        for cut in (10, len(traw) // 3, len(traw) - 40):
            dp = os.path.join(tmp, "track_cut.gz")
            with open(dp, "wb") as f:
                f.write(traw[:cut])
            track_rows(dp, "chr1", 0, 60000)
    # a damaged index: bit flips in the .tbi
    tp = os.path.join(tmp, "track1.gz")
    iraw = open(tp + ".tbi", "rb").read()
    for k in range(40):
        bb = bytearray(iraw)
        bb[int(rs.randint(0, len(iraw)))] ^= 1 << int(rs.randint(0, 8))
        dp = os.path.join(tmp, "track_badidx.gz")
        import shutil
        shutil.copy(tp, dp)
        with open(dp + ".tbi", "wb") as f:
            f.write(bytes(bb))
        track_rows(dp, "chr1", 0, 60000)

    # ---- the track writer: text in pieces -> bgzip + tabix files, read back through the reader above; bad text
    L.fpt_track_writer_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fpt_track_writer_write.argtypes = [vp, C.c_char_p, i64]
    L.fpt_track_writer_close.argtypes = [vp]
    for threads in ("1", "6"):
        os.environ["FPT_TRACK_THREADS"] = threads
        wp = os.path.join(tmp, "written%s.gz" % threads)
        h = vp()
        assert L.fpt_track_writer_open(wp.encode(), C.byref(h)) == 0
        cuts = sorted(set([0, len(ttext)] + rs.randint(0, len(ttext), 25).tolist()))
        for a, b in zip(cuts[:-1], cuts[1:]):
            assert L.fpt_track_writer_write(h, ttext[a:b], b - a) == 0, L.fpt_last_error()
        assert L.fpt_track_writer_close(h) == 0, L.fpt_last_error()
        import gzip
        assert gzip.open(wp, "rb").read() == ttext
        for a, b in ((0, 60000), (16384, 40000), (59990, 70000)):
            assert track_rows(wp, "chr1", a, b) == (0, 0, int(((tpos >= a) & (tpos < b)).sum())), (threads, a, b)
    big = b"".join(b"chrB\t%d\t%d\t1.0\n" % (k, k + 1) for k in range(0, 3000000, 3))  # several groups of members
    h = vp()
    assert L.fpt_track_writer_open(os.path.join(tmp, "big.gz").encode(), C.byref(h)) == 0
    for a in range(0, len(big), 7000001):
        assert L.fpt_track_writer_write(h, big[a:a + 7000001], len(big[a:a + 7000001])) == 0
    assert L.fpt_track_writer_close(h) == 0
    assert track_rows(os.path.join(tmp, "big.gz"), "chrB", 2999000, 3000000) == (0, 0, 333)
    for badtext in (b"chr1\t5\t6\nchr1\t1\t2\n", b"chr1\t5\n", b"chr1\t-4\t6\n", b"chr1\t5\t99999999999999999999\n",
                    b"\t\t\t\n", b"chr1\t5\t6", b"a\t1\t2\nb\t1\t2\na\t5\t6\n"):
        h = vp()
        assert L.fpt_track_writer_open(os.path.join(tmp, "badw.gz").encode(), C.byref(h)) == 0
        rc1 = L.fpt_track_writer_write(h, badtext, len(badtext))
        rc2 = L.fpt_track_writer_close(h)
        assert rc1 != 0 or rc2 != 0, badtext

    # ---- text formatter: against Python's own formatting, serial and threaded, tight buffers
    for threads, n in (("1", 300), ("5", 40000)):
        os.environ["FPT_TEXT_THREADS"] = threads
        m = rs.standard_normal((n, 5)) * 10.0 ** rs.randint(-6, 7, (n, 5))
        m[rs.randint(0, n, 20), rs.randint(0, 5, 20)] = rs.choice([np.nan, np.inf, -np.inf, 0.0, -0.0, 0.5, 2.5e-5, 1e300], 20)
        for prec in (4, 0, 9, 12):
            want = "".join("chrX\t%d\t%d\t%s\n" % (1000 + i, 1001 + i, "\t".join(("{:0.%df}" % prec).format(v) for v in m[i]))
                           for i in range(n)).encode()
            for cap in (len(want) + 64 + (n // 4096 + 1) * 64, len(want) // 2):
                buf = C.create_string_buffer(max(cap, 1))
                out = i64()
                rc = L.fpt_format_stats(b"chrX", 1000, m.ctypes.data, n, 5, None, 0, b"\t", prec, buf, cap, C.byref(out))
                if cap >= len(want):
                    assert rc == 0 and buf.raw[:out.value] == want, (threads, prec, rc, L.fpt_last_error())
                else:
                    assert rc != 0
        rows = np.array([0, n - 1, 5], dtype=np.int64)
        buf, out = C.create_string_buffer(4096), i64()
        assert L.fpt_format_stats(b"c", 0, m.ctypes.data, n, 5, rows.ctypes.data, 3, b" ", 4, buf, 4096, C.byref(out)) == 0
        rows[1] = n  # outside the matrix
        assert L.fpt_format_stats(b"c", 0, m.ctypes.data, n, 5, rows.ctypes.data, 3, b" ", 4, buf, 4096, C.byref(out)) != 0

    # ---- a batch of intervals formatted and written in one call: against the text of the per-interval
    #      formatter, into a buffer (exact, too small) and into a track (after whole lines and after half a
    #      line of text); values that take the long path; bad ids / offsets
    L.fpt_format_stats_batch.argtypes = [i64, vp, i32, vp, vp, vp, vp, i32, C.c_char, i32, vp, i64, C.POINTER(i64)]
    L.fpt_track_writer_write_stats.argtypes = [vp, i64, vp, i32, vp, vp, vp, vp, i32, i32]
    import gzip
    for threads in ("1", "7"):
        os.environ["FPT_TEXT_THREADS"] = os.environ["FPT_TRACK_THREADS"] = threads
        lens = rs.randint(0, 400, 700)
        lens[::9] = 0
        boff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        bst = (1000 + np.cumsum(lens + 3) - lens).astype(np.int64)
        bid = (np.arange(700) >= 400).astype(np.int32)
        bst[400:] -= bst[400] - 5
        m = rs.standard_normal((int(boff[-1]), 5)) * 10.0 ** rs.randint(-6, 7, (int(boff[-1]), 5))
        m[3, 1], m[4, 2], m[6, 0] = 1e300, np.nan, -np.inf
        names = (C.c_char_p * 2)(b"chr1", b"chr2_alt")
        want = b"".join((b"chr1", b"chr2_alt")[bid[j]] + b"\t%d\t%d\t" % (bst[j] + i, bst[j] + i + 1)
                        + "\t".join("{:0.4f}".format(v) for v in m[boff[j] + i]).encode() + b"\n"
                        for j in range(700) for i in range(lens[j]))
        for cap in (len(want), len(want) - 1):
            buf, out = C.create_string_buffer(max(cap, 1)), i64()
            rc = L.fpt_format_stats_batch(700, names, 2, bid.ctypes.data, bst.ctypes.data, boff.ctypes.data, m.ctypes.data, 5,
                                          b"\t", 4, buf, cap, C.byref(out))
            assert out.value == len(want) and (rc == 0 and buf.raw[:cap] == want if cap == len(want) else rc != 0), (threads, rc)
        for lead in (b"", b"# a header line\n", b"chr1\t1\t2\thalf a li"):
            wp = os.path.join(tmp, "batch%s.gz" % threads)
            h = vp()
            assert L.fpt_track_writer_open(wp.encode(), C.byref(h)) == 0
            assert L.fpt_track_writer_write(h, lead, len(lead)) == 0
            rc = L.fpt_track_writer_write_stats(h, 700, names, 2, bid.ctypes.data, bst.ctypes.data, boff.ctypes.data,
                                                m.ctypes.data, 5, 4)
            assert rc == 0, L.fpt_last_error()  # (the half line takes the batch's first line for its last columns)
            assert L.fpt_track_writer_write(h, b"chr3\t7\t8\t1\n", 11) == 0
            assert L.fpt_track_writer_close(h) == 0, L.fpt_last_error()
            assert gzip.open(wp, "rb").read() == lead + want + b"chr3\t7\t8\t1\n"
            a0 = int(bst[450])
            assert track_rows(wp, "chr2_alt", a0, a0 + 50)[2] == int(sum(((bst[j] + np.arange(lens[j]) >= a0) &
                                                                        (bst[j] + np.arange(lens[j]) < a0 + 50)).sum()
                                                                       for j in range(400, 700)))
        bad_id = bid.copy()
        bad_id[5] = 2
        bad_off = boff.copy()
        bad_off[7] = bad_off[6] - 1
        out = i64()
        for ids_, off_ in ((bad_id, boff), (bid, bad_off)):
            assert L.fpt_format_stats_batch(700, names, 2, ids_.ctypes.data, bst.ctypes.data, off_.ctypes.data, m.ctypes.data,
                                            5, b"\t", 4, None, 0, C.byref(out)) != 0
        h = vp()  # unsorted batch: the sticky error of the writer
        assert L.fpt_track_writer_open(os.path.join(tmp, "badb.gz").encode(), C.byref(h)) == 0
        rev = bst[:400][::-1].copy()
        rc = L.fpt_track_writer_write_stats(h, 400, names, 2, bid.ctypes.data, rev.ctypes.data, boff.ctypes.data, m.ctypes.data, 5, 4)
        assert rc != 0 and L.fpt_track_writer_close(h) != 0

    # ---- the CPU checker under the sanitizers: the expected-cleavage path with rounding ties, the
    # NB / window functions and one small whole-path batch
    O = C.CDLL(os.path.join(ROOT, "oracle", "libfpt_oracle_asan.so"))
    f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    O.orc_fast_predict.argtypes = [f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_double, f64p, f64p]
    g = np.load(os.path.join(ROOT, "tests", "golden", "predict_ties.npz"))
    for c, (hw, shw, clip, l, _) in enumerate(g["meta"]):
        e, w = np.empty(int(l)), np.empty(int(l))
        O.orc_fast_predict(np.ascontiguousarray(g["obs%d" % c]), np.ascontiguousarray(g["probs%d" % c]), int(l), int(hw),
                           int(shw), float(clip), e, w)
        assert np.array_equal(e, g["exp%d" % c])
    for l in (0, 1, 7, 11, 101, 102):  # degenerate lengths
        e, w = np.empty(max(l, 1)), np.empty(max(l, 1))
        O.orc_fast_predict(np.ones(max(l, 1)), np.full(max(l, 1), 0.1), l, 5, 50, 0.01, e, w)
    print("ASAN-DRIVER OK")


if __name__ == "__main__":
    main(sys.argv[1])
