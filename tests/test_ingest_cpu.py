"""Host side of the cut-count ingestion (no GPU): the library's BGZF/BAM reader on files written by
tests/bamwriter.py, and the FASTA reader.  The reader's parity with pysam is unpinned (pysam /
htslib are not in the image); what is pinned here is the record layout of the BAM specification."""
import os

import numpy as np
import pytest

from .bamwriter import write_bam


def _reads(rs, n, n_ref=2):
    out = []
    for k in range(n):
        cig = rs.choice(["36M", "20M2D16M", "5S31M", "10M100N26M", "30M1I5M", "18=2X16=", "36M4H"])
        out.append(dict(ref=int(rs.randint(0, n_ref)), pos=int(rs.randint(0, 5000)), cigar=str(cig),
                        flag=int(rs.choice([0, 16, 99, 147, 83, 163, 1024, 512, 256, 4, 2048 + 16, 1 + 16])),
                        mapq=int(rs.choice([0, 1, 30, 60, 255]))))
    out.sort(key=lambda r: (r["ref"], r["pos"]))
    return out


def _ref_span(cigar):
    span, num = 0, ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            if ch in "MDN=X":
                span += int(num)
            num = ""
    return span


def test_bam_reader_roundtrip(tmp_path):
    from footprint_tools_amd.cutcounts import read_alignments
    rs = np.random.RandomState(0)
    refs = [("chr1", 100000), ("chrUn_gl000220", 161802)]
    reads = _reads(rs, 5000)
    path = str(tmp_path / "t.bam")
    write_bam(path, refs, reads, block_bytes=2500)
    got_refs, rid, st, en, fl, mq = read_alignments(path, batch=777)  # several read calls
    assert got_refs == refs
    assert rid.size == len(reads)
    assert np.array_equal(rid, [r["ref"] for r in reads]) and np.array_equal(st, [r["pos"] for r in reads])
    assert np.array_equal(fl, [r["flag"] for r in reads]) and np.array_equal(mq, [r["mapq"] for r in reads])
    assert np.array_equal(en, [r["pos"] + _ref_span(r["cigar"]) for r in reads])  # pysam's reference_end
    # an empty file body, and a file that is not BAM
    write_bam(path, refs, [])
    assert read_alignments(path)[1].size == 0
    write_bam(path, refs, reads[:50])
    whole = open(path, "rb").read()
    cut = tmp_path / "cut.bam"
    cut.write_bytes(whole[:len(whole) // 2])  # a file cut in the middle of a block is an error, not an early end
    with pytest.raises((IOError, ValueError)):
        read_alignments(str(cut))
    bad = tmp_path / "x.bam"
    bad.write_bytes(b"not a bam file at all")
    with pytest.raises(IOError):
        read_alignments(str(bad))
    with pytest.raises(IOError):
        read_alignments(str(tmp_path / "missing.bam"))


def test_bam_region_access_through_the_index(tmp_path):
    """With a BAI index beside the file, `regions` reads what samfile.fetch(chrom, start, end) iterates
    (cutcounts.py:191) -- every alignment overlapping a region, none twice -- and not the rest of the
    file; without an index the whole file is read; a damaged index is an error."""
    import ctypes as C
    from footprint_tools_amd import _lib
    from footprint_tools_amd.cutcounts import _bind, merge_regions, read_alignments
    rs = np.random.RandomState(5)
    refs = [("chr1", 400000), ("chr2", 300000), ("chrEmpty", 1000)]
    reads = []
    for k in range(30000):
        ref = int(rs.choice([0, 1], p=[0.6, 0.4]))
        cig = str(rs.choice(["36M", "20M2D16M", "10M30000N26M", "5S31M"], p=[0.7, 0.1, 0.02, 0.18]))
        reads.append(dict(ref=ref, pos=int(rs.randint(0, refs[ref][1] - 40000)), cigar=cig,
                          flag=int(rs.choice([0, 16])), mapq=30))
    reads.sort(key=lambda r: (r["ref"], r["pos"]))
    path = str(tmp_path / "r.bam")
    write_bam(path, refs, reads, block_bytes=4000, index=True)
    st_all = np.array([r["pos"] for r in reads])
    en_all = np.array([r["pos"] + _ref_span(r["cigar"]) for r in reads])
    ref_all = np.array([r["ref"] for r in reads])
    regions = [("chr1", 1000, 3000), ("chr1", 20000, 20500), ("chr1", 150000, 260000), ("chr1", 262000, 262100),
               ("chr2", 16384, 32768), ("chr2", 290000, 299999), ("chrEmpty", 0, 1000), ("chrNone", 5, 10)]
    got_refs, rid, st, en, fl, mq = read_alignments(path, regions=regions, batch=513)
    assert got_refs == refs
    merged = merge_regions(regions)
    assert ("chr1", 150000, 262100) in merged  # closer than the merge gap
    want = np.zeros(len(reads), bool)
    for c, a, b in merged:
        if c in ("chr1", "chr2"):
            want |= (ref_all == (0 if c == "chr1" else 1)) & (st_all < b) & (en_all > a)
    key_got = sorted(zip(rid.tolist(), st.tolist(), en.tolist()))
    key_all = sorted(zip(ref_all[want].tolist(), st_all[want].tolist(), en_all[want].tolist()))
    # everything that overlaps a region is there, once ...
    import collections
    cg, cw = collections.Counter(key_got), collections.Counter(key_all)
    assert all(cg[k] >= v for k, v in cw.items())
    full = collections.Counter(zip(ref_all.tolist(), st_all.tolist(), en_all.tolist()))
    assert all(v <= full[k] for k, v in cg.items())  # ... nothing more often than the file has it
    # ... and what comes along besides (alignments of the start window that end before the region) is little
    assert len(key_got) < 0.8 * len(reads) and len(key_got) - len(key_all) < 0.1 * len(reads)
    # seek_region / read at the C ABI: a region beyond the last alignment, and an empty one
    L = _bind(_lib.load())
    h = C.c_void_p()
    _lib.check(L.fpt_bam_open(path.encode(), C.byref(h)))
    yes = C.c_int32()
    _lib.check(L.fpt_bam_has_index(h, C.byref(yes)))
    assert yes.value == 1
    bufs = [np.empty(100, dt) for dt in (np.int32, np.int32, np.int32, np.uint16, np.uint8)]
    got = C.c_int64()
    for ref_id, a, b in ((0, 399000, 400000), (2, 0, 1000), (1, 500, 500)):
        _lib.check(L.fpt_bam_seek_region(h, ref_id, a, b))
        _lib.check(L.fpt_bam_read(h, 100, *[x.ctypes.data for x in bufs], C.byref(got)))
        assert got.value == 0
    assert L.fpt_bam_seek_region(h, 7, 0, 10) != 0
    L.fpt_bam_close(h)
    # no index: regions are ignored, the whole file comes back
    plain = str(tmp_path / "plain.bam")
    write_bam(plain, refs, reads[:2000], block_bytes=4000)
    assert read_alignments(plain, regions=[("chr1", 0, 10)])[1].size == 2000
    h = C.c_void_p()
    _lib.check(L.fpt_bam_open(plain.encode(), C.byref(h)))
    assert L.fpt_bam_seek_region(h, 0, 0, 10) != 0 and b"index" in L.fpt_last_error()
    L.fpt_bam_close(h)
    # a damaged index
    bai = open(path + ".bai", "rb").read()
    for damaged in (bai[:len(bai) // 2], b"BAX\1" + bai[4:], bai[:8] + b"\xff\xff\xff\x7f" + bai[12:]):
        open(plain + ".bai", "wb").write(damaged)
        with pytest.raises(IOError):
            read_alignments(plain)


class _Iv(object):
    def __init__(self, c, s, e):
        self.chrom, self.start, self.end = c, s, e


def test_fasta_reader(tmp_path):
    from footprint_tools_amd.fasta import FastaFile
    rs = np.random.RandomState(1)
    seqs = {"chr1": "".join(rs.choice(list("ACGTacgtN"), 1234)), "chr2": "".join(rs.choice(list("ACGT"), 61)),
            "empty": ""}
    path = tmp_path / "g.fa"
    with open(path, "w") as f:
        for name, s in seqs.items():
            f.write(">%s some description\n" % name)
            for a in range(0, len(s), 60):
                f.write(s[a:a + 60] + "\n")
    for use_fai in (False, True):
        if use_fai:
            off = 0
            with open(str(path) + ".fai", "w") as fai, open(path) as f:
                text = f.read()
            pos = 0
            with open(str(path) + ".fai", "w") as fai:
                for name, s in seqs.items():
                    pos = text.index(">" + name) + len(">%s some description\n" % name)
                    fai.write("%s\t%d\t%d\t60\t61\n" % (name, len(s), pos))
        fa = FastaFile(str(path))
        assert fa.references == list(seqs)
        for chrom, a, b in (("chr1", 0, 1234), ("chr1", 59, 61), ("chr1", 60, 120), ("chr1", 100, 1000),
                            ("chr2", 0, 61), ("chr2", 10, 11), ("chr1", 1200, 1234)):
            assert fa.fetch(chrom, a, b) == seqs[chrom][a:b], (chrom, a, b, use_fai)
        # outside the chromosome: N (the scan then uses the default propensity)
        assert fa.fetch("chr2", -5, 3) == "NNNNN" + seqs["chr2"][:3]
        assert fa.fetch("chr2", 58, 66) == seqs["chr2"][58:] + "NNNNN"
        assert fa.fetch("nope", 0, 4) == "NNNN"
        ivs = [_Iv("chr1", 200, 300), _Iv("chr2", 5, 20)]
        batch = fa.fetch_batch(ivs, pad=55)
        want = "".join(fa.fetch(iv.chrom, iv.start - 55 - 1 - 3, iv.end + 55 + 3) for iv in ivs)
        assert batch.tobytes().decode() == want and batch.size == sum(iv.end - iv.start + 117 for iv in ivs)
        fa.close()


def test_tabix_track_reader_and_load_data(tmp_path):
    """cli/post.py:57-87 `_load_data` on bgzip-compressed per-nucleotide tracks (the format
    cli/utils.py:119-144 writes): columns 3 / 4 / 7 -> exp / obs / fdr, w = 1 where a dataset has
    a row, defaults (0, 0, 1, 0) elsewhere."""
    import json
    from footprint_tools_amd.post import posterior_stats
    from footprint_tools_amd.tabix import TabixFile
    from footprint_tools_amd.modeling import dispersion
    from .bamwriter import _bgzf_block
    rs = np.random.RandomState(5)
    dm = dispersion.dispersion_model()
    dm.mu_params = [25, 50, 75, 0, 0.5, 1.0, 1.0, 0.98, 0.97]
    dm.r_params = [3, 7, 15, 25, 75, 0.05, 0.08, 0.115, 0.16, 0.185, 0.02, 0.01, 0.005, 0.002, 0.001]
    dm_path = tmp_path / "dm.json"
    dm_path.write_text(dispersion.write_dispersion_model(dm))
    truth, rows = [], []
    for d in range(3):
        lines, cols = ["#chrom\tstart\tend\texp\tobs\tlnp\twinlnp\tfdr"], {}
        for chrom, a, b in (("chr1", 1000, 1400), ("chr1", 5000, 5100), ("chr2", 10, 60)):
            if d == 2 and chrom == "chr2":
                continue  # this dataset has no hotspot there
            for x in range(a, b):
                e, o, f = float(rs.randint(0, 30)), float(rs.randint(0, 40)), float(rs.rand())
                cols[(chrom, x)] = (e, o, f)
                lines.append("%s\t%d\t%d\t%.4f\t%.4f\t%.4f\t%.4f\t%.4f" % (chrom, x, x + 1, e, o, rs.rand(), rs.rand(), f))
        data = ("\n".join(lines) + "\n").encode()
        path = tmp_path / ("d%d.bedgraph.gz" % d)
        with open(path, "wb") as fh:
            for k in range(0, len(data), 5000):
                fh.write(_bgzf_block(data[k:k + 5000]))
            fh.write(_bgzf_block(b""))
        truth.append(cols)
        rows.append(dict(id="s%d" % d, tabix_file=str(path), dm_file=str(dm_path), beta_a=1.0 + d, beta_b=2.0))
    t = TabixFile(rows[0]["tabix_file"])
    pos, vals = t.fetch_columns("chr1", 1390, 5003)
    assert list(pos) == list(range(1390, 1400)) + [5000, 5001, 5002] and vals.shape == (13, 7)
    assert [r[1] for r in t.fetch("chr2", 58, 70)] == ["58", "59"]
    ps = posterior_stats([("chr1", 1350, 1450), ("chr2", 0, 70)], rows, fdr_cutoff=0.05)
    ps._open_tabix_files()
    for iv in ps.intervals:
        obs, exp, fdr, w = ps._load_data(iv)
        assert obs.shape == (3, len(iv))
        for d in range(3):
            for j in range(len(iv)):
                e, o, f = truth[d].get((iv.chrom, iv.start + j), (0.0, 0.0, 1.0))
                assert abs(exp[d, j] - e) < 1e-9 and abs(obs[d, j] - o) < 1e-9 and abs(fdr[d, j] - round(f, 4)) < 1e-9
                assert w[d, j] == float((iv.chrom, iv.start + j) in truth[d])
    ps.cleanup()


def test_bam_reader_threads_and_missing_cigar(tmp_path, monkeypatch):
    """the block-parallel inflate gives the same alignments whatever the team size, and a read
    without a reference-consuming CIGAR operation has no reference_end (-1, pysam: None)."""
    from footprint_tools_amd.cutcounts import read_alignments
    rs = np.random.RandomState(3)
    refs = [("chr1", 100000)]
    reads = _reads(rs, 20000, n_ref=1)
    reads[10]["cigar"] = "36S"      # soft clip only
    reads[11]["cigar"] = "10I26S"
    path = str(tmp_path / "t.bam")
    write_bam(path, refs, reads, block_bytes=1500)
    got = {}
    for nt in ("1", "3", "16"):
        monkeypatch.setenv("FPT_BAM_THREADS", nt)
        got[nt] = read_alignments(path, batch=4096)
    for nt in ("3", "16"):
        for a, b in zip(got["1"][1:], got[nt][1:]):
            assert np.array_equal(a, b)
    en = got["1"][3]
    assert en[10] == -1 and en[11] == -1
    assert np.array_equal(np.delete(en, [10, 11]), [r["pos"] + _ref_span(r["cigar"]) for i, r in enumerate(reads) if i not in (10, 11)])


def test_sanitizer_build(tmp_path):
    """`make asan`: the BGZF / BAM reader, the text formatter (footprint_tools_amd/csrc) and the CPU
    checker built with -fsanitize=address,undefined, driven by tests/asan_driver.py under
    LD_PRELOAD=libasan.so over good files and a corpus of truncated / bit-flipped / inconsistent
    ones (bad l_name, n_cigar past the record, negative l_text, BSIZE / XLEN / ISIZE lies...)."""
    import os
    import shutil
    import subprocess
    import sys
    from .conftest import ROOT
    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE).stdout.decode().strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan is not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "footprint_tools_amd", "csrc"), "asan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py"), str(tmp_path)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0 and "ASAN-DRIVER OK" in text, text[-3000:]
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-3000:]


def _track_text(rs, chroms=(("chr1", 70000), ("chr2", 40000), ("chrX", 20000))):
    """a position-sorted `detect` track with gaps; returns (text, {chrom: (pos, values)})"""
    lines, truth = [b"# generated by a test"], {}
    for name, length in chroms:
        pos = np.flatnonzero(rs.rand(length) < 0.3) + 5
        pos = pos[(pos < 100) | (pos > 20000)]  # windows of the linear index without a row
        vals = np.round(rs.gamma(2.0, 5.0, (pos.size, 5)), 4)
        truth[name] = (pos, vals)
        for p, v in zip(pos, vals):
            lines.append(("%s\t%d\t%d\t%.4f\t%.4f\t%.4f\t%.4f\t%.4f" % ((name, p, p + 1) + tuple(v))).encode())
    return b"\n".join(lines) + b"\n", truth


@pytest.mark.parametrize("form", ["bgzf+tbi", "bgzf", "plain"])
def test_track_reader_region_access(tmp_path, form, monkeypatch):
    """the library's tabix-style reader (fpt_track_*): region queries against the rows written, with
    a .tbi (tabix linear index), with the index built at open, and on an uncompressed file; the
    batched fetch scatters the wanted columns into (bases) arrays like cli/post.py:70-83."""
    from footprint_tools_amd.tabix import TabixFile
    from .tbiwriter import write_bgzf_with_tbi
    rs = np.random.RandomState(12)
    text, truth = _track_text(rs)
    path = str(tmp_path / "t.bedgraph.gz")
    if form == "plain":
        path = str(tmp_path / "t.bedgraph")
        open(path, "wb").write(text)
    else:
        write_bgzf_with_tbi(path, text, block_bytes=1777, tbi=(form == "bgzf+tbi"))
    tb = TabixFile(path)
    assert tb.has_tbi == (form == "bgzf+tbi") and tb.contigs == ["chr1", "chr2", "chrX"]
    queries = [("chr1", 0, 50), ("chr1", 90, 20100), ("chr1", 5000, 9000), ("chr1", 30000, 30001), ("chr2", 39990, 50000),
               ("chrX", 0, 20010), ("chr2", 16384, 32768), ("chrQ", 5, 50), ("chr1", 65000, 66000), ("chr1", 500000, 600000)]
    for chrom, a, b in queries:
        pos, vals = tb.fetch_columns(chrom, a, b)
        tp, tv = truth.get(chrom, (np.zeros(0, int), np.zeros((0, 5))))
        sel = (tp >= a) & (tp < b)
        assert np.array_equal(pos, tp[sel]), (chrom, a, b)
        assert np.array_equal(vals[:, 2:7], tv[sel]) and np.array_equal(vals[:, 0], tp[sel]) and np.array_equal(vals[:, 1], tp[sel] + 1)
    rows = list(tb.fetch("chr2", 20005, 20300))
    assert len(rows) == int(((truth["chr2"][0] >= 20005) & (truth["chr2"][0] < 20300)).sum()) and rows[0][0] == "chr2"
    # batched, on one thread and on a team
    ivs = [(c, int(a), int(a + l)) for c, a, l in zip(rs.choice(["chr1", "chr2", "chrX", "chrM"], 400),
                                                      rs.randint(0, 45000, 400), rs.randint(1, 600, 400))]
    # ... in any order, and sorted (neighbours are then read in one walk from where the last one stopped:
    # abutting, overlapping, nested, far apart, a change of chromosome)
    walk = sorted(ivs) + [("chr2", a, a + 37) for a in range(100, 3000, 37)] + [("chr2", 2990, 3500), ("chr2", 3000, 3010),
                                                                               ("chr2", 3010, 3010), ("chr2", 20000, 20100),
                                                                               ("chrX", 5, 90), ("chrX", 90, 300)]
    for nt, ivs in (("1", ivs), ("7", ivs), ("1", walk), ("5", walk)):
        monkeypatch.setenv("FPT_TRACK_THREADS", nt)
        tb2 = TabixFile(path)
        out, present, off = tb2.fetch_batch([c for c, _, _ in ivs], [a for _, a, _ in ivs], [b for _, _, b in ivs], [3, 4, 7])
        for k, (c, a, b) in enumerate(ivs):
            want = np.full((3, b - a), np.nan)
            wp = np.zeros(b - a)
            if c in truth:
                tp, tv = truth[c]
                sel = (tp >= a) & (tp < b)
                want[:, tp[sel] - a] = tv[sel][:, [0, 1, 4]].T
                wp[tp[sel] - a] = 1.0
            sl = slice(off[k], off[k + 1])
            for j in range(3):
                assert np.array_equal(out[j][sl], want[j], equal_nan=True), (nt, k)
            assert np.array_equal(present[sl], wp)
        tb2.close()
    tb.close()
    with pytest.raises(IOError):
        TabixFile(str(tmp_path / "missing.gz"))


def test_track_reader_damage(tmp_path):
    """a truncated or bit-flipped track file is an error (at open without an index, at the query with
    one), never a crash"""
    from footprint_tools_amd.tabix import TabixFile
    from .tbiwriter import write_bgzf_with_tbi
    rs = np.random.RandomState(13)
    text, truth = _track_text(rs, chroms=(("chr1", 30000),))
    path = str(tmp_path / "t.gz")
    write_bgzf_with_tbi(path, text, block_bytes=2500, tbi=False)
    raw = open(path, "rb").read()
    bad = tmp_path / "cut.gz"
    bad.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(IOError):
        TabixFile(str(bad))
    for k in range(20):
        b = bytearray(raw)
        b[int(rs.randint(0, len(raw) - 28))] ^= 1 << int(rs.randint(0, 8))
        bad.write_bytes(bytes(b))
        try:
            tb = TabixFile(str(bad))
            tb.fetch_columns("chr1", 0, 30000)
            tb.close()
        except (IOError, ValueError):
            pass
    write_bgzf_with_tbi(path, text, block_bytes=2500, tbi=True)  # with an index the damage shows at the query
    raw = open(path, "rb").read()
    b = bytearray(raw)
    b[len(raw) // 2] ^= 0x10
    open(path, "wb").write(bytes(b))
    tb = TabixFile(path)
    with pytest.raises((IOError, ValueError)):
        tb.fetch_columns("chr1", 0, 30000)


def test_track_writer_round_trip(tmp_path, monkeypatch):
    """TrackWriter: bedGraph text in -> bgzip + tabix files out.  The data decompresses to the text
    (Python's gzip reads BGZF), the index equals the one tests/tbiwriter.py makes of the same text cut
    at the same member size, the library's reader serves regions through it, and unsorted or malformed
    lines are errors."""
    import gzip
    from footprint_tools_amd.tabix import TabixFile, TrackWriter
    from .tbiwriter import write_bgzf_with_tbi
    rs = np.random.RandomState(11)
    lines = [b"#chrom\tstart\tend\texp\tobs\tlp\tlwp\tfdr"]
    truth = {}
    for chrom, n, span in (("chr1", 60000, 400000), ("chr2", 9000, 70000), ("chrX", 3, 100)):
        pos = np.sort(rs.choice(span, n, replace=False))
        vals = rs.rand(n, 5) * 9
        truth[chrom] = (pos, vals)
        for p_, v in zip(pos, vals):
            lines.append(("%s\t%d\t%d\t%.4f\t%.4f\t%.4f\t%.4f\t%.4f" % ((chrom, p_, p_ + 1) + tuple(v))).encode())
    text = b"\n".join(lines) + b"\n"
    for threads in ("1", "5"):
        monkeypatch.setenv("FPT_TRACK_THREADS", threads)
        path = str(tmp_path / ("t%s.bed.gz" % threads))
        with TrackWriter(path) as w:
            cuts = sorted(set([0, len(text)] + rs.randint(0, len(text), 40).tolist()))  # pieces that split lines
            for a, b in zip(cuts[:-1], cuts[1:]):
                w.write(text[a:b])
        assert gzip.open(path, "rb").read() == text
        # the same text through the test helper at the writer's member size: identical index
        ref = str(tmp_path / "ref.bed.gz")
        write_bgzf_with_tbi(ref, text, block_bytes=0xff00)
        got_idx, want_idx = gzip.open(path + ".tbi", "rb").read(), gzip.open(ref + ".tbi", "rb").read()
        if open(path, "rb").read() == open(ref, "rb").read():  # (same zlib, same members: then the offsets agree too)
            assert got_idx == want_idx
        assert got_idx[:4] == b"TBI\x01" and len(got_idx) == len(want_idx)
        tb = TabixFile(path)
        assert tb.has_tbi and tb.contigs == ["chr1", "chr2", "chrX"]
        for chrom, a, b in (("chr1", 0, 400000), ("chr1", 16384, 16484), ("chr1", 123456, 200001), ("chr2", 69000, 70000),
                            ("chrX", 0, 100), ("chr2", 0, 1)):
            pos, vals = truth[chrom]
            sel = (pos >= a) & (pos < b)
            rows = list(tb.fetch(chrom, a, b))
            assert len(rows) == int(sel.sum())
            assert [int(r[1]) for r in rows] == pos[sel].tolist()
            assert np.allclose([float(r[3]) for r in rows], vals[sel, 0], atol=5e-5)
        tb.close()
    # errors: unsorted positions, a chromosome that comes back, a malformed line, no final newline
    for bad in (b"chr1\t5\t6\t1\nchr1\t2\t3\t1\n", b"chr1\t5\t6\nchr2\t1\t2\nchr1\t9\t10\n", b"chr1\tx\t6\n", b"chr1\t5\n",
                b"chr1\t5\t6\t1"):
        w = TrackWriter(str(tmp_path / "bad.gz"))
        with pytest.raises(ValueError):
            w.write(bad)
            w.close()
    with pytest.raises(IOError):
        TrackWriter(str(tmp_path / "no_such_dir" / "x.gz"))


def test_detect_writer_into_an_indexed_track(tmp_path):
    """write_stats_to_output (cli/utils.py:119-144) with a TrackWriter for the file: the records of a
    run of intervals become a bgzip + tabix track that the posterior caller's reader serves back --
    the reference's detect -> bgzip -> tabix -> post chain without the external tools."""
    from footprint_tools_amd import detect
    from footprint_tools_amd.tabix import TabixFile, TrackWriter
    rs = np.random.RandomState(2)
    ivs = [_Iv("chr1", 1000, 1800), _Iv("chr1", 5000, 5050), _Iv("chr3", 10, 700)]
    stats = [np.round(rs.rand(iv.end - iv.start, 5) * 7, 4) for iv in ivs]
    path = str(tmp_path / "stats.bed.gz")
    with TrackWriter(path) as w:
        for iv, st in zip(ivs, stats):
            detect.write_stats_to_output(iv, st, file=w)
    tb = TabixFile(path)
    assert tb.has_tbi and tb.contigs == ["chr1", "chr3"]
    out, present, off = tb.fetch_batch([iv.chrom for iv in ivs], [iv.start for iv in ivs], [iv.end for iv in ivs], [3, 4, 7])
    assert present.all()
    for k, c in enumerate((0, 1, 4)):  # exp, obs, fdr columns of the records
        assert np.allclose(out[k], np.concatenate([st[:, c] for st in stats]), atol=5e-5)
    rows = list(tb.fetch("chr1", 1798, 5002))
    assert [r[1] for r in rows] == ["1798", "1799", "5000", "5001"] and rows[0][2] == "1799" and len(rows[0]) == 8


def test_batch_writer_matches_the_per_interval_writer(tmp_path, monkeypatch):
    """write_batch_to_output (fpt_format_stats_batch / fpt_track_writer_write_stats: a batch_iter step
    formatted in one call, on threads) writes the bytes the loop of write_stats_to_output over the
    step's records does (cli/detect.py:399-411 + cli/utils.py:119-144) -- as text, as bytes, with
    other formats and delimiters, into a track (file and index), on one thread and on many; empty
    intervals, a change of chromosome, nan / inf / huge / negative-zero values included."""
    import io
    from footprint_tools_amd import detect
    from footprint_tools_amd.tabix import TrackWriter
    rs = np.random.RandomState(0)
    ivs, off, pos = [], [0], 100
    for j in range(3000):
        n = int(rs.randint(0, 300)) if j % 50 else 0
        ivs.append(_Iv("chr1" if j < 2000 else "chrX_random", pos, pos + n))
        pos = 10 if j == 1999 else pos + n + 5
        off.append(off[-1] + n)
    table = rs.lognormal(0, 3, (off[-1], 5))
    table[5, 2], table[7, 3], table[9, 1], table[11, 0], table[13, 4] = np.nan, np.inf, 3e20, -0.0, -np.inf
    stats = [table[a:b] for a, b in zip(off[:-1], off[1:])]
    batch = {"interval": ivs, "stats": stats, "table": table, "row_off": np.array(off)}

    def loop(n=None, **kw):
        f = io.StringIO()
        for iv, st in zip(ivs[:n], stats[:n]):
            detect.write_stats_to_output(iv, st, file=f, **kw)
        return f.getvalue()

    want = loop()
    for threads in ("8", "1"):
        monkeypatch.setenv("FPT_TEXT_THREADS", threads)
        f = io.StringIO()
        detect.write_batch_to_output(batch, file=f)
        assert f.getvalue() == want
    monkeypatch.delenv("FPT_TEXT_THREADS")
    fb = io.BytesIO()
    detect.write_batch_to_output(batch, file=fb)
    assert fb.getvalue() == want.encode()
    part = {"interval": ivs[:200], "stats": stats[:200], "table": table, "row_off": np.array(off[:201])}
    for fmt, delim in (("0.2f", ","), ("0.9f", "\t"), ("0.12f", " "), ("0.3e", "\t"), ("0.40f", "\t")):  # the last two: the loop
        f = io.StringIO()
        detect.write_batch_to_output(part, file=f, fmt_string=fmt, delim=delim)
        assert f.getvalue() == loop(200, fmt_string=fmt, delim=delim), fmt
    f = io.StringIO()  # a batch without its matrix (somebody else's records): the loop
    detect.write_batch_to_output({"interval": ivs[:20], "stats": stats[:20]}, file=f)
    assert f.getvalue() == loop(20)
    f = io.StringIO()
    detect.write_batch_to_output({"interval": [], "stats": [], "table": np.empty((0, 5)), "row_off": np.array([0])}, file=f)
    assert f.getvalue() == ""
    a, b = str(tmp_path / "a.bed.gz"), str(tmp_path / "b.bed.gz")
    with TrackWriter(a) as w:
        for iv, st in zip(ivs, stats):
            detect.write_stats_to_output(iv, st, file=w)
    with TrackWriter(b) as w:
        detect.write_batch_to_output(batch, file=w)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a + ".tbi", "rb").read() == open(b + ".tbi", "rb").read()
    import gzip
    from footprint_tools_amd.tabix import TabixFile
    fast = str(tmp_path / "fast.bed.gz")  # zlib level 1: other bytes, the same text and the same answers
    with TrackWriter(fast, level=1) as w:
        detect.write_batch_to_output(batch, file=w)
    assert gzip.open(fast, "rb").read() == want.encode() and os.path.getsize(fast) > os.path.getsize(a)
    ta, tf = TabixFile(a), TabixFile(fast)
    iv = ivs[2501]
    assert list(ta.fetch(iv.chrom, iv.start, iv.end)) == list(tf.fetch(iv.chrom, iv.start, iv.end)) != []
    with pytest.raises(ValueError):
        TrackWriter(str(tmp_path / "lvl.gz"), level=12)
    with pytest.raises(ValueError):  # offsets beyond the matrix
        detect.write_batch_to_output(dict(batch, row_off=np.array(off) + 1), file=io.StringIO())
    w = TrackWriter(str(tmp_path / "c.bed.gz"))
    with pytest.raises(ValueError):  # unsorted: the later interval first
        w.write_stats(["chr1", "chr1"], [500, 100], [0, 2, 4], table[:4])
    with pytest.raises(ValueError):  # (and close reports it again)
        w.close()


@pytest.mark.parametrize("tbi", [True, False])
def test_track_reader_passes_over_members_safely(tmp_path, tbi):
    """A query goes from the index's 16 kb window towards its first row by looking at how the members in
    between begin (fpt_track.cpp hop_members): rows that share a start on both sides of a member's edge,
    members that begin in the middle of a line, a chromosome that ends inside the window, a header line,
    very long lines -- every query still returns exactly its rows."""
    from footprint_tools_amd.tabix import TabixFile
    from .tbiwriter import write_bgzf_with_tbi
    rs = np.random.RandomState(31)
    rows = []
    for pos in range(100, 9000):
        for _ in range(40 if pos in (5000, 5001, 7777) else 1):  # 40 rows of one start: several members' worth
            rows.append(("chrA", pos, "%.4f" % rs.rand()))
    rows.append(("chrA", 9000, "x" * 3000))  # a line longer than the look at a member's beginning
    rows += [("chrA", pos, "1.0") for pos in range(9001, 9100)]
    rows += [("chrB", pos, "2.0") for pos in range(0, 3000)]
    text = b"# a header\n" + b"".join(b"%s\t%d\t%d\t%s\n" % (c.encode(), p, p + 1, v.encode()) for c, p, v in rows)
    path = str(tmp_path / "dups.bed.gz")
    write_bgzf_with_tbi(path, text, block_bytes=700, tbi=tbi)
    tb = TabixFile(path)
    starts = {}
    for c, p, _ in rows:
        starts.setdefault(c, []).append(p)
    starts = {c: np.array(v) for c, v in starts.items()}
    for chrom, a, b in [("chrA", 5000, 5001), ("chrA", 5001, 5002), ("chrA", 4999, 5003), ("chrA", 7777, 7778), ("chrA", 8999, 9002),
                        ("chrA", 9050, 9200), ("chrB", 0, 5), ("chrB", 2990, 4000), ("chrA", 100, 101), ("chrA", 0, 100)] + \
                       [("chrA", int(x), int(x) + 30) for x in rs.randint(90, 9100, 60)] + \
                       [("chrB", int(x), int(x) + 30) for x in rs.randint(0, 3000, 20)]:
        got = [int(r[1]) for r in tb.fetch(chrom, a, b)]
        want = starts[chrom][(starts[chrom] >= a) & (starts[chrom] < b)].tolist()
        assert got == want, (chrom, a, b, len(got), len(want))
    tb.close()


