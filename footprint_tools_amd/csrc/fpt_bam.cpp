// fpt_bam.cpp -- host-side alignment reader for the cut-count ingestion (no htslib in this image).
//
// BGZF (SAM/BAM specification, section 4.1) is a chain of independent gzip members of at most
// 64 KiB, each carrying its own compressed size in the "BC" extra subfield.  The reader takes the
// file a chunk of compressed bytes at a time, walks the member headers to find the block
// boundaries, inflates the blocks of the chunk on a team of threads (raw deflate per block, CRC32
// and ISIZE checked) into one contiguous buffer, and parses the BAM records (section 4.2) of that
// buffer in one sequential walk -- a record header is 36 bytes and the walk only follows
// block_size, so it is the inflate that the threads are for.  Every alignment's (reference id,
// start, end, flag, MAPQ) goes to the caller in batches; the cut position and the read filters of
// the reference (cutcounts.py:119-145, 196-205, 231-248) are applied on the device by
// k_cut_counts.  reference_end = start + reference-consuming CIGAR operations (M, D, N, =, X), as
// pysam computes it; an alignment without such an operation (no CIGAR, or the placeholder of a
// CIGAR kept in the CG tag) has none -- pysam returns None -- and is handed over with end = -1.
//
// This file parses untrusted input: every length read from the file is checked against the bytes
// that are there before it is used (tests/test_ingest_cpu.py runs a corpus of truncated and
// bit-flipped files through an AddressSanitizer build, `make asan`).
//
// Region access (what the reference does per interval, samfile.fetch(chrom, start - 10, end + 10),
// cutcounts.py:191): when a BAI index lies beside the file (SAM specification section 5.2),
// fpt_bam_seek_region takes the linear index entry of the 16 kb window that holds the start of the
// region -- the smallest virtual offset of an alignment overlapping that window --, seeks to that
// block and fpt_bam_read then hands out alignments until one starts at or beyond the end of the
// region.  (The bins of the index are skipped: for a coordinate-sorted file the linear index alone
// bounds the scan from below, and the scan ends at the first alignment past the region.)
//
// PARITY UNPINNED for this reader: pysam / htslib are not in the image and the reference ships no
// alignment fixtures, so it is tested on BAM files written by the tests themselves.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fpt.h"
#include "fpt_host_threads.hpp"

int fpt_internal_fail(int code, const char *fmt, ...);  // fpt_capi.cpp

#include "fpt_bgzf.hpp"

namespace {

using fptz::bgzf_block;
using fptz::bgzf_member_size;
using fptz::kMaxBlock;
constexpr size_t kChunk = (size_t)32 << 20;      // compressed bytes taken from the file at a time
constexpr size_t kSeekChunk = (size_t)128 << 10;  // ... after a seek to a region (doubling up to kChunk): a short region is a block or two

}  // namespace

struct fpt_bam {
    FILE *f = nullptr;
    bool eof = false;
    std::vector<unsigned char> in;   // compressed bytes not yet inflated (whole members + a partial tail)
    size_t in_len = 0;
    std::vector<unsigned char> out;  // inflated bytes not yet consumed
    size_t out_pos = 0;
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    std::string error;
    int n_threads = 1;
    size_t chunk = kSeekChunk / 4;  // (the header first: a region read must not inflate 32 MB to see it)
    // BAI: per reference the linear index (virtual offset of the first alignment overlapping each
    // 16 kb window; 0 = none recorded), empty when the file has no index
    bool have_index = false;
    std::vector<std::vector<uint64_t>> linear;
    // region mode (after fpt_bam_seek_region): alignments of region_rid that start before region_end
    bool in_region = false, region_done = false;
    int32_t region_rid = -1;
    int64_t region_end = 0;

    // inflate the whole members sitting in `in`; returns false on error (message in `error`)
    bool inflate_chunk() {
        std::vector<bgzf_block> blocks;
        size_t p = 0, total_out = 0;
        while (p < in_len) {
            size_t pay = 0;
            const long sz = bgzf_member_size(in.data() + p, in_len - p, &pay);
            if (sz < 0) {
                error = "not a BGZF block (bad gzip member header)";
                return false;
            }
            if (sz == 0 || (size_t)sz > in_len - p) break;  // partial member: wait for more bytes
            const unsigned char *tail = in.data() + p + sz - 8;
            bgzf_block b;
            b.cpos = p + pay;
            b.clen = (uint32_t)(sz - (long)pay - 8);
            std::memcpy(&b.crc, tail, 4);
            std::memcpy(&b.isize, tail + 4, 4);
            if (b.isize > kMaxBlock) {
                error = "corrupt BGZF block (ISIZE beyond 64 KiB)";
                return false;
            }
            b.opos = total_out;
            total_out += b.isize;
            blocks.push_back(b);
            p += (size_t)sz;
        }
        if (blocks.empty()) {
            if (eof && in_len > 0) error = "truncated BGZF block at the end of the file";
            return error.empty();
        }
        // drop what has been consumed, make room, inflate in parallel
        if (out_pos > 0) {
            out.erase(out.begin(), out.begin() + (long)out_pos);
            out_pos = 0;
        }
        const size_t base = out.size();
        out.resize(base + total_out);
        std::atomic<size_t> next(0);
        std::atomic<int> bad(0);
        auto work = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= blocks.size() || bad.load()) return;
                const bgzf_block &b = blocks[i];
                if (!fptz::inflate_block(in.data() + b.cpos, b.clen, out.data() + base + b.opos, b.isize, b.crc)) {
                    bad.store(2);
                    return;
                }
            }
        };
        const int nt = (int)std::min<size_t>((size_t)n_threads, blocks.size());
        if (nt <= 1) {
            work();
        } else {
            std::vector<std::thread> team;
            for (int t = 0; t < nt; ++t) team.emplace_back(work);
            for (std::thread &t : team) t.join();
        }
        if (bad.load()) {
            error = "corrupt BGZF block";
            return false;
        }
        std::memmove(in.data(), in.data() + p, in_len - p);
        in_len -= p;
        return true;
    }

    // make at least `n` inflated bytes available at out[out_pos..]; false at end of file or on error
    bool need(size_t n) {
        while (out.size() - out_pos < n) {
            if (!error.empty()) return false;
            if (!eof && in_len < chunk) {
                if (in.size() < chunk + kMaxBlock) in.resize(chunk + kMaxBlock);
                const size_t got = fread(in.data() + in_len, 1, chunk + kMaxBlock - in_len, f);
                in_len += got;
                if (got == 0) eof = true;
                if (chunk < kChunk) chunk = std::min(kChunk, chunk * 2);  // (a region read: start small)
            }
            const size_t before = out.size() - out_pos;
            if (!inflate_chunk()) return false;
            if (out.size() - out_pos == before) {  // nothing new
                if (eof) {
                    if (in_len > 0 && error.empty()) error = "truncated BGZF block at the end of the file";
                    return false;
                }
            }
        }
        return true;
    }
    // the BAI index beside the file, if there is one (<path>.bai, or <path without .bam>.bai);
    // a damaged index is an error, a missing one is not
    bool load_index(const std::string &path) {
        FILE *g = fopen((path + ".bai").c_str(), "rb");
        if (!g && path.size() > 4 && path.compare(path.size() - 4, 4, ".bam") == 0)
            g = fopen((path.substr(0, path.size() - 4) + ".bai").c_str(), "rb");
        if (!g) return true;
        std::vector<unsigned char> buf;
        unsigned char tmp[1 << 16];
        size_t got;
        while ((got = fread(tmp, 1, sizeof tmp, g)) > 0) buf.insert(buf.end(), tmp, tmp + got);
        fclose(g);
        size_t p = 0;
        auto rd32 = [&](int32_t *v) {
            if (buf.size() - p < 4) return false;
            std::memcpy(v, buf.data() + p, 4);
            p += 4;
            return true;
        };
        int32_t n_ref = 0;
        if (buf.size() < 8 || std::memcmp(buf.data(), "BAI\1", 4) != 0) {
            error = "not a BAI index";
            return false;
        }
        p = 4;
        if (!rd32(&n_ref) || n_ref < 0 || (size_t)n_ref > buf.size()) {
            error = "damaged BAI index";
            return false;
        }
        linear.assign((size_t)n_ref, std::vector<uint64_t>());
        for (int32_t r = 0; r < n_ref; ++r) {
            int32_t n_bin = 0, n_intv = 0;
            if (!rd32(&n_bin) || n_bin < 0) {
                error = "damaged BAI index";
                return false;
            }
            for (int32_t b = 0; b < n_bin; ++b) {  // bins: skipped (bin id, chunk count, 16 bytes per chunk)
                int32_t bin = 0, n_chunk = 0;
                if (!rd32(&bin) || !rd32(&n_chunk) || n_chunk < 0 || (uint64_t)n_chunk * 16 > buf.size() - p) {
                    error = "damaged BAI index";
                    return false;
                }
                p += (size_t)n_chunk * 16;
            }
            if (!rd32(&n_intv) || n_intv < 0 || (uint64_t)n_intv * 8 > buf.size() - p) {
                error = "damaged BAI index";
                return false;
            }
            linear[(size_t)r].resize((size_t)n_intv);
            if (n_intv) std::memcpy(linear[(size_t)r].data(), buf.data() + p, (size_t)n_intv * 8);
            p += (size_t)n_intv * 8;
        }
        have_index = true;
        return true;
    }
    template <typename T>
    T get() {
        T v;
        std::memcpy(&v, out.data() + out_pos, sizeof(T));
        out_pos += sizeof(T);
        return v;
    }
};

extern "C" {
#pragma GCC visibility push(default)

int fpt_bam_open(const char *path, fpt_bam **out) {
    if (!path || !out) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fpt_internal_fail(FPT_ERR_INVALID, "Cannot open BAM file: %s", path);  // cutcounts.py:103
    fpt_bam *b = new fpt_bam();
    b->f = f;
    b->n_threads = std::min(fpt_host_cpus(), 64);
    if (const char *e = getenv("FPT_BAM_THREADS")) b->n_threads = atoi(e) > 0 ? atoi(e) : 1;
    auto bad = [&](const char *what) {
        std::string msg = b->error.empty() ? what : b->error;
        fclose(f);
        delete b;
        return fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", path, msg.c_str());
    };
    if (!b->need(12) || std::memcmp(b->out.data() + b->out_pos, "BAM\1", 4) != 0) return bad("not a BAM file");
    b->out_pos += 4;
    const int32_t l_text = b->get<int32_t>();
    if (l_text < 0 || !b->need((size_t)l_text + 4)) return bad("truncated header");
    b->out_pos += (size_t)l_text;
    const int32_t n_ref = b->get<int32_t>();
    if (n_ref < 0) return bad("bad reference count");
    for (int i = 0; i < n_ref; ++i) {
        if (!b->need(4)) return bad("truncated reference list");
        const int32_t l_name = b->get<int32_t>();
        if (l_name <= 0 || l_name > (1 << 20) || !b->need((size_t)l_name + 4)) return bad("truncated reference list");
        b->ref_names.emplace_back((const char *)b->out.data() + b->out_pos, (size_t)l_name - 1);
        b->out_pos += (size_t)l_name;
        b->ref_lens.push_back(b->get<int32_t>());
    }
    if (!b->load_index(path)) return bad("damaged BAI index");
    *out = b;
    return FPT_OK;
}

int fpt_bam_has_index(fpt_bam *b, int32_t *yes_out) {
    if (!b || !yes_out) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    *yes_out = b->have_index ? 1 : 0;
    return FPT_OK;
}

int fpt_bam_seek_region(fpt_bam *b, int32_t ref_id, int64_t beg, int64_t end) {
    if (!b) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    if (!b->have_index) return fpt_internal_fail(FPT_ERR_INVALID, "the BAM file has no BAI index beside it");
    if (ref_id < 0 || ref_id >= (int32_t)b->ref_names.size()) return fpt_internal_fail(FPT_ERR_INVALID, "bad reference index");
    if (beg < 0) beg = 0;
    b->in_region = true;
    b->region_done = true;  // until an offset is found
    b->region_rid = ref_id;
    b->region_end = end;
    b->error.clear();
    if (end <= beg || (size_t)ref_id >= b->linear.size()) return FPT_OK;
    // first window at or after the region's start with an entry: alignments overlapping the start
    // window are at or after its entry; an empty window (0) means none overlap it, the next entry
    // holds the first alignment after it
    const std::vector<uint64_t> &lin = b->linear[(size_t)ref_id];
    uint64_t voff = 0;
    for (size_t w = (size_t)(beg >> 14); w < lin.size(); ++w)
        if (lin[w] != 0) {
            voff = lin[w];
            break;
        }
    if (voff == 0) return FPT_OK;  // nothing at or after the region on this reference
    if (fseeko(b->f, (off_t)(voff >> 16), SEEK_SET) != 0) return fpt_internal_fail(FPT_ERR_INVALID, "seek failed");
    b->eof = false;
    b->in_len = 0;
    b->out.clear();
    b->out_pos = 0;
    b->chunk = kSeekChunk;
    const size_t within = (size_t)(voff & 0xffff);
    if (!b->need(within + 1)) {  // the block the offset points into
        if (b->error.empty()) b->error = "BAI index points beyond the end of the file";
        return fpt_internal_fail(FPT_ERR_INVALID, "%s", b->error.c_str());
    }
    b->out_pos = within;
    b->region_done = false;
    return FPT_OK;
}

int fpt_bam_close(fpt_bam *b) {
    if (!b) return FPT_OK;
    if (b->f) fclose(b->f);
    delete b;
    return FPT_OK;
}

int fpt_bam_n_refs(fpt_bam *b, int32_t *n_out) {
    if (!b || !n_out) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    *n_out = (int32_t)b->ref_names.size();
    return FPT_OK;
}

int fpt_bam_ref(fpt_bam *b, int32_t i, char *name_out, int32_t cap, int64_t *len_out) {
    if (!b || i < 0 || i >= (int32_t)b->ref_names.size()) return fpt_internal_fail(FPT_ERR_INVALID, "bad reference index");
    if (name_out && cap > 0) {
        std::strncpy(name_out, b->ref_names[i].c_str(), (size_t)cap - 1);
        name_out[cap - 1] = 0;
    }
    if (len_out) *len_out = b->ref_lens[i];
    return FPT_OK;
}

int fpt_bam_read(fpt_bam *b, int64_t max_reads, int32_t *ref_id, int32_t *ref_start, int32_t *ref_end,
                 uint16_t *flag, uint8_t *mapq, int64_t *n_out) {
    if (!b || !n_out || max_reads < 0 || (max_reads > 0 && (!ref_id || !ref_start || !ref_end || !flag || !mapq)))
        return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    int64_t n = 0;
    while (n < max_reads) {
        if (b->in_region && b->region_done) break;
        if (!b->need(4)) break;  // end of file (or an error, reported below)
        const int32_t block = b->get<int32_t>();
        if (block < 32 || block > (1 << 28) || !b->need((size_t)block)) {
            if (b->error.empty()) b->error = "truncated or damaged alignment record";
            break;
        }
        const size_t rec = b->out_pos;
        const int32_t rid = b->get<int32_t>(), pos = b->get<int32_t>();
        if (b->in_region && (rid != b->region_rid || (int64_t)pos >= b->region_end)) {
            b->region_done = true;  // sorted by coordinate: nothing of the region follows
            b->out_pos = rec + (size_t)block;
            break;
        }
        const uint8_t l_name = b->get<uint8_t>(), mq = b->get<uint8_t>();
        (void)b->get<uint16_t>();  // bin
        const uint16_t n_cig = b->get<uint16_t>(), fl = b->get<uint16_t>();
        const int32_t l_seq = b->get<int32_t>();
        b->out_pos = rec + 32;  // next_refID, next_pos, tlen are not needed
        int64_t span = 0;
        if ((size_t)32 + l_name + 4u * (size_t)n_cig <= (size_t)block) {
            const unsigned char *cig = b->out.data() + rec + 32 + l_name;
            // a CIGAR of more than 65535 operations lives in the CG tag; the record then holds the
            // placeholder <l_seq>S<ref span>N, whose N is the reference span (SAM spec 4.2.2)
            for (int k = 0; k < n_cig; ++k) {
                uint32_t v;
                std::memcpy(&v, cig + 4 * k, 4);
                const uint32_t op = v & 0xf;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;  // M D N = X
            }
            (void)l_seq;
        }
        b->out_pos = rec + (size_t)block;
        ref_id[n] = rid;
        ref_start[n] = pos;
        // no reference-consuming operation: pysam's reference_end is None; -1 tells k_cut_counts
        // to leave a reverse read of that kind out (the reference would fail on it)
        const int64_t end = (int64_t)pos + span;
        ref_end[n] = span > 0 && end <= 0x7fffffff ? (int32_t)end : -1;
        flag[n] = fl;
        mapq[n] = mq;
        ++n;
    }
    *n_out = n;
    if (!b->error.empty()) return fpt_internal_fail(FPT_ERR_INVALID, "%s", b->error.c_str());
    return FPT_OK;
}

// The records themselves (SAM/BAM specification 4.2: every record behind its 4-byte block_size), for a caller that
// needs more of an alignment than its coordinates (name, mate flags, template length, bases, qualities, tags).  Same
// walk, same region rule and the same checks as fpt_bam_read; a record is handed over whole or not at all.  When
// the FIRST record does not fit `cap` the call fails with *n_out = 0 and *bytes_out = the bytes that record needs.
int fpt_bam_read_raw(fpt_bam *b, int64_t max_reads, uint8_t *buf, int64_t cap, int64_t *n_out, int64_t *bytes_out) {
    if (!b || !n_out || !bytes_out || max_reads < 0 || cap < 0 || (cap > 0 && !buf))
        return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    int64_t n = 0, used = 0;
    while (n < max_reads) {
        if (b->in_region && b->region_done) break;
        if (!b->need(4)) break;
        int32_t block;  // (looked at, not consumed: need() drops consumed bytes when it inflates more)
        std::memcpy(&block, b->out.data() + b->out_pos, 4);
        if (block < 32 || block > (1 << 28) || !b->need(4 + (size_t)block)) {
            if (b->error.empty()) b->error = "truncated or damaged alignment record";
            break;
        }
        const unsigned char *rec = b->out.data() + b->out_pos;
        int32_t rid, pos;
        std::memcpy(&rid, rec + 4, 4);
        std::memcpy(&pos, rec + 8, 4);
        if (b->in_region && (rid != b->region_rid || (int64_t)pos >= b->region_end)) {
            b->region_done = true;
            b->out_pos += 4 + (size_t)block;
            break;
        }
        if (used + 4 + (int64_t)block > cap) {  // no room: the record stays for the next call
            if (n == 0) {  // nothing read; *bytes_out says how large a buffer this record needs
                *n_out = 0;
                *bytes_out = 4 + (int64_t)block;
                return fpt_internal_fail(FPT_ERR_INVALID, "buffer of %lld bytes cannot hold a record of %d", (long long)cap, block);
            }
            break;
        }
        std::memcpy(buf + used, rec, 4 + (size_t)block);
        used += 4 + (int64_t)block;
        b->out_pos += 4 + (size_t)block;
        ++n;
    }
    *n_out = n;
    *bytes_out = used;
    if (!b->error.empty()) return fpt_internal_fail(FPT_ERR_INVALID, "%s", b->error.c_str());
    return FPT_OK;
}

#pragma GCC visibility pop
}
