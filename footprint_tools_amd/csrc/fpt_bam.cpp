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

int fpt_internal_fail(int code, const char *fmt, ...);  // fpt_capi.cpp

#include "fpt_bgzf.hpp"

namespace {

using fptz::bgzf_block;
using fptz::bgzf_member_size;
using fptz::kMaxBlock;
constexpr size_t kChunk = (size_t)32 << 20;  // compressed bytes taken from the file at a time

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
            if (!eof && in_len < kChunk) {
                if (in.size() < kChunk + kMaxBlock) in.resize(kChunk + kMaxBlock);
                const size_t got = fread(in.data() + in_len, 1, in.size() - in_len, f);
                in_len += got;
                if (got == 0) eof = true;
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
    unsigned hc = std::thread::hardware_concurrency();
    b->n_threads = (int)(hc == 0 ? 1 : (hc > 64 ? 64 : hc));
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
    *out = b;
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
        if (!b->need(4)) break;  // end of file (or an error, reported below)
        const int32_t block = b->get<int32_t>();
        if (block < 32 || block > (1 << 28) || !b->need((size_t)block)) {
            if (b->error.empty()) b->error = "truncated or damaged alignment record";
            break;
        }
        const size_t rec = b->out_pos;
        const int32_t rid = b->get<int32_t>(), pos = b->get<int32_t>();
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

#pragma GCC visibility pop
}
