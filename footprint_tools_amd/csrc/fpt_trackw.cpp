// fpt_trackw.cpp -- writer of bgzip-compressed, tabix-indexed statistics tracks (host only).
//
// `ftd detect` writes bedGraph text (cli/utils.py:119-144); the posterior caller reads such tracks
// back per interval through pysam.TabixFile (cli/post.py:52-87), which needs them bgzip-compressed
// and indexed -- in the reference's workflow by the external `bgzip` and `tabix -p bed`.  Neither
// tool is in this image, so the library makes both files itself: the text is cut every kBlock
// bytes of the uncompressed stream into BGZF members (SAM/BAM specification 4.1; lines may straddle
// members), a group of members is deflated on a team of threads, and the tabix index (the TBI layout
// of the tabix manual: binning index + linear index of 16 kb windows, BED preset, itself a BGZF
// file) is accumulated on the way -- in positions of the uncompressed stream, turned into virtual
// offsets at the end, when every member's place in the file is known.  Consecutive lines of one bin
// merge into one chunk, so a per-base track of a whole genome keeps a few chunks per 16 kb.
//
// PARITY UNPINNED (no htslib here): tested by reading the files back with this library's own
// reader, by comparing the index with the one tests/tbiwriter.py makes of the same text, and by
// decompressing with Python's gzip module.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/fpt.h"
#include "fpt_host_threads.hpp"
#include "fpt_text_internal.hpp"

int fpt_internal_fail(int code, const char *fmt, ...);  // fpt_capi.cpp

namespace {

constexpr size_t kBlock = 0xff00;       // uncompressed bytes per member (what bgzip uses)
constexpr size_t kGroup = 256;          // members deflated together
constexpr int kMetaChar = '#';

// one BGZF member holding data[0..n): header with the BC subfield, raw deflate, CRC32, ISIZE
bool bgzf_member(const unsigned char *data, size_t n, std::vector<unsigned char> &out) {
    out.resize(18 + compressBound((uLong)n) + 8 + 64);
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char *>(data);
    zs.avail_in = (uInt)n;
    zs.next_out = out.data() + 18;
    zs.avail_out = (uInt)(out.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    const size_t total = 18 + clen + 8;
    if (total > 0x10000) return false;  // (cannot happen: 0xff00 bytes grow by at most a few dozen)
    const unsigned char head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0,
                                    (unsigned char)((total - 1) & 0xff), (unsigned char)((total - 1) >> 8)};
    std::memcpy(out.data(), head, 18);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isize = (uint32_t)n;
    std::memcpy(out.data() + 18 + clen, &crc, 4);
    std::memcpy(out.data() + 18 + clen + 4, &isize, 4);
    out.resize(total);
    return true;
}

// the same member from a deflate state that is kept between members (one per thread of a team:
// deflateInit2 allocates and clears ~260 KB) and a scratch buffer; `out` gets exactly the member
struct deflater {
    z_stream zs;
    bool live = false;
    std::vector<unsigned char> scratch;
    ~deflater() {
        if (live) deflateEnd(&zs);
    }
    int level = 6;
    bool member(const unsigned char *data, size_t n, std::vector<unsigned char> &out) {
        if (!live) {
            std::memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
            live = true;
            scratch.resize(18 + compressBound((uLong)kBlock) + 8 + 64);
        } else if (deflateReset(&zs) != Z_OK) {
            return false;
        }
        zs.next_in = const_cast<unsigned char *>(data);
        zs.avail_in = (uInt)n;
        zs.next_out = scratch.data() + 18;
        zs.avail_out = (uInt)(scratch.size() - 18 - 8);
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return false;
        const size_t clen = zs.total_out, total = 18 + clen + 8;
        if (total > 0x10000) return false;
        const unsigned char head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0,
                                        (unsigned char)((total - 1) & 0xff), (unsigned char)((total - 1) >> 8)};
        std::memcpy(scratch.data(), head, 18);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isize = (uint32_t)n;
        std::memcpy(scratch.data() + 18 + clen, &crc, 4);
        std::memcpy(scratch.data() + 18 + clen + 4, &isize, 4);
        out.assign(scratch.begin(), scratch.begin() + (long)total);
        return true;
    }
};

// pieces of text that follow each other in the uncompressed stream
struct segment {
    const unsigned char *p;
    size_t n;
};

int reg2bin(int64_t beg, int64_t end) {  // the binning scheme of the SAM specification 5.3
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct chunk {
    uint64_t u0, u1;  // positions in the uncompressed stream: [start of the first line, end of the last)
};
struct ref_index {
    std::string name;
    std::map<uint32_t, std::vector<chunk>> bins;
    std::vector<uint64_t> linear;  // per 16 kb window: start of the first line overlapping it, or ~0
};

}  // namespace

struct fpt_track_writer {
    FILE *f = nullptr;
    std::string path, error;
    int n_threads = 1;
    int level = 6;                       // zlib level of the data members (bgzip's default)
    std::vector<unsigned char> pending;  // text not yet cut into members (less than one group)
    std::string carry;                   // an unfinished line at the end of the last write call
    uint64_t upos = 0;                   // bytes of the uncompressed stream handed over so far
    uint64_t cpos = 0;                   // bytes of the compressed file written so far
    std::vector<uint64_t> member_at;     // compressed offset of every member
    std::vector<ref_index> refs;
    int cur_ref = -1;
    int64_t last_beg = -1;
    int64_t last_bin = -1;                // the bin of the line before and its chunk list (most lines stay in it)
    std::vector<chunk> *last_chunks = nullptr;

    bool flush_members(bool all) { return flush_segments({}, all, nullptr); }

    // `pending` followed by `more`, cut into members of kBlock bytes (a last shorter one when `all`),
    // deflated on a team of threads straight from where the text lies (a member that straddles two
    // pieces is put together in the thread's own buffer) and written in order; what is left over
    // becomes the new `pending`.  `meanwhile` runs on the calling thread while the team works
    // (the index of the same lines); if it returns false nothing is written.
    template <typename F>
    bool flush_segments(const std::vector<segment> &more, bool all, F meanwhile) {
        std::vector<segment> segs;
        std::vector<size_t> at;  // start of each piece in the joined text
        size_t total = 0;
        auto add = [&](const unsigned char *p, size_t n) {
            if (n == 0) return;
            segs.push_back(segment{p, n});
            at.push_back(total);
            total += n;
        };
        add(pending.data(), pending.size());
        for (const segment &sg : more) add(sg.p, sg.n);
        const size_t n_mem = all ? (total + kBlock - 1) / kBlock : total / kBlock;
        const size_t done = std::min(total, n_mem * kBlock);
        std::vector<std::vector<unsigned char>> outs(n_mem);
        std::atomic<size_t> next(0);
        std::atomic<int> bad(0);
        auto work = [&]() {
            deflater d;
            d.level = level;
            std::vector<unsigned char> joined;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n_mem) return;
                const size_t a = i * kBlock, len = std::min(kBlock, total - a);
                size_t k = (size_t)(std::upper_bound(at.begin(), at.end(), a) - at.begin()) - 1;
                const unsigned char *src = segs[k].p + (a - at[k]);
                if (a + len > at[k] + segs[k].n) {  // straddles pieces
                    joined.resize(len);
                    size_t got = 0;
                    for (size_t u = a; got < len; ++k) {
                        const size_t take = std::min(len - got, at[k] + segs[k].n - u);
                        std::memcpy(joined.data() + got, segs[k].p + (u - at[k]), take);
                        got += take;
                        u += take;
                    }
                    src = joined.data();
                }
                if (!d.member(src, len, outs[i])) bad.store(1);
            }
        };
        const int nt = (int)std::min<size_t>((size_t)n_threads, n_mem);
        bool ok = true;
        if (nt <= 1) {
            if constexpr (!std::is_same<F, std::nullptr_t>::value) ok = meanwhile();
            if (ok) work();
        } else {
            std::vector<std::thread> team;
            for (int t = 0; t < nt; ++t) team.emplace_back(work);
            if constexpr (!std::is_same<F, std::nullptr_t>::value) ok = meanwhile();
            for (std::thread &t : team) t.join();
        }
        if (!ok) return false;
        if (bad.load()) {
            error = "deflate failed";
            return false;
        }
        for (size_t i = 0; i < n_mem; ++i) {
            member_at.push_back(cpos);
            if (fwrite(outs[i].data(), 1, outs[i].size(), f) != outs[i].size()) {
                error = "write failed";
                return false;
            }
            cpos += outs[i].size();
        }
        // the tail: the bytes after the last whole member
        std::vector<unsigned char> rest;
        rest.reserve(total - done);
        for (size_t k = 0; k < segs.size(); ++k) {
            if (at[k] + segs[k].n <= done) continue;
            const size_t from = done > at[k] ? done - at[k] : 0;
            rest.insert(rest.end(), segs[k].p + from, segs[k].p + segs[k].n);
        }
        pending.swap(rest);
        return true;
    }

    // one whole line (without its newline) that starts at position `at` of the uncompressed stream
    bool index_line(const char *s, size_t n, uint64_t at) {
        if (n == 0 || s[0] == kMetaChar) return true;
        const char *e = s + n;
        const char *t1 = (const char *)std::memchr(s, '\t', n);
        if (!t1) return bad_line();
        const char *t2 = (const char *)std::memchr(t1 + 1, '\t', (size_t)(e - t1 - 1));
        if (!t2) return bad_line();
        const char *t3 = (const char *)std::memchr(t2 + 1, '\t', (size_t)(e - t2 - 1));
        if (!t3) t3 = e;
        int64_t beg = 0, end = 0;
        if (std::from_chars(t1 + 1, t2, beg).ptr != t2 || std::from_chars(t2 + 1, t3, end).ptr != t3) return bad_line();
        return index_known(s, (size_t)(t1 - s), beg, end, at, n);
    }
    // ... when its fields are known: chromosome s[0..ln), [beg, end), n bytes without the newline
    bool index_known(const char *s, size_t ln, int64_t beg, int64_t end, uint64_t at, size_t n) {
        if (beg < 0 || end <= beg || end > ((int64_t)1 << 29)) return bad_line();
        if (cur_ref < 0 || refs[(size_t)cur_ref].name.size() != ln || std::memcmp(refs[(size_t)cur_ref].name.data(), s, ln)) {
            for (const ref_index &r : refs)
                if (r.name.size() == ln && !std::memcmp(r.name.data(), s, ln)) {
                    error = "the lines of chromosome " + r.name + " are not contiguous";
                    return false;
                }
            refs.emplace_back();
            refs.back().name.assign(s, ln);
            cur_ref = (int)refs.size() - 1;
            last_beg = -1;
            last_bin = -1;
        }
        if (beg < last_beg) {
            error = "the lines are not sorted by position";
            return false;
        }
        last_beg = beg;
        ref_index &r = refs[(size_t)cur_ref];
        const uint64_t next_line = at + n + 1;
        const int bin = reg2bin(beg, end);
        if (bin != last_bin) {  // (a map lookup per line is most of what a line costs)
            last_chunks = &r.bins[(uint32_t)bin];
            last_bin = bin;
        }
        std::vector<chunk> &cs = *last_chunks;
        if (!cs.empty() && cs.back().u1 == at) cs.back().u1 = next_line;
        else cs.push_back(chunk{at, next_line});
        const size_t w1 = (size_t)((end - 1) >> 14);
        if (r.linear.size() <= w1) r.linear.resize(w1 + 1, ~(uint64_t)0);
        for (size_t w = (size_t)(beg >> 14); w <= w1; ++w)
            if (r.linear[w] == ~(uint64_t)0) r.linear[w] = at;
        return true;
    }
    bool bad_line() {
        error = "a line is not <chrom> TAB <start> TAB <end> [TAB ...] with 0 <= start < end <= 2^29";
        return false;
    }

    // virtual offset of a position of the uncompressed stream (members are kBlock bytes each)
    uint64_t voff(uint64_t u, uint64_t eof_at) const {
        const uint64_t k = u / kBlock;
        if (k >= member_at.size()) return eof_at << 16;  // the end of the data: the end-of-file member
        return (member_at[(size_t)k] << 16) | (u % kBlock);
    }
};

extern "C" {
#pragma GCC visibility push(default)

int fpt_track_writer_open(const char *path, fpt_track_writer **out) {
    if (!path || !out) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    *out = nullptr;
    FILE *f = fopen(path, "wb");
    if (!f) return fpt_internal_fail(FPT_ERR_INVALID, "cannot create %s", path);
    fpt_track_writer *w = new fpt_track_writer();
    w->f = f;
    w->path = path;
    w->n_threads = std::min(fpt_host_cpus(), 256);  // (deflate is ~20 MB/s per thread on this text)
    if (const char *e = getenv("FPT_TRACK_THREADS")) w->n_threads = atoi(e) > 0 ? atoi(e) : 1;
    *out = w;
    return FPT_OK;
}

int fpt_track_writer_set_level(fpt_track_writer *w, int32_t level) {
    if (!w || level < 0 || level > 9) return fpt_internal_fail(FPT_ERR_INVALID, "compression level must be 0..9");
    w->level = level;
    return FPT_OK;
}

int fpt_track_writer_write(fpt_track_writer *w, const char *text, int64_t n_bytes) {
    if (!w || n_bytes < 0 || (n_bytes > 0 && !text)) return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    if (!w->error.empty()) return fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", w->path.c_str(), w->error.c_str());
    // index the whole lines (an unfinished last line waits for the next call)
    const char *p = text, *e = text + n_bytes;
    uint64_t at = w->upos - w->carry.size();
    while (p < e) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(e - p));
        if (!nl) {
            w->carry.append(p, (size_t)(e - p));
            break;
        }
        bool ok;
        if (!w->carry.empty()) {
            w->carry.append(p, (size_t)(nl - p));
            ok = w->index_line(w->carry.data(), w->carry.size(), at);
            at += w->carry.size() + 1;
            w->carry.clear();
        } else {
            ok = w->index_line(p, (size_t)(nl - p), at);
            at += (uint64_t)(nl - p) + 1;
        }
        if (!ok) return fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", w->path.c_str(), w->error.c_str());
        p = nl + 1;
    }
    w->pending.insert(w->pending.end(), (const unsigned char *)text, (const unsigned char *)text + n_bytes);
    w->upos += (uint64_t)n_bytes;
    if (w->pending.size() >= kGroup * kBlock && !w->flush_members(false))
        return fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", w->path.c_str(), w->error.c_str());
    return FPT_OK;
}

int fpt_track_writer_write_stats(fpt_track_writer *w, int64_t n_intervals, const char *const *chrom_names, int32_t n_chroms,
                                 const int32_t *chrom_id, const int64_t *start, const int64_t *row_off, const double *stats,
                                 int32_t n_cols, int32_t precision) {
    if (!w) return fpt_internal_fail(FPT_ERR_INVALID, "null writer");
    if (!w->error.empty()) return fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", w->path.c_str(), w->error.c_str());
    static const bool times = getenv("FPT_TRACK_TIMES") != nullptr;  // diagnostic: where a call spends its time
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<fpt_text_part> parts;
    if (int rc = fpt_internal_format_batch(n_intervals, chrom_names, n_chroms, chrom_id, start, row_off, stats, n_cols, '\t',
                                           precision, true, parts))
        return rc;
    const auto t1 = std::chrono::steady_clock::now();
    bool lines = w->carry.empty();  // (after half a line of somebody's text: as text)
    for (const fpt_text_part &pt : parts) lines = lines && pt.line_len.size() == (size_t)(row_off[pt.j1] - row_off[pt.j0]);
    if (!lines) {
        for (const fpt_text_part &pt : parts)
            if (int rc = fpt_track_writer_write(w, pt.data.get(), (int64_t)pt.size)) return rc;
        return FPT_OK;
    }
    // The lines are indexed from what is known about them (position = start + row, length as the
    // formatter recorded it) while the team deflates the members they fall into.
    std::vector<segment> segs;
    size_t total = 0;
    for (const fpt_text_part &pt : parts) {
        segs.push_back(segment{(const unsigned char *)pt.data.get(), pt.size});
        total += pt.size;
    }
    std::chrono::steady_clock::time_point t2 = t1;
    auto index_all = [&]() {
        struct stamp {
            std::chrono::steady_clock::time_point &t;
            ~stamp() { t = std::chrono::steady_clock::now(); }
        } st{t2};
        uint64_t at = w->upos;
        for (const fpt_text_part &pt : parts) {
            const uint16_t *ll = pt.line_len.data();
            for (int64_t j = pt.j0; j < pt.j1; ++j) {
                const char *name = chrom_names[chrom_id[j]];
                const size_t ln = std::strlen(name);
                const int64_t n = row_off[j + 1] - row_off[j];
                for (int64_t i = 0; i < n; ++i) {
                    const size_t len = *ll++;
                    if (!w->index_known(name, ln, start[j] + i, start[j] + i + 1, at, len - 1)) return false;
                    at += len;
                }
            }
        }
        return true;
    };
    const bool ok = w->flush_segments(segs, false, index_all);
    if (!ok) return fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", w->path.c_str(), w->error.c_str());
    w->upos += (uint64_t)total;
    if (times) {
        const auto t3 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        fprintf(stderr, "fpt_track_writer_write_stats: %zu bytes: format %.1f ms, index %.1f ms, deflate + write (index inside) %.1f ms\n",
                total, ms(t0, t1), ms(t1, t2), ms(t1, t3));
    }
    return FPT_OK;
}

int fpt_track_writer_close(fpt_track_writer *w) {
    if (!w) return FPT_OK;
    int rc = FPT_OK;
    std::string msg = w->error;
    if (msg.empty() && !w->carry.empty()) msg = "the text does not end with a newline";
    if (msg.empty() && !w->flush_members(true)) msg = w->error;
    const uint64_t eof_at = w->cpos;
    std::vector<unsigned char> eof;
    if (msg.empty() && (!bgzf_member(nullptr, 0, eof) || fwrite(eof.data(), 1, eof.size(), w->f) != eof.size()))
        msg = "write failed";
    if (w->f && fclose(w->f) != 0 && msg.empty()) msg = "write failed";
    if (msg.empty()) {
        // the tabix index: header, names, then per reference the bins and the linear index
        std::vector<unsigned char> idx;
        auto put32 = [&](int32_t v) { idx.insert(idx.end(), (unsigned char *)&v, (unsigned char *)&v + 4); };
        auto put64 = [&](uint64_t v) { idx.insert(idx.end(), (unsigned char *)&v, (unsigned char *)&v + 8); };
        idx.insert(idx.end(), {'T', 'B', 'I', 1});
        std::string names;
        for (const ref_index &r : w->refs) names += r.name + '\0';
        put32((int32_t)w->refs.size());
        put32(0x10000);  // zero-based half-open coordinates (the BED preset)
        put32(1), put32(2), put32(3), put32(kMetaChar), put32(0), put32((int32_t)names.size());
        idx.insert(idx.end(), names.begin(), names.end());
        for (const ref_index &r : w->refs) {
            put32((int32_t)r.bins.size());
            for (const auto &b : r.bins) {
                put32((int32_t)b.first);
                put32((int32_t)b.second.size());
                for (const chunk &c : b.second) put64(w->voff(c.u0, eof_at)), put64(w->voff(c.u1, eof_at));
            }
            put32((int32_t)r.linear.size());
            uint64_t prev = 0;  // (htslib: a window without a line takes the offset of the window before)
            for (uint64_t u : r.linear) {
                if (u != ~(uint64_t)0) prev = w->voff(u, eof_at);
                put64(prev);
            }
        }
        FILE *g = fopen((w->path + ".tbi").c_str(), "wb");
        if (!g) {
            msg = "cannot create the index";
        } else {
            std::vector<unsigned char> mem;
            bool ok = true;
            for (size_t a = 0; a < idx.size() && ok; a += kBlock)
                ok = bgzf_member(idx.data() + a, std::min(kBlock, idx.size() - a), mem) &&
                     fwrite(mem.data(), 1, mem.size(), g) == mem.size();
            ok = ok && fwrite(eof.data(), 1, eof.size(), g) == eof.size();
            if (fclose(g) != 0 || !ok) msg = "writing the index failed";
        }
    }
    if (!msg.empty()) rc = fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", w->path.c_str(), msg.c_str());
    delete w;
    return rc;
}

#pragma GCC visibility pop
}
