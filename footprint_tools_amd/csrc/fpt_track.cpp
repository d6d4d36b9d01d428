// fpt_track.cpp -- indexed region access to per-nucleotide statistics tracks (host only).
//
// `ftd detect` writes bedGraph lines (cli/utils.py:119-144: chrom, start, start+1, exp, obs,
// -log p, -log win-p, fdr), the workflow compresses them with bgzip and indexes them with tabix,
// and the posterior caller reads them back per interval with pysam.TabixFile.fetch
// (cli/post.py:52-87).  pysam / htslib are not in this image; this is the library's own reader:
//   * the file is mapped, never read whole: a query inflates the BGZF members it needs;
//   * `<path>.tbi` is used when present (tabix's linear index: the virtual offset of the first
//     record of every 16 kb window); without it an index of the same shape is built by one pass
//     over the file at open;
//   * a batch of intervals is served by a team of threads, each walking its share of the (sorted or
//     not) interval list with a cache of the members it inflated last, parsing only the columns
//     asked for, and scattering the values into the caller's (bases) arrays -- what `_load_data`
//     does row by row in Python.  Where an interval starts at or a little beyond the row its
//     predecessor stopped at, the thread reads on from there instead of going through the index
//     (a sorted interval list is then one walk over the file per thread);
// Rows are selected by their start column: start <= row start < end (the rows of a `detect` track
// are one base wide, so this is tabix's overlap rule for them).
//
// Untrusted input: lengths and offsets from the file and the index are checked before use.
// PARITY UNPINNED for the reader (no htslib to compare with); tested on files the tests write.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/fpt.h"
#include "fpt_host_threads.hpp"
#include "fpt_bgzf.hpp"

int fpt_internal_fail(int code, const char *fmt, ...);  // fpt_capi.cpp

namespace {

constexpr int kWinShift = 14;  // tabix's linear index: 16 kb windows

struct mapped_file {
    int fd = -1;
    const unsigned char *p = nullptr;
    size_t n = 0;
    bool open(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const unsigned char *)m;
        return true;
    }
    void close() {
        if (p) munmap((void *)p, n);
        if (fd >= 0) ::close(fd);
        p = nullptr;
        fd = -1;
    }
};

}  // namespace

struct track_file {
    mapped_file file;
    bool bgzf = false;
    std::vector<std::string> names;                // chromosomes in file (or index) order
    std::unordered_map<std::string, int> name_id;
    std::vector<std::vector<uint64_t>> lin;        // per chromosome: virtual offset per 16 kb window (0 = none)
    int col_seq = 0, col_beg = 1;                  // 0-based columns of chromosome and start
    bool one_based = false;                        // start column counts from 1 (generic tabix presets)
    char meta = '#';
    int n_threads = 1;
};

namespace {

// A cursor over the lines of the track from a virtual offset on.  BGZF: (member offset << 16 |
// offset inside the inflated member); plain text: the byte offset itself.
struct line_reader {
    const track_file *t;
    // a small cache of inflated members (queries of neighbouring intervals hit the same ones)
    static constexpr int kCache = 4;
    size_t c_off[kCache];
    size_t c_next[kCache];
    uint32_t c_len[kCache];
    std::vector<unsigned char> c_buf[kCache];
    int c_age[kCache];
    int clock = 0;
    std::string carry;  // a line that straddles members
    // position
    size_t coff = 0;  // current member (BGZF) / byte offset (plain)
    int cur = -1;     // cache slot of the current member
    uint32_t upos = 0;
    bool failed = false;

    explicit line_reader(const track_file *tr) : t(tr) {
        for (int i = 0; i < kCache; ++i) {
            c_off[i] = (size_t)-1;
            c_age[i] = 0;
            c_len[i] = 0;
            c_next[i] = 0;
        }
    }
    // inflate (or find) the member at file offset `off`; false at the end of the file or on damage
    bool load(size_t off) {
        for (int i = 0; i < kCache; ++i)
            if (c_off[i] == off) {
                cur = i;
                c_age[i] = ++clock;
                return true;
            }
        if (off >= t->file.n) return false;
        fptz::bgzf_block b;
        const long sz = fptz::bgzf_parse_member(t->file.p + off, t->file.n - off, &b);
        if (sz <= 0) {
            failed = true;
            return false;
        }
        int slot = 0;
        for (int i = 1; i < kCache; ++i)
            if (c_age[i] < c_age[slot]) slot = i;
        c_buf[slot].resize(b.isize ? b.isize : 1);
        if (!fptz::inflate_block(t->file.p + off + b.cpos, b.clen, c_buf[slot].data(), b.isize, b.crc)) {
            failed = true;
            c_off[slot] = (size_t)-1;
            return false;
        }
        c_off[slot] = off;
        c_next[slot] = off + (size_t)sz;
        c_len[slot] = b.isize;
        c_age[slot] = ++clock;
        cur = slot;
        return true;
    }
    bool seek(uint64_t voff) {
        carry.clear();
        if (!t->bgzf) {
            coff = (size_t)voff;
            return coff <= t->file.n;
        }
        coff = (size_t)(voff >> 16);
        upos = (uint32_t)(voff & 0xffff);
        if (!load(coff)) return false;
        if (upos > c_len[cur]) {
            failed = true;
            return false;
        }
        return true;
    }
    // the virtual offset of the next line to be returned
    uint64_t tell() const { return t->bgzf ? ((uint64_t)coff << 16 | upos) : (uint64_t)coff; }
    // next line (without its newline); false at the end
    bool next(const char **line, size_t *len) {
        if (!t->bgzf) {
            if (coff >= t->file.n) return false;
            const unsigned char *s = t->file.p + coff;
            const unsigned char *e = (const unsigned char *)memchr(s, '\n', t->file.n - coff);
            const size_t l = e ? (size_t)(e - s) : t->file.n - coff;
            *line = (const char *)s;
            *len = (l && s[l - 1] == '\r') ? l - 1 : l;
            coff += l + (e ? 1 : 0);
            return true;
        }
        carry.clear();
        for (;;) {
            if (cur < 0 || c_off[cur] != coff) {
                if (!load(coff)) return !carry.empty() && (*line = carry.data(), *len = carry.size(), true);
            }
            const unsigned char *base = c_buf[cur].data();
            const uint32_t n = c_len[cur];
            if (upos < n) {
                const unsigned char *s = base + upos;
                const unsigned char *e = (const unsigned char *)memchr(s, '\n', n - upos);
                if (e) {
                    const size_t l = (size_t)(e - s);
                    upos += (uint32_t)l + 1;
                    if (carry.empty()) {
                        *line = (const char *)s;
                        *len = l;
                    } else {
                        carry.append((const char *)s, l);
                        *line = carry.data();
                        *len = carry.size();
                    }
                    if (*len && (*line)[*len - 1] == '\r') --*len;
                    if (upos >= n) {  // the next line starts in the next member
                        coff = c_next[cur];
                        upos = 0;
                    }
                    return true;
                }
                carry.append((const char *)s, n - upos);
            }
            coff = c_next[cur];
            upos = 0;
            if (coff >= t->file.n) {
                if (carry.empty()) return false;
                *line = carry.data();
                *len = carry.size();
                return true;
            }
        }
    }
};

// field k (0-based) of a tab-separated line; false if the line has fewer
inline bool field(const char *line, size_t len, int k, const char **f, size_t *fl) {
    const char *p = line, *end = line + len;
    for (int i = 0; i < k; ++i) {
        p = (const char *)memchr(p, '\t', (size_t)(end - p));
        if (!p) return false;
        ++p;
    }
    const char *q = (const char *)memchr(p, '\t', (size_t)(end - p));
    *f = p;
    *fl = q ? (size_t)(q - p) : (size_t)(end - p);
    return true;
}

inline bool parse_i64(const char *f, size_t n, int64_t *v) {
    auto r = std::from_chars(f, f + n, *v);
    return r.ec == std::errc() && r.ptr == f + n;
}

inline double parse_f64(const char *f, size_t n) {
    double v;
    auto r = std::from_chars(f, f + n, v);
    if (r.ec == std::errc() && r.ptr == f + n) return v;
    // what Python's float() also takes: nan / inf / -inf in any case, a leading '+'
    std::string s(f, n);
    char *e = nullptr;
    v = strtod(s.c_str(), &e);
    return (e && *e == 0 && !s.empty()) ? v : NAN;
}

// ---- the index: <path>.tbi, or one pass over the file

bool inflate_whole(const unsigned char *p, size_t n, std::vector<unsigned char> *out) {
    size_t off = 0;
    while (off < n) {
        fptz::bgzf_block b;
        const long sz = fptz::bgzf_parse_member(p + off, n - off, &b);
        if (sz <= 0) return false;
        const size_t at = out->size();
        out->resize(at + b.isize);
        if (!fptz::inflate_block(p + off + b.cpos, b.clen, out->data() + at, b.isize, b.crc)) return false;
        off += (size_t)sz;
        if (out->size() > ((size_t)1 << 31)) return false;
    }
    return true;
}

// tabix index (the TBI layout of the tabix manual / htslib): only the linear index is used
bool load_tbi(track_file *t, const char *path) {
    mapped_file f;
    if (!f.open(path) || f.n == 0) {
        f.close();
        return false;
    }
    std::vector<unsigned char> d;
    const bool ok = inflate_whole(f.p, f.n, &d);
    f.close();
    if (!ok || d.size() < 36 || memcmp(d.data(), "TBI\1", 4) != 0) return false;
    size_t q = 4;
    auto i32 = [&](int32_t *v) {
        if (q + 4 > d.size()) return false;
        memcpy(v, d.data() + q, 4);
        q += 4;
        return true;
    };
    int32_t n_ref, format, col_seq, col_beg, col_end, meta, skip, l_nm;
    if (!i32(&n_ref) || !i32(&format) || !i32(&col_seq) || !i32(&col_beg) || !i32(&col_end) || !i32(&meta) || !i32(&skip) ||
        !i32(&l_nm))
        return false;
    if (n_ref < 0 || n_ref > (1 << 24) || l_nm < 0 || q + (size_t)l_nm > d.size() || col_seq < 1 || col_beg < 1) return false;
    std::vector<std::string> names;
    for (size_t a = q, e = q + (size_t)l_nm; a < e;) {
        const size_t l = strnlen((const char *)d.data() + a, e - a);
        names.emplace_back((const char *)d.data() + a, l);
        a += l + 1;
    }
    q += (size_t)l_nm;
    if ((int)names.size() != n_ref) return false;
    std::vector<std::vector<uint64_t>> lin((size_t)n_ref);
    for (int r = 0; r < n_ref; ++r) {
        int32_t n_bin;
        if (!i32(&n_bin) || n_bin < 0) return false;
        for (int b = 0; b < n_bin; ++b) {
            int32_t n_chunk;
            q += 4;  // bin number
            if (!i32(&n_chunk) || n_chunk < 0 || q + (size_t)n_chunk * 16 > d.size()) return false;
            q += (size_t)n_chunk * 16;
        }
        int32_t n_intv;
        if (!i32(&n_intv) || n_intv < 0 || q + (size_t)n_intv * 8 > d.size()) return false;
        lin[(size_t)r].resize((size_t)n_intv);
        if (n_intv) memcpy(lin[(size_t)r].data(), d.data() + q, (size_t)n_intv * 8);
        q += (size_t)n_intv * 8;
        // Windows before a reference's first row: this library's writer (and the tabix manual's text) give
        // them the offset 0, htslib the offset of the reference's first row.  A search that starts at 0
        // walks every line of every reference before this one; the first non-zero entry is where the
        // reference starts, and nothing of it lies before that (any reference but the first in the file)
        std::vector<uint64_t> &v = lin[(size_t)r];
        size_t first = 0;
        while (first < v.size() && v[first] == 0) ++first;
        if (first < v.size() && r > 0)
            for (size_t i = 0; i < first; ++i) v[i] = v[first];
    }
    t->names = names;
    t->lin = lin;
    t->col_seq = col_seq - 1;
    t->col_beg = col_beg - 1;
    t->one_based = !(format & 0x10000);  // TBX_UCSC: zero-based, half-open (the BED preset)
    t->meta = (char)meta;
    for (int r = 0; r < n_ref; ++r) t->name_id[t->names[(size_t)r]] = r;
    return true;
}

// the same linear index from one pass over the lines
bool build_index(track_file *t, std::string *err) {
    line_reader rd(t);
    if (!rd.seek(0)) {
        if (t->file.n == 0) return true;
        *err = "not a BGZF / text track";
        return false;
    }
    const char *line;
    size_t len;
    int cur_ref = -1;
    std::string cur_name;
    for (;;) {
        const uint64_t voff = rd.tell();
        if (!rd.next(&line, &len)) break;
        if (len == 0 || line[0] == t->meta) continue;
        const char *f;
        size_t fl;
        int64_t beg;
        if (!field(line, len, t->col_seq, &f, &fl)) continue;
        if (cur_ref < 0 || fl != cur_name.size() || memcmp(f, cur_name.data(), fl) != 0) {
            cur_name.assign(f, fl);
            auto it = t->name_id.find(cur_name);
            if (it == t->name_id.end()) {
                cur_ref = (int)t->names.size();
                t->names.push_back(cur_name);
                t->name_id[cur_name] = cur_ref;
                t->lin.emplace_back();
            } else {
                cur_ref = it->second;  // (a chromosome in two stretches: the first stretch is what the index finds)
            }
        }
        const char *g;
        size_t gl;
        if (!field(line, len, t->col_beg, &g, &gl) || !parse_i64(g, gl, &beg) || beg < 0) continue;
        const size_t w = (size_t)(beg >> kWinShift);
        std::vector<uint64_t> &lv = t->lin[(size_t)cur_ref];
        if (w >= lv.size()) {
            if (w > ((size_t)1 << 26)) continue;  // positions beyond 2^40: not a genome
            lv.resize(w + 1, 0);
        }
        if (lv[w] == 0) lv[w] = voff + 1;  // +1: 0 means "no line" (a line can sit at virtual offset 0)
    }
    if (rd.failed) {
        *err = "corrupt BGZF block";
        return false;
    }
    // an empty window starts where the next non-empty one does; then drop the +1
    for (std::vector<uint64_t> &lv : t->lin) {
        uint64_t nxt = 0;
        for (size_t w = lv.size(); w-- > 0;) {
            if (lv[w] == 0) lv[w] = nxt; else nxt = lv[w];
        }
    }
    return true;
}

// From the index's offset for a 16 kb window towards the member that holds `start`: the members after
// the window's first are looked at -- their header for the size, the first KB of what they inflate to
// for the first whole line -- and passed over while that line is a row of the chromosome before
// `start`.  A window of a per-base track is ~14 members; without this a query inflates and walks the
// half of them that lie before its first row (~1 ms), with it the one or two it needs.  Returns the
// virtual offset to start reading at (the one given, or the first whole line of a later member).
inline uint64_t hop_members(const track_file *t, int ref, int64_t start, uint64_t voff) {
    const std::string &name = t->names[(size_t)ref];
    size_t off = (size_t)(voff >> 16);
    uint64_t best = voff;
    unsigned char head[1024];
    for (int hops = 0; hops < 64; ++hops) {
        if (off >= t->file.n) break;
        fptz::bgzf_block b;
        const long sz = fptz::bgzf_parse_member(t->file.p + off, t->file.n - off, &b);
        if (sz <= 0) break;
        const size_t nxt = off + (size_t)sz;
        if (nxt >= t->file.n) break;
        fptz::bgzf_block nb;
        if (fptz::bgzf_parse_member(t->file.p + nxt, t->file.n - nxt, &nb) <= 0) break;
        const uint32_t got = fptz::inflate_prefix(t->file.p + nxt + nb.cpos, nb.clen, head, (uint32_t)sizeof head);
        // the first whole line of the next member: after its first newline (what comes before may be
        // the tail of a line that began in this member)
        const unsigned char *nl = got ? (const unsigned char *)memchr(head, '\n', got) : nullptr;
        if (!nl) break;
        const char *line = (const char *)nl + 1;
        const size_t room = (size_t)(head + got - (const unsigned char *)line);
        const char *le = (const char *)memchr(line, '\n', room);
        if (!le) break;  // (a line longer than the look: read on from where we are)
        const size_t len = (size_t)(le - line);
        const char *f;
        size_t fl;
        int64_t beg;
        if (len == 0 || line[0] == t->meta || !field(line, len, t->col_seq, &f, &fl) || fl != name.size() ||
            memcmp(f, name.data(), fl) != 0 || !field(line, len, t->col_beg, &f, &fl) || !parse_i64(f, fl, &beg))
            break;
        if (t->one_based) beg -= 1;
        if (beg >= start) break;  // (strictly before: rows that share a start may lie on both sides of a member's edge)
        off = nxt;
        best = (uint64_t)nxt << 16 | (uint64_t)(line - (const char *)head);
    }
    return best;
}

// Where a scan stopped: the first row at or beyond its interval's end, read but not used.  A batch
// of sorted intervals goes on from there instead of through the index -- a 16 kb window of a
// per-base track is 16,384 rows in ~14 members, and a query that starts in the middle of one inflates
// and walks half of that before its first row (a batch of 160-base intervals spent 98 % of its time
// there: tools/bench_post_e2e.py).
struct row_cursor {
    bool valid = false;
    int ref = -1;
    int64_t beg = 0;
    const char *line = nullptr;  // inside the reader's buffers: good until the reader moves
    size_t len = 0;
};
constexpr int64_t kStreamAhead = 4096;  // rows worth walking over rather than seeking

// rows of [start, end) on chromosome `ref`: fn(beg, line, len) for each; false on damage
template <typename Fn>
bool scan_rows(const track_file *t, line_reader &rd, int ref, int64_t start, int64_t end, bool own_index, Fn fn,
               row_cursor *cur = nullptr) {
    if (ref < 0 || start >= end) return true;
    const char *line;
    size_t len;
    bool seen;
    if (cur && cur->valid && cur->ref == ref && cur->beg <= start && start - cur->beg <= kStreamAhead) {
        // on from the row the scan before stopped at
        cur->valid = false;
        seen = true;
        if (cur->beg >= end) {
            cur->valid = true;  // (still ahead of this interval too)
            return true;
        }
        if (cur->beg >= start) fn(cur->beg, cur->line, cur->len);
    } else {
        if (cur) cur->valid = false;
        const std::vector<uint64_t> &lv = t->lin[(size_t)ref];
        const int64_t s0 = start < 0 ? 0 : start;
        const size_t w = (size_t)(s0 >> kWinShift);
        if (w >= lv.size()) return true;
        uint64_t voff = lv[w];
        if (own_index) {
            if (voff == 0) return true;
            voff -= 1;
        } else if (voff == ~(uint64_t)0) {  // (an unset window of some writers) from the start of the file
            voff = 0;
        }
        if (t->bgzf) voff = hop_members(t, ref, start, voff);
        if (!rd.seek(voff)) return !rd.failed;
        seen = own_index;  // our own windows start inside the chromosome; a .tbi offset may precede its first row
    }
    const std::string &name = t->names[(size_t)ref];
    while (rd.next(&line, &len)) {
        if (len == 0 || line[0] == t->meta) continue;
        const char *f;
        size_t fl;
        if (!field(line, len, t->col_seq, &f, &fl)) continue;
        if (fl != name.size() || memcmp(f, name.data(), fl) != 0) {
            if (seen) break;  // the chromosome's rows are over
            continue;
        }
        seen = true;
        int64_t beg;
        if (!field(line, len, t->col_beg, &f, &fl) || !parse_i64(f, fl, &beg)) continue;
        if (t->one_based) beg -= 1;
        if (beg >= end) {
            if (cur) *cur = row_cursor{true, ref, beg, line, len};
            break;
        }
        if (beg >= start) fn(beg, line, len);
    }
    return !rd.failed;
}

}  // namespace

struct fpt_track {
    track_file t;
    bool own_index = true;
};

extern "C" {
#pragma GCC visibility push(default)

int fpt_track_open(const char *path, fpt_track **out) {
    if (!path || !out) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    *out = nullptr;
    fpt_track *h = new fpt_track();
    track_file &t = h->t;
    if (!t.file.open(path)) {
        t.file.close();
        delete h;
        return fpt_internal_fail(FPT_ERR_INVALID, "Cannot open track file: %s", path);
    }
    t.bgzf = t.file.n >= 2 && t.file.p[0] == 31 && t.file.p[1] == 139;
    t.n_threads = std::min(fpt_host_cpus(), 32);
    if (const char *e = getenv("FPT_TRACK_THREADS")) t.n_threads = atoi(e) > 0 ? atoi(e) : 1;
    std::string tbi = std::string(path) + ".tbi";
    if (t.bgzf && load_tbi(&t, tbi.c_str())) {
        h->own_index = false;
    } else {
        std::string err;
        if (!build_index(&t, &err)) {
            t.file.close();
            delete h;
            return fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", path, err.c_str());
        }
    }
    *out = h;
    return FPT_OK;
}

int fpt_track_close(fpt_track *h) {
    if (!h) return FPT_OK;
    h->t.file.close();
    delete h;
    return FPT_OK;
}

int fpt_track_n_refs(fpt_track *h, int32_t *n_out, int32_t *indexed_out) {
    if (!h || !n_out) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    *n_out = (int32_t)h->t.names.size();
    if (indexed_out) *indexed_out = h->own_index ? 0 : 1;
    return FPT_OK;
}

int fpt_track_ref(fpt_track *h, int32_t i, char *name_out, int32_t cap) {
    if (!h || i < 0 || i >= (int32_t)h->t.names.size() || !name_out || cap < 1)
        return fpt_internal_fail(FPT_ERR_INVALID, "bad reference index");
    std::strncpy(name_out, h->t.names[(size_t)i].c_str(), (size_t)cap - 1);
    name_out[cap - 1] = 0;
    return FPT_OK;
}

int fpt_track_fetch(fpt_track *h, int64_t n_iv, const char *const *chroms, const int64_t *starts,
                    const int64_t *ends, const int64_t *out_off, int32_t n_cols, const int32_t *cols,
                    double *const *out, double *present) {
    if (!h || n_iv < 0 || n_cols < 0 || (n_iv > 0 && (!chroms || !starts || !ends || !out_off)) ||
        (n_cols > 0 && (!cols || !out)))
        return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    for (int c = 0; c < n_cols; ++c)
        if (cols[c] < 0 || cols[c] > 4096 || !out[c]) return fpt_internal_fail(FPT_ERR_INVALID, "bad column");
    const track_file &t = h->t;
    std::vector<int> ref((size_t)n_iv, -1);
    for (int64_t i = 0; i < n_iv; ++i) {
        if (!chroms[i]) return fpt_internal_fail(FPT_ERR_INVALID, "null chromosome name");
        auto it = t.name_id.find(chroms[i]);
        if (it != t.name_id.end()) ref[(size_t)i] = it->second;
    }
    int nt = t.n_threads;
    if ((int64_t)nt > n_iv / 16) nt = (int)std::max<int64_t>(1, n_iv / 16);
    std::atomic<int> bad(0);
    auto work = [&](int64_t a, int64_t b) {
        line_reader rd(&t);
        row_cursor cur;  // sorted neighbours are read in one walk
        for (int64_t i = a; i < b && !bad.load(); ++i) {
            const int64_t s = starts[i], o = out_off[i];
            const bool ok = scan_rows(&t, rd, ref[(size_t)i], s, ends[i], h->own_index, [&](int64_t beg, const char *line, size_t len) {
                const int64_t j = o + (beg - s);
                for (int c = 0; c < n_cols; ++c) {
                    const char *f;
                    size_t fl;
                    out[c][j] = field(line, len, cols[c], &f, &fl) ? parse_f64(f, fl) : NAN;
                }
                if (present) present[j] = 1.0;
            }, &cur);
            if (!ok) bad.store(1);
        }
    };
    if (nt <= 1) {
        work(0, n_iv);
    } else {
        std::vector<std::thread> team;
        for (int k = 0; k < nt; ++k) team.emplace_back(work, n_iv * k / nt, n_iv * (k + 1) / nt);
        for (std::thread &th : team) th.join();
    }
    if (bad.load()) return fpt_internal_fail(FPT_ERR_INVALID, "corrupt BGZF block in the track");
    return FPT_OK;
}

int fpt_track_fetch_rows(fpt_track *h, const char *chrom, int64_t start, int64_t end, int32_t n_cols,
                         const int32_t *cols, int64_t cap, int64_t *pos_out, double *vals_out, int64_t *n_out) {
    if (!h || !chrom || !n_out || n_cols < 0 || cap < 0 || (cap > 0 && !pos_out) || (n_cols > 0 && (!cols || (cap > 0 && !vals_out))))
        return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    const track_file &t = h->t;
    auto it = t.name_id.find(chrom);
    int64_t n = 0;
    line_reader rd(&t);
    const bool ok = scan_rows(&t, rd, it == t.name_id.end() ? -1 : it->second, start, end, h->own_index,
                              [&](int64_t beg, const char *line, size_t len) {
                                  if (n < cap) {
                                      pos_out[n] = beg;
                                      for (int c = 0; c < n_cols; ++c) {
                                          const char *f;
                                          size_t fl;
                                          vals_out[n * n_cols + c] = field(line, len, cols[c], &f, &fl) ? parse_f64(f, fl) : NAN;
                                      }
                                  }
                                  ++n;
                              });
    if (!ok) return fpt_internal_fail(FPT_ERR_INVALID, "corrupt BGZF block in the track");
    *n_out = n;  // rows found (may exceed cap: call again with room for them)
    return FPT_OK;
}

#pragma GCC visibility pop
}
