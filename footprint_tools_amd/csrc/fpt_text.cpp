// fpt_text.cpp -- host-side text of the per-nucleotide output (cli/utils.py:119-163,
// write_stats_to_output): once the statistics take milliseconds, formatting "{:0.4f}" of five
// columns per base in Python is what a `detect` run waits for.
//
// "{:0.Nf}".format(v) is the correctly rounded decimal expansion of the double's exact value
// (round-half-even on exact ties), sign kept for negative zero and for negatives that round to
// zero, "nan" / "inf" / "-inf" otherwise.  For N <= 9 and |v| * 10^N < 9e18 that is integer arithmetic:
// v = m * 2^e exactly, so v * 10^N = (m * 10^N) >> -e with the remainder deciding the rounding --
// 128-bit integers hold it.  Everything else goes through snprintf, which glibc also rounds
// correctly.  Tested against Python's own formatting on random and boundary values.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fpt.h"
#include "fpt_host_threads.hpp"
#include "fpt_text_internal.hpp"

int fpt_internal_fail(int code, const char *fmt, ...);  // fpt_capi.cpp

namespace {

const uint64_t kPow10[10] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull,
                             100000000ull, 1000000000ull};

inline char *put_uint(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

inline char *put_int(char *p, int64_t v) {
    if (v < 0) {
        *p++ = '-';
        return put_uint(p, (uint64_t)0 - (uint64_t)v);
    }
    return put_uint(p, (uint64_t)v);
}

// appends "{:0.<prec>f}".format(v); at most 40 bytes on the fast path
inline char *put_fixed(char *p, char *end, double v, int prec) {
    if (std::isnan(v)) {
        memcpy(p, "nan", 3);
        return p + 3;
    }
    if (std::signbit(v)) *p++ = '-';
    const double a = std::fabs(v);
    if (std::isinf(a)) {
        memcpy(p, "inf", 3);
        return p + 3;
    }
    if (prec <= 9 && a < 4503599627370496.0 && a * (double)kPow10[prec] < 9.0e18) {  // < 2^52, and the scaled integer fits 64 bits
        int e;
        const double fr = std::frexp(a, &e);             // a = fr * 2^e, fr in [0.5, 1) (or 0)
        const uint64_t m = (uint64_t)std::ldexp(fr, 53);  // 53-bit integer: a = m * 2^(e - 53)
        const int sh = 53 - e;                            // >= 1 here: a < 2^52
        unsigned __int128 x = (unsigned __int128)m * kPow10[prec];
        uint64_t q;
        if (sh >= 128) {
            q = 0;  // m * 10^prec < 2^83: far below half a unit
        } else {
            const unsigned __int128 one = (unsigned __int128)1 << sh, rem = x & (one - 1), half = one >> 1;
            q = (uint64_t)(x >> sh);
            if (rem > half || (rem == half && (q & 1))) ++q;
        }
        p = put_uint(p, q / kPow10[prec]);
        if (prec > 0) {
            *p++ = '.';
            uint64_t f = q % kPow10[prec];
            for (int i = prec - 1; i >= 0; --i) {
                p[i] = (char)('0' + f % 10);
                f /= 10;
            }
            p += prec;
        }
        return p;
    }
    const int n = snprintf(p, (size_t)(end - p), "%.*f", prec, a);
    return (n < 0 || n >= end - p) ? nullptr : p + n;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

// rows [r0, r1) of the selection into [p, end): the new end, or nullptr when the buffer is too
// small / a row index is outside the matrix (*bad_row set)
static char *format_rows(const char *chrom, size_t lc, int64_t start, const double *stats, int64_t n_rows, int32_t n_cols,
                         const int64_t *rows, int64_t r0, int64_t r1, char delim, int32_t precision, char *p, char *end,
                         int64_t *bad_row, uint16_t *line_len = nullptr) {
    for (int64_t r = r0; r < r1; ++r) {
        const char *const line = p;
        const int64_t i = rows ? rows[r] : r;
        if (i < 0 || i >= n_rows) {
            *bad_row = i;
            return nullptr;
        }
        if ((size_t)(end - p) < lc + 48) return nullptr;
        memcpy(p, chrom, lc);
        p += lc;
        *p++ = delim;
        p = put_int(p, start + i);
        *p++ = delim;
        p = put_int(p, start + i + 1);
        *p++ = delim;
        const double *row = stats + i * (int64_t)n_cols;
        for (int32_t c = 0; c < n_cols; ++c) {
            if (end - p < 48) return nullptr;
            if (c) *p++ = delim;
            p = put_fixed(p, end - 2, row[c], precision);
            if (!p) return nullptr;
        }
        *p++ = '\n';
        if (line_len) line_len[r - r0] = (uint16_t)(p - line);
    }
    return p;
}

int fpt_format_stats(const char *chrom, int64_t start, const double *stats, int64_t n_rows, int32_t n_cols,
                     const int64_t *rows, int64_t n_sel, char delim, int32_t precision, char *buf, int64_t cap,
                     int64_t *len_out) {
    if (!chrom || !len_out || n_rows < 0 || n_cols < 0 || precision < 0 || precision > 30 || cap < 0 || (!buf && cap > 0))
        return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    if ((!stats && n_rows > 0 && n_cols > 0) || (rows && n_sel < 0)) return fpt_internal_fail(FPT_ERR_INVALID, "null matrix");
    const size_t lc = strlen(chrom);
    const int64_t count = rows ? n_sel : n_rows;
    int64_t bad_row = -1;
    // large selections: row blocks on a team of threads, each into its own buffer, joined in order
    // (a line is 30-70 bytes; the text of a million rows is what a `detect` run waits for)
    int nt = std::min(fpt_host_cpus(), 32);
    if (const char *e = getenv("FPT_TEXT_THREADS")) nt = atoi(e) > 0 ? atoi(e) : 1;
    if (count < 8192 || nt < 2) {
        char *p = format_rows(chrom, lc, start, stats, n_rows, n_cols, rows, 0, count, delim, precision, buf, buf + cap, &bad_row);
        if (!p) {
            if (bad_row != -1) return fpt_internal_fail(FPT_ERR_INVALID, "row %lld outside the matrix", (long long)bad_row);
            return fpt_internal_fail(FPT_ERR_INVALID, "output buffer too small");
        }
        *len_out = p - buf;
        return FPT_OK;
    }
    if ((int64_t)nt > count / 4096) nt = (int)(count / 4096);
    // worst case of one line on the fast path; values beyond 1e17 take the snprintf path, whose
    // lines can be longer: a block that overflows its private buffer is redone with more room
    const size_t per_line = lc + 44 + (size_t)n_cols * (size_t)(22 + precision);
    std::vector<std::string> parts((size_t)nt);
    std::vector<int64_t> bad((size_t)nt, -1);
    std::vector<int> fail((size_t)nt, 0);
    std::vector<std::thread> team;
    for (int t = 0; t < nt; ++t)
        team.emplace_back([&, t]() {
            const int64_t r0 = count * t / nt, r1 = count * (t + 1) / nt;
            size_t room = (size_t)(r1 - r0) * per_line + 64;
            for (int attempt = 0; attempt < 2; ++attempt) {
                parts[(size_t)t].resize(room);
                char *b0 = &parts[(size_t)t][0];
                char *p = format_rows(chrom, lc, start, stats, n_rows, n_cols, rows, r0, r1, delim, precision, b0, b0 + room,
                                      &bad[(size_t)t]);
                if (p) {
                    parts[(size_t)t].resize((size_t)(p - b0));
                    return;
                }
                if (bad[(size_t)t] != -1) break;
                room = (size_t)(r1 - r0) * (lc + 44 + (size_t)n_cols * (size_t)(330 + precision)) + 64;
            }
            fail[(size_t)t] = 1;
        });
    for (std::thread &t : team) t.join();
    size_t total = 0;
    for (int t = 0; t < nt; ++t) {
        if (bad[(size_t)t] != -1) return fpt_internal_fail(FPT_ERR_INVALID, "row %lld outside the matrix", (long long)bad[(size_t)t]);
        if (fail[(size_t)t]) return fpt_internal_fail(FPT_ERR_INVALID, "output buffer too small");
        total += parts[(size_t)t].size();
    }
    if ((int64_t)total > cap) return fpt_internal_fail(FPT_ERR_INVALID, "output buffer too small");
    char *p = buf;
    for (int t = 0; t < nt; ++t) {
        memcpy(p, parts[(size_t)t].data(), parts[(size_t)t].size());
        p += parts[(size_t)t].size();
    }
    *len_out = (int64_t)total;
    return FPT_OK;
}

// ---- a whole batch of intervals at once (detect's batch_iter: one (bases, n_cols) matrix, the rows
// of interval j at [row_off[j], row_off[j+1])).  Formatting interval by interval from Python costs
// ~20 us of interpreter per interval on top of the text; here intervals are dealt to a team of
// threads by rows and the parts come back in order.
#pragma GCC visibility pop
}  // extern "C"

size_t fpt_internal_line_bound(size_t chrom_len, int32_t n_cols, int32_t precision) {
    return chrom_len + 44 + (size_t)n_cols * (size_t)(330 + precision);
}

int fpt_internal_format_batch(int64_t n_intervals, const char *const *chrom_names, int32_t n_chroms, const int32_t *chrom_id,
                              const int64_t *start, const int64_t *row_off, const double *stats, int32_t n_cols, char delim,
                              int32_t precision, bool want_lines, std::vector<fpt_text_part> &parts) {
    if (n_intervals < 0 || n_cols < 0 || precision < 0 || precision > 30 || n_chroms < 0)
        return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    parts.clear();
    if (n_intervals == 0) return FPT_OK;
    if (!chrom_names || !chrom_id || !start || !row_off) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    std::vector<size_t> lc((size_t)n_chroms);
    size_t lc_max = 0;
    for (int32_t c = 0; c < n_chroms; ++c) {
        if (!chrom_names[c]) return fpt_internal_fail(FPT_ERR_INVALID, "null chromosome name");
        lc[(size_t)c] = strlen(chrom_names[c]);
        lc_max = std::max(lc_max, lc[(size_t)c]);
    }
    if (row_off[0] < 0) return fpt_internal_fail(FPT_ERR_INVALID, "bad row offsets");
    for (int64_t j = 0; j < n_intervals; ++j) {
        if (chrom_id[j] < 0 || chrom_id[j] >= n_chroms) return fpt_internal_fail(FPT_ERR_INVALID, "chromosome id out of range");
        if (row_off[j + 1] < row_off[j]) return fpt_internal_fail(FPT_ERR_INVALID, "bad row offsets");
    }
    const int64_t rows0 = row_off[0], total = row_off[n_intervals] - rows0;
    if (total > 0 && n_cols > 0 && !stats) return fpt_internal_fail(FPT_ERR_INVALID, "null matrix");
    want_lines = want_lines && fpt_internal_line_bound(lc_max, n_cols, precision) < 65536;
    int nt = std::min(fpt_host_cpus(), 64);
    if (const char *e = getenv("FPT_TEXT_THREADS")) nt = atoi(e) > 0 ? atoi(e) : 1;
    if ((int64_t)nt > total / 4096 + 1) nt = (int)(total / 4096 + 1);
    // interval ranges of about equal row counts
    std::vector<int64_t> cut((size_t)nt + 1, n_intervals);
    cut[0] = 0;
    {
        int t = 1;
        for (int64_t j = 0; j < n_intervals && t < nt; ++j)
            while (t < nt && row_off[j] - rows0 >= total * t / nt) cut[(size_t)t++] = j;
    }
    parts.resize((size_t)nt);
    std::vector<int> fail((size_t)nt, 0);
    auto work = [&](int t) {
        fpt_text_part &out = parts[(size_t)t];
        const int64_t j0 = out.j0 = cut[(size_t)t], j1 = out.j1 = cut[(size_t)t + 1];
        if (j0 >= j1) return;
        const int64_t rows = row_off[j1] - row_off[j0];
        size_t per_line = lc_max + 44 + (size_t)n_cols * (size_t)(22 + precision);
        for (int attempt = 0; attempt < 2; ++attempt) {
            const size_t room = (size_t)rows * per_line + 64;
            out.data.reset(new char[room]);
            if (want_lines) out.line_len.resize((size_t)rows);
            char *b0 = out.data.get(), *p = b0, *end = b0 + room;
            uint16_t *ll = want_lines ? out.line_len.data() : nullptr;
            int64_t bad_row = -1;
            for (int64_t j = j0; j < j1 && p; ++j) {
                const int32_t c = chrom_id[j];
                const int64_t n = row_off[j + 1] - row_off[j];
                p = format_rows(chrom_names[c], lc[(size_t)c], start[j], stats + row_off[j] * (int64_t)n_cols, n, n_cols, nullptr,
                                0, n, delim, precision, p, end, &bad_row, ll);
                if (ll) ll += n;
            }
            if (p) {
                out.size = (size_t)(p - b0);
                return;
            }
            per_line = fpt_internal_line_bound(lc_max, n_cols, precision);  // values beyond 1e17: the snprintf path
        }
        fail[(size_t)t] = 1;
    };
    if (nt <= 1) {
        work(0);
    } else {
        std::vector<std::thread> team;
        for (int t = 0; t < nt; ++t) team.emplace_back(work, t);
        for (std::thread &t : team) t.join();
    }
    for (int t = 0; t < nt; ++t)
        if (fail[(size_t)t]) return fpt_internal_fail(FPT_ERR_INVALID, "formatting failed");
    return FPT_OK;
}

extern "C" {
#pragma GCC visibility push(default)

int fpt_format_stats_batch(int64_t n_intervals, const char *const *chrom_names, int32_t n_chroms, const int32_t *chrom_id,
                           const int64_t *start, const int64_t *row_off, const double *stats, int32_t n_cols, char delim,
                           int32_t precision, char *buf, int64_t cap, int64_t *len_out) {
    if (!len_out || cap < 0 || (!buf && cap > 0)) return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    std::vector<fpt_text_part> parts;
    if (int rc = fpt_internal_format_batch(n_intervals, chrom_names, n_chroms, chrom_id, start, row_off, stats, n_cols, delim,
                                           precision, false, parts))
        return rc;
    size_t total = 0;
    std::vector<size_t> at;
    for (const fpt_text_part &s : parts) at.push_back(total), total += s.size;
    *len_out = (int64_t)total;  // also when it does not fit: the size to come back with
    if ((int64_t)total > cap) return fpt_internal_fail(FPT_ERR_INVALID, "output buffer too small");
    if (parts.size() <= 1) {
        if (!parts.empty()) memcpy(buf, parts[0].data.get(), parts[0].size);
    } else {  // (the copies on threads too: a fresh buffer of the caller is faulted in page by page)
        std::vector<std::thread> team;
        for (size_t t = 0; t < parts.size(); ++t)
            team.emplace_back([&, t]() { memcpy(buf + at[t], parts[t].data.get(), parts[t].size); });
        for (std::thread &t : team) t.join();
    }
    return FPT_OK;
}

#pragma GCC visibility pop
}
