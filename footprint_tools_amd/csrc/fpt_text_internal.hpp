// fpt_text_internal.hpp -- the batch formatter of fpt_text.cpp as the track writer uses it
// (fpt_track_writer_write_stats): the text of a batch in parts, one per thread of the team, in order.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

struct fpt_text_part {
    std::unique_ptr<char[]> data;  // (not a std::string: growing one writes zeros first)
    size_t size = 0;
    int64_t j0 = 0, j1 = 0;          // the intervals whose lines these are
    std::vector<uint16_t> line_len;  // bytes of every line with its newline, when asked for
};

// worst-case bytes of one line (values beyond 1e17 included); line lengths are recorded only while
// this fits 16 bits
size_t fpt_internal_line_bound(size_t chrom_len, int32_t n_cols, int32_t precision);
int fpt_internal_format_batch(int64_t n_intervals, const char *const *chrom_names, int32_t n_chroms, const int32_t *chrom_id,
                              const int64_t *start, const int64_t *row_off, const double *stats, int32_t n_cols, char delim,
                              int32_t precision, bool want_lines, std::vector<fpt_text_part> &parts);
