// fpt_kernels.hpp -- host-visible launch interface of fpt_kernels.hip
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fpt.h"

namespace fptk {

struct scan_launch {
    int64_t n_intervals;
    int32_t interval_len;
    const int64_t *interval_off;
    const int32_t *tile_iv;
    const int32_t *tile_t0;
    const int32_t *tile_tl;
    const void *tile_recs;  // the same table as lean_tile_rec records (the lean kernel reads those)
    int64_t tile_first;
    int32_t tiles_per_interval;
    int32_t tile_len;
    int32_t hw, shw, k_trim;
    int32_t n_scales;
    int32_t scales[FPT_MAX_SCALES];
    int32_t nc_max;
    int64_t total_bases;
    const double *counts_plus, *counts_minus;
    const uint8_t *seq;
    const double *table;
    const double *model;
    double *exp_out, *obs_out, *pval_out, *winp_out;
    int32_t *status_out;
    const void *memo;  // double2[memo_exp * memo_obs] or nullptr
    int32_t memo_exp, memo_obs;
    int32_t counts_only;
    int32_t ablate;
    int32_t *redo;         // per-tile redo flags (memo mode), or nullptr
    int32_t *redo_list;    // second pass: room for the flagged tiles of all launches (one int32 per tile) ...
    int32_t *redo_cursor;  // ... and two device ints (count, next): zero when the launch starts ...
    int32_t redo_cursor_clear;  // ... or zeroed by launch_scan when this is set
    const int32_t *dm_ids; // per-interval model slot relative to `model`, or nullptr
    int32_t table_global;  // bias table read through the L1/L2 caches (default) instead of an LDS copy
    const void *table2;    // bias table in the lean kernel's order (build_lean_table), or nullptr
    int32_t n_cu;          // compute units of the device
    // second-level (exp, obs) table of the redo pass (launch_nb_memo2), or nullptr
    const void *memo2;
    int32_t *memo2_max;    // device: [0] largest exp, [1] largest obs the first pass missed (-1: none)
    const int32_t *memo2_have;  // device: the table is filled for exp <= [0] and obs <= [1] already (kept across calls)
    int32_t memo2_rows, memo2_stride;
};

// fpt_scan_lean.hip: the first pass of memo mode for the `detect` defaults (hw 5, shw 50, clip 0.01)
struct lean_tile_rec {  // a tile of a ragged batch, read by the kernel with one scalar load
    int64_t out_off;    // offset of the tile's interval in the output tracks (= interval_off[iv])
    int32_t iv, t0, tl, len;
    int32_t pad_[2];
};
static_assert(sizeof(lean_tile_rec) == 32, "one s_load_dwordx8");
constexpr int kLeanClasses = 7;
constexpr int kLeanNT[kLeanClasses] = {128, 192, 256, 384, 512, 768, 1024};  // its workgroup sizes
// The tiles of a ragged batch are binned into kLeanClasses classes by the bases (+ halo) they hold: k_scan_lean's
// workgroup sizes.  (Rounds 4-5 also had one-wavefront-per-interval kernels for the first three classes --
// fpt_scan_wave.hip, in the history up to 9f0b6e1 -- which landed level with these and were removed.)
struct lean_class_set {
    int lmax[kLeanClasses];     // class c holds tiles of (lmax[c-1], lmax[c]] bases incl. halo
    int nt[kLeanClasses];       // k_scan_lean's workgroup size for the class
    int first_split;            // the first class a PIECE of a split interval may go to
};
lean_class_set make_lean_classes();
bool scan_lean_applies(const scan_launch &sl);
bool scan_lean_applies_hw(int hw, int shw, int k_trim);
size_t scan_lean_lds_bytes(int nt);
hipError_t scan_lean_set_lds(int nt);
void launch_scan_lean(hipStream_t st, int nt, int grid, const scan_launch &sl);
void build_lean_table(const double *table4096, double *out8192);

struct fdr_launch {
    int64_t n_intervals;
    int32_t interval_len;
    const int64_t *interval_off;
    int64_t base_index0;
    int32_t hw, times;
    uint64_t seed;
    const double *model;
    const void *memo;
    const void *alias;  // nb_alias_bytes(), filled by launch_nb_alias after launch_nb_memo
    int32_t n_models;   // models in `memo` / `alias` (1 without dm_ids)
    bool light, light_dbuf;  // split launches: the light draw instance first (see k_fdr_null MODE 3)
    // split + light launches of intervals of more than 256 bases: their light draws as slices (k_fdr_slice)
    const int32_t *slice_iv, *slice_start;  // DEVICE, n_slices each: interval and first output position of a slice
    int64_t n_slices;
    int64_t ws_total;                       // positions ws_key / ws_idx have room for
    const int64_t *goff;                    // DEVICE, per interval: start of its L + 2 counts in ghist
    int32_t *ghist, *gnan;                  // DEVICE, zeroed by the caller: counts per interval, NaN windows per interval
    int32_t memo_exp, memo_obs;
    const double *exp, *winp;
    const double *obs;  // optional (see fpt_fdr_desc.obs)
    double *efdr;
    const double *null_uniform;
    double *null_out;
    const int32_t *dm_ids;
    int32_t ablate;
    int32_t n2_max;
    int32_t nt;              // lanes per workgroup: 64 / 128 / 192 / 256 (0: chosen from n2_max)
    int32_t max_len;         // longest interval of the launch if known (0: n2_max is what is known)
    const int32_t *iv_list;  // optional DEVICE list of the intervals to process (n_list of them)
    int64_t n_list;
    // the hand-over of the set-up launch to the draw launch (all three, or none: one launch does both):
    // total_bases doubles, total_bases uint16, 2 x n_intervals int32
    double *ws_key;
    uint16_t *ws_idx;
    int32_t *ws_misc;
    void *gws;               // non-null: global-memory buffers, gws_stride bytes per workgroup,
    int64_t gws_stride;      //           room for gws_blocks workgroups at a time
    int64_t gws_blocks;
};

hipError_t launch_fdr(hipStream_t st, const fdr_launch &fl);

// fpt_posterior.hip: the multi-dataset posterior caller (cli/post.py:98-124) as one kernel
struct posterior_launch {
    int64_t n_intervals;
    int32_t interval_len;
    const int64_t *interval_off;
    int64_t total_bases;
    int32_t max_len;                    // longest interval (sizes workgroups and the grid; any length is handled)
    int32_t n_datasets, hw;
    double cutoff, pseudocount;
    const double *obs, *exp, *fdr, *w;  // (n_datasets, total_bases)
    const double *models;               // n_datasets x 24
    const double *betas;                // n_datasets x 2
    double *post_out;                   // (total_bases, n_datasets)
    double *prior_out, *delta_out, *ll_on_out, *ll_off_out;
    int32_t *status_out;
    double *off_table, *lgam_table;     // workspace of posterior_table_bytes(): filled by the launch (or both nullptr)
    bool all_simple;                    // every dataset's model passes posterior_model_simple() (decided on the host's copy)
    bool max_len_unknown;               // max_len is a guess: plan for longer intervals anyway
    void *plan_ws;                      // ragged batches: posterior_plan_bytes() of workspace for the chunk list
};
size_t posterior_plan_bytes(int64_t total_bases, int hw);
bool posterior_model_simple(const double *par24);  // finite parameters, ascending breakpoints: the fits by their active segment
size_t posterior_table_bytes(int n_datasets);  // off_table: n_datasets x 256 x 256 doubles, then lgam_table: 4096
size_t posterior_lds_bytes(int n_datasets, int nt);
hipError_t launch_posterior(hipStream_t st, const posterior_launch &pl);

struct segment_launch {
    int64_t n_intervals;
    int32_t interval_len;
    const int64_t *interval_off;
    const double *track;
    double threshold;
    int32_t w, decreasing;
    int32_t *counts;          // pass 1 output, one per interval
    const int64_t *offsets;   // pass 2 input: exclusive prefix of counts
    int32_t *seg_iv, *seg_start, *seg_end;
    double *seg_score;
};
void launch_segment(hipStream_t st, const segment_launch &sl, bool fill);
size_t fdr_lds_bytes(int n2, bool dbuf = false, bool global_buffers = false);
size_t fdr_slice_lds_bytes(int n2, int lanes);
int fdr_slices_of(int L, bool uniform);            // slices of an interval of L bases (of a uniform / a ragged batch) ...
int fdr_slice_positions_of(int L, bool uniform);   // ... and the output positions of each
void launch_nb_alias(hipStream_t st, const void *memo, int n_models, int memo_exp, int memo_obs, void *tables);
// the tile table of a ragged batch (three int32 arrays of n_tiles in `flat`, 32-byte records in `recs`)
// from the device offsets: class-major, intervals in order.  block_base (device): for every
// kPlanBlock intervals and class, the table index of the first tile of the block's first interval.
constexpr int kPlanBlock = 256;
void launch_plan_tiles(hipStream_t st, const int64_t *off, int64_t n_intervals, int64_t n_tiles, int H, int split_len,
                       const lean_class_set &cls, const int32_t *block_base, int32_t *flat, void *recs);
size_t nb_alias_bytes(int n_models, int memo_exp, int memo_obs);
size_t nb_alias_z_offset(int n_models, int memo_exp, int memo_obs);

void launch_kmer_probs(hipStream_t st, const uint8_t *seq, int64_t n_out, const double *table,
                       double *fwd, double *rev);
void launch_predict_rows(hipStream_t st, const double *obs, const double *probs, int64_t n_rows,
                         int l, int hw, int shw, int k_trim, double *exp_out, double *win_out);
void launch_nb_values(hipStream_t st, int what, const double *model, const double *ex,
                      const double *ob, int64_t n, double *out, int *flags);
void launch_nb_scalar(hipStream_t st, int what, const int32_t *k, const double *p, const double *r,
                      int64_t n, double *out);
void launch_special(hipStream_t st, int fn, const double *a, const double *b, const double *x,
                    int64_t n, double *out);
void launch_window_rows(hipStream_t st, int op, const double *x, const double *w, int64_t n_rows,
                        int n, int hw, double *out);
size_t scan_lds_bytes(int nc_max, bool tblg, bool memo_only);
hipError_t scan_set_lds(int nt, int hw, int shw, bool tblg, bool memo_only, bool second_pass, size_t lds);
void launch_scan(hipStream_t st, int nt, int grid, size_t lds, const scan_launch &sl, bool memo_only);
// clear / n_clear: int32 words set to 0; state: 8 ints set to (-1, -1, 0 x 6) -- the largest missed pair
// and three (count, cursor) pairs of the second pass; both optional
// have / have_state: the bounds of the kept second-level table (2 ints) -- 0 left as they are, 1 the
// missed pair recorded in `state` (which the last launch_nb_memo2 filled the table up to) joins
// them, 2 set to (-1, -1): nothing kept; rows / stride: that table's capacity
void launch_nb_memo(hipStream_t st, const double *models, int n_models, int memo_exp, int memo_obs, void *memo,
                    int32_t *clear = nullptr, int64_t n_clear = 0, int32_t *state = nullptr, int32_t *have = nullptr,
                    int have_state = 0, int rows = 0, int stride = 0);
// fills the entries up to max(have, miss_max) that lie outside `have` and outside the first-level table
void launch_nb_memo2(hipStream_t st, const double *models, int n_models, const int32_t *miss_max, const int32_t *have,
                     int memo_exp, int memo_obs, int rows, int stride, void *memo2);
void launch_detect_columns(hipStream_t st, int64_t n_intervals, int32_t interval_len, const int64_t *interval_off,
                           const int32_t *status, const double *ex, const double *ob, const double *pv,
                           const double *wp, const double *ef, int64_t total, double *out);
void launch_hist2d(hipStream_t st, const double *ex, const double *ob, int64_t n, int rows, int cols,
                   unsigned long long *hist);
void launch_synth(hipStream_t st, uint64_t seed, int64_t pos0_counts, int64_t n_counts,
                  double *counts_plus, double *counts_minus, int64_t pos0_seq, int64_t n_seq,
                  uint8_t *seq);
void launch_synth_hotspots(hipStream_t st, uint64_t seed, int64_t pos0, int64_t n, int padded_len, int per_mille,
                           double *counts_plus, double *counts_minus);
void launch_checksum(hipStream_t st, const double *x, int64_t n, unsigned long long *out);

}  // namespace fptk
