/*
 * fpt.h -- C ABI of libfpt_hip.so: the MI355X (gfx950) implementation of the
 * footprint-tools per-nucleotide expected-cleavage / deviation-statistics scan.
 *
 * This is the drop-in boundary: plain pointers and sizes, no framework types.
 * Each entry point names the reference interface it replaces (paths under the
 * reference repo vierstralab/footprint-tools v1.3.7).  The reference-side
 * bindings are shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 (FPT_OK) or a negative fpt_status; the message
 *     of the last failure on the calling thread is fpt_last_error().
 *   - "host" entry points take caller-owned HOST buffers (numpy arrays) and do
 *     the H2D/D2H copies themselves: they mirror the one-interval-at-a-time
 *     reference calls.  "_dev" entry points take DEVICE pointers, enqueue on
 *     the context's stream and return without synchronising: they are the
 *     batched, HBM-resident path.
 *   - all floating-point data is float64, C-contiguous, like the reference's
 *     typed memoryviews; lengths that the reference types as C int stay int.
 *   - there is no CPU fallback: without a HIP device fpt_ctx_create fails.
 *   - a context owns its device tables, a workspace and a stream binding: calls on ONE context
 *     must not overlap (serialise them, as the Python layer does with a lock); different
 *     contexts -- one per GPU, or several on one GPU -- are independent.
 */
#ifndef FPT_H
#define FPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fpt_ctx fpt_ctx;

enum fpt_status {
    FPT_OK = 0,
    FPT_ERR_INVALID = -1,  /* bad argument */
    FPT_ERR_HIP = -2,      /* HIP runtime error (message has the hipError string) */
    FPT_ERR_NODEVICE = -3, /* no usable gfx950 device */
    FPT_ERR_ZERODIV = -4,  /* reference would raise ZeroDivisionError (dispersion.pyx:160-161) */
    FPT_ERR_NOMEM = -5
};

/* reducers of stats/windowing.pyx:60-178 */
enum fpt_window_op {
    FPT_WIN_SUM = 0,      /* windowing.sum               -> windowing.h:11-22  */
    FPT_WIN_PRODUCT = 1,  /* windowing.product           -> windowing.h:24-35  */
    FPT_WIN_FISHER = 2,   /* windowing.fishers_combined  -> windowing.h:38-51  */
    FPT_WIN_STOUFFER = 3, /* windowing.stouffers_z       -> windowing.h:53-67  */
    FPT_WIN_WSTOUFFER = 4 /* windowing.weighted_stouffers_z -> windowing.h:86-102 */
};

/* per-base NB quantities of modeling/dispersion.pyx */
enum fpt_nb_what {
    FPT_NB_CDF = 0,    /* dispersion_model.p_values        dispersion.pyx:291-316 */
    FPT_NB_LOGPMF = 1, /* dispersion_model.log_pmf_values  dispersion.pyx:170-196 (+ _0 :228-258) */
    FPT_NB_PMF = 2     /* dispersion_model.pmf_values      dispersion.pyx:199-225 (+ _0 :260-289) */
};

/* hcephes entry points reachable from the path (unit-test / diagnostics hook) */
enum fpt_special_fn {
    FPT_FN_GAMMA = 0, FPT_FN_LGAM = 1, FPT_FN_NDTR = 2, FPT_FN_NDTRI = 3, FPT_FN_LOG1P = 4,
    FPT_FN_ERF = 5, FPT_FN_ERFC = 6, FPT_FN_INCBET = 7, /* (a,b,x) */
    FPT_FN_CHDTRC = 8, /* (df = a, x) */
    FPT_FN_NDTR_WINDOW = 9, /* the normal cdf as the fused scan's Stouffer windows evaluate it */
    FPT_FN_NDTR_WINDOW_TAB = 10, /* ... and its table form (256 cubics for g; measured, not in use: DESIGN.md 4) */
    /* not hcephes: the short logarithm of the posterior kernel's likelihoods (positive normal x) and its
     * log1p on [0, 1] (fpt_math.hpp log_pos_fast / log1p_unit_fast; where the reference calls libm's log) */
    FPT_FN_LOG_FAST = 11, FPT_FN_LOG1P_FAST = 12
};

#define FPT_MAX_DISPERSION_MODELS 64
#define FPT_MAX_SCALES 8
#define FPT_KMER_TABLE 4096

const char *fpt_last_error(void);
int fpt_version(void);
int fpt_device_count(int *n_out);

/* ---- context: one per (process, GPU); owns a stream, the bias table, the
 * dispersion-model table and a grow-only device workspace. */
int fpt_ctx_create(int device_id, fpt_ctx **out);
int fpt_ctx_destroy(fpt_ctx *ctx);
/* run on a caller-provided hipStream_t (e.g. the framework's current stream).  The handle is
 * taken as it is: NULL is the device's default (null) stream, which is what a framework running
 * on its default stream hands over, and work is then ordered with that stream.  To go back to
 * the context's own non-blocking stream call fpt_ctx_use_own_stream. */
int fpt_ctx_set_stream(fpt_ctx *ctx, void *hip_stream);
int fpt_ctx_use_own_stream(fpt_ctx *ctx);
int fpt_ctx_synchronize(fpt_ctx *ctx);

/* bias_model.__getitem__ / kmer_model.read_model (modeling/bias.py:16-17, 63-86):
 * 4096 propensities indexed by sum code(s_m)*4^(5-m), A=0 C=1 G=2 T=3; `dflt` is the
 * value for k-mers that are missing or contain a non-ACGT letter (1e-6 in the reference). */
int fpt_set_bias_table(fpt_ctx *ctx, const double *table4096, double dflt);

/* dispersion_model.mu_params / .r_params (modeling/dispersion.pyx:113-124): 9 and 15 doubles,
 * (breaks, intercepts, slopes).  dm_id selects one of FPT_MAX_DISPERSION_MODELS slots
 * (the reference has one model per dataset: cli/detect.py:334-336, stats/posterior.py:119). */
int fpt_set_dispersion(fpt_ctx *ctx, int dm_id, const double *mu_params9, const double *r_params15);

/* ---- host-buffer entry points (one reference call each) ----------------- */

/* kmer_model.probs on both strands as prediction.compute uses it
 * (modeling/bias.py:88-111; modeling/predict.pyx:47-61, 150-153):
 * fwd[j] = T[seq[j..j+5]], rev[j] = T[revcomp(seq[j+1..j+6])], j in [0, seq_len-6).
 * Letters are matched case-insensitively (predict.pyx:140 upper-cases first). */
int fpt_kmer_probs(fpt_ctx *ctx, const uint8_t *seq, int64_t seq_len, double *fwd, double *rev);

/* fast_predict (modeling/predict.h:23-74, smoothing.h:107-133) for n_rows independent rows of
 * length l laid out back to back: the `cdef predict()` of modeling/predict.pyx:23-45 is the
 * n_rows == 1 case.  exp_out / win_out: n_rows*l doubles. */
int fpt_predict(fpt_ctx *ctx, const double *obs, const double *probs, int64_t n_rows, int l,
                int half_win_width, int smoothing_half_win_width, double smoothing_clip,
                double *exp_out, double *win_out);

/* dispersion_model.{p_values, log_pmf_values, pmf_values}(exp, obs) with the model in slot dm_id.
 * FPT_ERR_ZERODIV when fit_r's piecewise value is exactly 0 for some element. */
int fpt_nb_values(fpt_ctx *ctx, int what, int dm_id, const double *exp, const double *obs,
                  int64_t n, double *out);

/* nbinom.{cdf,logpmf,pmf}(k, p, r) element-wise (stats/distributions/nbinom.pyx:82-138) */
int fpt_nb_scalar(fpt_ctx *ctx, int what, const int32_t *k, const double *p, const double *r,
                  int64_t n, double *out);

/* windowing.<op>(x, hw) (stats/windowing.pyx:34-58,132-158; windowing.h:69-122) applied to
 * n_rows independent rows of length n: out[i] = f(x[i-hw..i+hw]) for i in [hw, n-hw), 1.0
 * elsewhere.  w (weights) only for FPT_WIN_WSTOUFFER, else NULL. */
int fpt_window(fpt_ctx *ctx, int op, const double *x, const double *w, int64_t n_rows, int n,
               int hw, double *out);

/* element-wise hcephes functions (see fpt_special_fn); b/x may be NULL for 1-argument functions */
int fpt_special(fpt_ctx *ctx, int fn, const double *a, const double *b, const double *x,
                int64_t n, double *out);

/* ---- batched, HBM-resident scan (cli/detect.py:120-130 for many intervals) */

typedef struct fpt_scan_desc {
    int64_t n_intervals;
    /* uniform batches: every interval has `interval_len` bases and interval_off == NULL.
     * ragged batches: interval_off (DEVICE, n_intervals+1 int64) are offsets into the output
     * tracks, interval_off_host the same array on the HOST (the launches are sized from the interval
     * lengths; NULL: the offsets come back from the device first, behind a wait for the stream).  The
     * tile table itself is made on the device in every call: nothing is kept between calls. */
    int32_t interval_len;
    const int64_t *interval_off;
    const int64_t *interval_off_host;
    int32_t half_win_width;           /* prediction(half_win_width=5)             predict.pyx:85 */
    int32_t smoothing_half_win_width; /* prediction(smoothing_half_win_width=..)  predict.pyx:85 */
    double smoothing_clip;            /* prediction(smoothing_clip=0.01)          predict.pyx:85 */
    int32_t n_scales;                 /* number of Stouffer windows (detect uses one, hw=3) */
    int32_t scales[FPT_MAX_SCALES];   /* half window widths                       detect.py:84   */
    int32_t dm_id;                    /* dispersion model slot (first slot when dm_ids is given) */
    /* optional per-interval dispersion models: dm_ids (DEVICE int32[n_intervals]) holds, for each
     * interval, a slot index in [0, n_dm) relative to dm_id; slots dm_id .. dm_id+n_dm-1 must be
     * set.  The reference has one model per dataset (cli/detect.py:334-336; a list of models in
     * stats/posterior.py:119); a batch that mixes intervals of several datasets uses this. */
    const int32_t *dm_ids;
    int32_t n_dm;
    int32_t nb_mode;                  /* how p = nbinom.cdf(obs; exp) is evaluated per base:
                                       * FPT_NB_AUTO, FPT_NB_DIRECT, FPT_NB_MEMO or FPT_NB_NONE (below) */
    /* inputs (DEVICE).  With pad = hw + shw, interval i of length L_i owns
     *   counts_*[ off_i + i*(2*pad+1) .. +L_i+2*pad+1 )   padded cut counts, genomic order
     *   seq     [ off_i + i*(2*pad+7) .. +L_i+2*pad+7 )   ASCII bases, 3 extra on each side
     * exactly the arrays prediction.compute fetches (predict.pyx:132-140). */
    const double *counts_plus;
    const double *counts_minus;
    const uint8_t *seq;
    /* outputs (DEVICE), tracks of sum(L_i) doubles; any may be NULL to skip the store.
     * winp holds n_scales tracks back to back. */
    double *exp_out;
    double *obs_out;
    double *pval_out;
    double *winp_out;
    /* optional per-interval status (DEVICE int32[n_intervals]): bit 0 = ZeroDivisionError */
    int32_t *status_out;
} fpt_scan_desc;

/* p-value evaluation modes of the fused scan.  Both give the same numbers: expected counts are
 * integers (a sum of two round()s) and the reference truncates obs to a C int
 * (dispersion.pyx:314), so p and z = ndtri(1-p) are functions of the integer pair (exp, obs).
 *   DIRECT  every base evaluates hcephes incbet / ndtri itself.
 *   MEMO    each fpt_scan_dev call first fills a (memo_exp x memo_obs) table of (p, z) with the
 *           same device incbet/ndtri (one small launch on the same stream, so it is part of the
 *           timed work), the scan kernel looks pairs up and falls back to DIRECT evaluation for
 *           pairs outside the table or non-integer / non-finite exp.
 *   AUTO    MEMO when the batch has at least 8x more bases than the table has entries.
 *   NONE    no p-values at all: only exp_out / obs_out are written (what `ftd learn_dm` needs
 *           before a dispersion model exists, cli/learn_dm.py:102-107); n_scales must be 0 and
 *           no dispersion slot has to be set. */
enum fpt_nb_mode { FPT_NB_AUTO = 0, FPT_NB_DIRECT = 1, FPT_NB_MEMO = 2, FPT_NB_NONE = 3 };

/* table extent for FPT_NB_MEMO (defaults 256 x 256; each in [1, 4096]).  fpt_fdr_dev's sampling
 * table has the same rows and by default 2048 obs columns; this call sets its columns too. */
int fpt_set_memo_dims(fpt_ctx *ctx, int memo_exp, int memo_obs);

/* FPT_NB_MEMO keeps one thing across calls: the second-level (exp, obs) table for pairs beyond the
 * first-level one (hotspots: counts in the hundreds).  It is a function of the dispersion models
 * alone, up to 4096 x 4096 entries per model, and grows to the largest pair any call has met: the
 * call that first meets a range computes it between its two passes (and sends the tiles that
 * needed it through the general kernel), later calls look it up in the first pass.  It is emptied
 * when a model of the batch changes (fpt_set_dispersion with other values), when the batch uses
 * other model slots, when the context is given another stream, and by this call (measurements of the cold path; FPT_MEMO2_KEEP=0 in the
 * environment of fpt_ctx_create empties it at every call).  No reference counterpart: the reference
 * evaluates scipy's nbinom.cdf per base (modeling/dispersion.pyx:311-314). */
int fpt_drop_kept_tables(fpt_ctx *ctx);

/* Enqueue the fused scan on the context's stream (no synchronisation). */
int fpt_scan_dev(fpt_ctx *ctx, const fpt_scan_desc *desc);

/* The same scan on HOST arrays: what the reference's per-call API is (`prediction.compute` /
 * `dm.p_values` / `windowing.stouffers_z` take numpy arrays and return numpy arrays:
 * modeling/predict.pyx:116-163, modeling/dispersion.pyx:291-316, stats/windowing.pyx:114-130).  EVERY
 * pointer of `desc` is a host pointer here (interval_off or interval_off_host: the host offsets of a ragged
 * batch, either field; status_out optional).  The batch is cut into chunks of about `chunk_bases` output bases
 * (0: 2^21) that travel through a three-stage pipeline, three chunks in flight: the calling thread copies a
 * chunk's inputs to the device and launches fpt_scan_dev on it, a second thread (started per call) copies every
 * chunk's results back in order -- each stage on its own stream, so that both directions of the PCIe link and the
 * kernel overlap.  The caller's arrays are used where they lie, page-locked (fpt_host_alloc, hipHostRegister) or
 * pageable: nothing is staged by the library.  Returns when every output is in place.  Results are those of
 * fpt_scan_dev on the whole batch, bit for bit (intervals are independent). */
int fpt_scan_host(fpt_ctx *ctx, const fpt_scan_desc *desc, int64_t chunk_bases);

/* what the last fpt_scan_host of the context did */
typedef struct fpt_scan_host_stats {
    double seconds;           /* wall time of the call */
    int64_t bases;            /* output bases */
    int64_t chunks;
    int64_t bytes_h2d, bytes_d2h;
    int32_t inputs_pinned, outputs_pinned; /* 1: the caller's arrays are page-locked (copies return at once; with
                                            * pageable arrays a copy blocks the thread that issued it) */
    /* where the calling thread spent the call: waiting for a slot or for the last results, (unused since the
     * library stages nothing: 0), and issuing copies and launches -- with pageable inputs the copies themselves */
    double wait_seconds, stage_seconds, issue_seconds;
} fpt_scan_host_stats;
int fpt_scan_host_last(fpt_ctx *ctx, fpt_scan_host_stats *out);

/* Touches every page of a FRESH (never written) host array with a team of threads, so that the first copy into it
 * does not pay a page fault per 4 KiB (a new 1.6 GB numpy array takes a device-to-host copy at 20 GB/s instead of 56).
 * Writes one zero byte per page: only for arrays that hold nothing yet.  No context needed. */
int fpt_host_prefault(void *host, int64_t bytes);

/* page-locked host memory for the arrays of fpt_scan_host (hipHostMalloc / hipHostFree) */
int fpt_host_alloc(fpt_ctx *ctx, int64_t bytes, void **host_out);
int fpt_host_free(fpt_ctx *ctx, void *host);

/* ---- empirical FDR of the window p-values (cli/detect.py:132-135 for many intervals)
 *
 * Reference, per interval: `_, pvals_null = dm.sample(exp, times)` draws `times` NB counts per
 * base from the model at that base's expected count and takes their lower-tail p-values
 * (dispersion.pyx:318-355); every column of draws goes through the same Stouffer window
 * (detect.py:133); `emperical_fdr` ranks each observed window p-value in the interval's
 * pooled null values: efdr = min(1, #{null <= p} / (L*times)), NaN -> 1 (fdr/__init__.py:12-33,
 * utils.pyx:52-79).
 *
 * Here all of that is one kernel per interval.  The draw and its p-value are taken together from a
 * 32-bit word w of Philox4x32-10 keyed by `seed` with counter (global base index, sample index), so
 * results are reproducible and independent of how intervals are sharded; they are statistically, not
 * bitwise, the reference's (numpy MT19937).
 *
 * The null draws, exactly (tests restate this; doubles operation for operation):
 *   - at an expected value that is an integer below the table's height (256) the draw is an ALIAS-table
 *     lookup.  The row's outcomes are k = 0 .. n-2 with probability cdf(k) - cdf(k-1) and "n-1 or more"
 *     with 1 - cdf(n-2); n = 2^lg is the smallest power of two with 1 - cdf(n-2) <= 2^-32, at most the
 *     largest power of two <= the table's width (2048), lg >= 1.  With q(k) = n p(k), negative p taken
 *     as 0: outcomes with q < 1 are queued as "small", the others as "large", both in index order; while
 *     both queues hold something, the first small s gets alias = first large l, q(l) = (q(l) + q(s)) - 1,
 *     and l moves to the end of the small queue when that is < 1; whatever is left gets q = 1.
 *     entry(k) = threshold << lg | alias, threshold = floor(q(k) 2^(32-lg) + 1/2) capped at 2^(32-lg) - 1.
 *     Draw: slot = w >> (32 - lg), t = w mod 2^(32-lg), outcome = t < threshold(slot) ? slot : alias(slot);
 *     its p-value is cdf(outcome).  (A row with a NaN: outcome = slot.)
 *   - the outcome "n-1 or more" is the smallest k >= n-1 with cdf(k) >= u', u' = cdf(n-2) + (1 - cdf(n-2)) f,
 *     f = (position of t within its part of the slot -- [0, threshold) or [threshold, 2^(32-lg)) -- + 1/2)
 *     / (size of that part): galloping + bisection on the direct incbet.
 *   - at any other expected value: the smallest k with cdf(k) >= u, u = (w + 1/2) 2^-32, by the same search.
 * A table's probabilities are the row's to about 2^-31 per outcome (thresholds are rounded to 2^-32). */
typedef struct fpt_fdr_desc {
    int64_t n_intervals;
    int32_t interval_len;             /* uniform batches (interval_off == NULL) */
    const int64_t *interval_off;      /* ragged: DEVICE offsets into the tracks (n_intervals+1) */
    int64_t base_index0;              /* global index of this batch's first base (RNG counter) */
    int32_t half_win_width;           /* Stouffer window of the null tracks (detect: 3) */
    int32_t times;                    /* fdr_shuffle_n (detect default 100) */
    uint64_t seed;
    int32_t dm_id;                    /* model slot (first slot when dm_ids is given) */
    const int32_t *dm_ids;            /* optional DEVICE int32[n_intervals], values in [0, n_dm) */
    int32_t n_dm;
    const double *exp;                /* DEVICE: expected counts track */
    const double *winp;               /* DEVICE: observed window p-values, same window */
    double *efdr_out;                 /* DEVICE: empirical FDR track */
    const double *null_uniform;       /* optional DEVICE [sum(L) * times] uniforms replacing Philox
                                       * (row-major base x sample; w = floor(u 2^32)): deterministic tests */
    double *null_winp_out;            /* optional DEVICE [sum(L) * times]: the null window p-values
                                       * (detect.py:133 win_pvals_null, row-major base x sample) */
    const double *obs;                /* optional DEVICE: the observed counts track the p-values were made
                                       * from.  With it the observed window p-values are re-made inside the
                                       * call by the operations (and the normal cdf) the null windows go
                                       * through, so that a null window of the same counts TIES with the
                                       * observed one exactly, as in the reference (both go through one
                                       * stouffers_z); `winp` then only says which positions are NaN or 1.
                                       * Without it observed values are ranked as given, and such ties -- a
                                       * large share of the null for sparse counts -- fall either way by the
                                       * rounding of whatever made `winp`. */
    const int64_t *interval_off_host; /* optional HOST copy of interval_off (like fpt_scan_desc's): the
                                       * launches are sized from the interval lengths, and without it the
                                       * offsets come back from the device first -- a copy and a wait for
                                       * everything queued on the stream */
} fpt_fdr_desc;

/* Enqueue the null sampling + ranking on the context's stream (no synchronisation).
 * Intervals of up to 2048 bases are processed out of LDS (with the `detect` width in three kinds of launches:
 * a per-interval set-up, the draws -- an interval of more than 256 bases by several workgroups, each a slice
 * of it --, and the draws of the few intervals that need the direct inverse cdf); longer ones (up to 2^22
 * bases) by one kernel over buffers in global memory, which is slower per base.  interval_off_host, when
 * given, must be the very array that was uploaded to interval_off: launches, buffers and slices are sized
 * from it.  Where it disagrees with the device's offsets nothing is written out of bounds, and an interval it
 * misdescribes -- longer than its launch's buffers, beyond the total the host array ends at, or not covered
 * by its slices -- is not processed: its efdr is set to NaN. */
int fpt_fdr_dev(fpt_ctx *ctx, const fpt_fdr_desc *desc);

/* ---- the multi-dataset posterior caller (BASELINE config 5; SURVEY.md 8a row A11 + 8f row 4).
 * One call = cli/post.py:98-124 (`posterior_stats.__getitem__`) for a whole batch of intervals
 * and all datasets:
 *     prior  = posterior.compute_prior_weighted(fdr, w, cutoff)           stats/posterior.py:12-42
 *     delta  = posterior.compute_delta_prior(obs, exp, fdr, betas, cutoff) stats/posterior.py:45-90
 *     ll_on  = posterior.log_likelihood(obs, exp, dms, delta=delta, w=hw)  stats/posterior.py:93-121
 *     ll_off = posterior.log_likelihood(obs, exp, dms, w=hw)               (dm.log_pmf_values ->
 *              nbinom.logpmf, dispersion.pyx:170-226, nbinom.pyx:82-100; windowing.sum, edges 1.0)
 *     post   = -posterior.posterior(prior, ll_on, ll_off); post[post <= 0] = 0
 *                                                  stats/posterior.py:124-149, cli/post.py:121-122
 * Tracks are DEVICE arrays of shape (n_datasets, sum(L)), row-major: row d is dataset d's track
 * over the batch's intervals back to back -- the reference's (n, m) arrays of `_load_data`
 * (cli/post.py:57-87: exp = file column 3, obs = 4, fdr = 7, w = 1 where the dataset has a row;
 * defaults 0 / 0 / 1 / 0) concatenated over intervals.  Dataset d uses dispersion slot dm_id + d
 * and the Beta prior betas[d] (sample file columns beta_a, beta_b).  post_out is
 * (sum(L), n_datasets) row-major: the slice of an interval is the reference's `stats` (= post.T).
 * status_out[i] = 1 where a dispersion fit of interval i divides by zero (dm.log_pmf_values raises
 * ZeroDivisionError there).  Enqueued on the context's stream; no synchronisation. */
typedef struct fpt_posterior_desc {
    int64_t n_intervals;
    int32_t interval_len;             /* uniform batches (interval_off == NULL) */
    const int64_t *interval_off;      /* ragged: DEVICE offsets into the tracks (n_intervals+1) */
    int64_t total_bases;              /* sum(L): the row length of the tracks */
    int32_t max_interval_len;         /* ragged: the longest interval (a sizing hint; 0 = unknown) */
    int32_t n_datasets;               /* >= 1; more than FPT_MAX_DISPERSION_MODELS need `models` */
    int32_t dm_id;                    /* model slot of dataset 0; dataset d uses dm_id + d (ignored with `models`) */
    int32_t half_win_width;           /* likelihood window (cli/post.py: 3), <= 32 */
    double fdr_cutoff;                /* cli/post.py --fdr_cutoff (0.05) */
    double pseudocount;               /* compute_prior_weighted's default 0.5 */
    const double *betas;              /* HOST: n_datasets x 2 (beta_a, beta_b) */
    const double *obs, *exp, *fdr, *w;/* DEVICE tracks */
    double *post_out;                 /* DEVICE (sum(L), n_datasets) */
    double *prior_out;                /* optional DEVICE (n_datasets, sum(L)) */
    double *delta_out;                /* optional DEVICE (sum(L)) */
    double *ll_on_out, *ll_off_out;   /* optional DEVICE (n_datasets, sum(L)) */
    int32_t *status_out;              /* optional DEVICE int32[n_intervals], zeroed by the caller */
    const double *models;             /* optional HOST n_datasets x 24 (mu_params 9, r_params 15 per dataset): the
                                       * datasets' dispersion models handed over with the call instead of through
                                       * slots dm_id .. -- ANY number of datasets (cli/post.py:98-124 loops over
                                       * however many samples the sample file lists; the slots hold 64), up to
                                       * 2^20.  The call's tables of the unoccupied log-pmf (512 KiB per dataset,
                                       * rebuilt per call) are used for up to 4,096 datasets and batches of 16,384
                                       * bases or more; beyond that, or when they cannot be allocated, every value
                                       * is evaluated in the kernel -- the same records */
} fpt_posterior_desc;
int fpt_posterior_dev(fpt_ctx *ctx, const fpt_posterior_desc *desc);

/* The record columns of `ftd detect` for a whole batch, on the device (cli/detect.py:136-146):
 *     stats = np.column_stack((exp, obs, -np.log(pvals), -np.log(win_pvals), efdr))
 * into out_dev, a DEVICE (total_bases, 5) row-major matrix whose slice per interval is that
 * interval's `stats`; where status[i] != 0 (the scan's per-interval status: the reference's
 * `except Exception` branch, detect.py:136-140) interval i gets pvals = win_pvals = efdr = 1.
 * All tracks are DEVICE arrays of total_bases doubles; status may be NULL.  Enqueued on the
 * context's stream; no synchronisation. */
int fpt_detect_columns_dev(fpt_ctx *ctx, int64_t n_intervals, int32_t interval_len, const int64_t *interval_off_dev,
                           int64_t total_bases, const int32_t *status_dev, const double *exp_dev, const double *obs_dev,
                           const double *pval_dev, const double *winp_dev, const double *efdr_dev, double *out_dev);

/* (exp, obs) histogram of `ftd learn_dm` (cli/learn_dm.py:276-287): hist[int(exp), int(obs)] += 1
 * for the n pairs of two DEVICE tracks, pairs outside the rows x cols histogram (the reference
 * uses 200 x 1000) ignored like its IndexError branch; negative or non-finite values are
 * skipped.  hist_dev: DEVICE uint64[rows*cols], accumulated into (zero it first). */
int fpt_hist2d_dev(fpt_ctx *ctx, const double *exp_dev, const double *obs_dev, int64_t n, int rows,
                   int cols, uint64_t *hist_dev);

/* Footprint calling on a whole track: `utils.segment(x, threshold, w, decreasing)`
 * (stats/utils.pyx:15-50) of every interval, as `write_segments_to_output` applies it to the FDR
 * column with w = 3, decreasing = 1 (cli/utils.py:204).  A run opens at the first element with
 * dir*x >= dir*threshold and closes at the first with dir*x < dir*threshold (NaN does neither; a
 * run still open at the end of its interval is dropped, and a passing element at a position
 * below w - 1 does not open one, both as in the reference, whose "no open run" state is
 * curr_start < 0); it is reported as [first - w + 1, closing - 1 + w) and merged with the
 * previous one when it starts at or before that one's end.  score = min(x[start:end]) clipped to the interval (NaN if any NaN),
 * the reference's default score_fn.  Two calls: count, then fill buffers of at least that size. */
typedef struct fpt_segment_desc {
    int64_t n_intervals;
    int32_t interval_len;             /* uniform batches (interval_off == NULL) */
    const int64_t *interval_off;      /* ragged: DEVICE offsets into the track (n_intervals+1) */
    const double *track;              /* DEVICE */
    double threshold;
    int32_t w;
    int32_t decreasing;
} fpt_segment_desc;

/* pass 1: number of segments of the whole batch -> *total_out (synchronises the stream) */
int fpt_segment_count_dev(fpt_ctx *ctx, const fpt_segment_desc *desc, int64_t *total_out);
/* pass 2 (same desc as the preceding count call): segments ordered by interval, then position.
 * DEVICE outputs of `capacity` >= total entries: interval index, start, end (relative to the
 * interval, end exclusive and possibly past the interval's end like the reference's), score. */
int fpt_segment_fill_dev(fpt_ctx *ctx, const fpt_segment_desc *desc, int64_t capacity,
                         int32_t *seg_interval, int32_t *seg_start, int32_t *seg_end, double *seg_score);

/* Fill device buffers with the synthetic workload of BASELINE.json configs 1-3:
 * counter-hash generator, element at global position p of stream s is
 * mix(mix(seed+s)+p); counts = U{0..19} as float64, bases uniform ACGT.
 * Any pointer may be NULL. */
int fpt_synth_dev(fpt_ctx *ctx, uint64_t seed, int64_t pos0_counts, int64_t n_counts,
                  double *counts_plus, double *counts_minus, int64_t pos0_seq, int64_t n_seq,
                  uint8_t *seq);

/* ---- cut-count ingestion: the step right before the path.
 * Reference: cutcounts.bamfile (footprint_tools/cutcounts.py): `validate_read` :119-145, the
 * pairing rules of `read_pair_generator` :196-205, `_add_read` :231-248 (forward reads cut at
 * reference_start + offset[0] on '+', reverse reads at reference_end + offset[1] on '-'; default
 * offset (0, -1)) and `lookup` :274-313.  A read is counted once whether or not its mate is in the
 * fetched window, so counts[x] = number of valid reads whose cut position is x, for any interval.
 *
 * fpt_bam_*: a sequential BAM reader (BGZF through zlib; htslib is not in this image -- parity of
 * the reader is UNPINNED, it is tested on files the tests write).  fpt_bam_read hands out up to
 * max_reads alignments per call (reference id, 0-based start, end = start + reference-consuming
 * CIGAR operations as pysam's reference_end, SAM flag, MAPQ); *n_out = 0 at the end of the file. */
typedef struct fpt_bam fpt_bam;
int fpt_bam_open(const char *path, fpt_bam **out);
int fpt_bam_close(fpt_bam *bam);
int fpt_bam_n_refs(fpt_bam *bam, int32_t *n_out);
int fpt_bam_ref(fpt_bam *bam, int32_t i, char *name_out, int32_t cap, int64_t *len_out);
int fpt_bam_read(fpt_bam *bam, int64_t max_reads, int32_t *ref_id, int32_t *ref_start, int32_t *ref_end,
                 uint16_t *flag, uint8_t *mapq, int64_t *n_out);

/* Region access, what the reference does per interval (samfile.fetch(chrom, start - 10, end + 10),
 * cutcounts.py:191): with a BAI index beside the file (<path>.bai or <path without .bam>.bai; SAM
 * specification 5.2) fpt_bam_seek_region positions the reader on the first alignment that can
 * overlap [beg, end) of reference ref_id (linear index), and fpt_bam_read then hands out the
 * alignments of that reference that start before `end`, returning 0 alignments after the last.
 * Alignments that end before `beg` may come along (the index works in 16 kb windows).
 * fpt_bam_seek_region fails with FPT_ERR_INVALID when the file has no index. */
int fpt_bam_has_index(fpt_bam *bam, int32_t *yes_out);
int fpt_bam_seek_region(fpt_bam *bam, int32_t ref_id, int64_t beg, int64_t end);
/* The alignment records themselves, each behind its 4-byte block_size as in the file (SAM/BAM specification 4.2),
 * up to max_reads of them or as many as fit `cap` bytes: for a caller that needs more of an alignment than its
 * coordinates (name, mate flags, template length, bases, base qualities, tags -- what pysam's AlignedSegment carries in
 * cutcounts.py:170-229).  Same walk and region rule as fpt_bam_read (the two share the reader's position).  If the
 * first record does not fit `cap`, FPT_ERR_INVALID with *n_out = 0 and *bytes_out = the bytes that record needs. */
int fpt_bam_read_raw(fpt_bam *bam, int64_t max_reads, uint8_t *buf, int64_t cap, int64_t *n_out, int64_t *bytes_out);

/* Alignments -> cut counts of a batch of intervals, added into the padded CSR count arrays the
 * fused scan reads (zero them first; several calls accumulate, e.g. one per fpt_bam_read batch).
 * All pointers are device pointers.  Intervals are given by ascending start_key =
 * (reference id << 32 | padded start), padded start = start - pad - 1 (modeling/predict.pyx:
 * 132-134), with maxend_key[i] = max over j <= i of (reference id << 32 | padded start + padded
 * len); an alignment lands in every interval whose padded range holds its cut position. */
typedef struct fpt_cutcount_desc {
    int64_t n_reads;
    const int32_t *ref_id, *ref_start, *ref_end;
    const uint16_t *flag;
    const uint8_t *mapq;
    int32_t offset_plus, offset_minus;               /* bamfile(offset=(0, -1)) */
    int32_t min_qual, remove_dups, remove_qcfail;    /* bamfile(min_qual=1, remove_dups=False, remove_qcfail=True) */
    int64_t n_intervals;
    const int64_t *start_key, *maxend_key;
    const int32_t *padded_len;
    const int64_t *counts_off;                       /* element offset of each interval in the count arrays */
    double *counts_plus, *counts_minus;
    const uint8_t *flip;                             /* optional, per interval: 1 = strand '-' interval: the reference
                                                      * returns {'+': rev[::-1], '-': fw[::-1]} (cutcounts.py:307-311) */
} fpt_cutcount_desc;
int fpt_cut_counts_dev(fpt_ctx *ctx, const fpt_cutcount_desc *d);

/* Sequence gather on the device (the FASTA half): `seq_out` = the ASCII bytes
 * of every interval's [start, start + len) of its chromosome, back to back at the given offsets --
 * what `fasta_func.fetch(chrom, start, end)` returns per interval (modeling/predict.pyx:136-140;
 * the scan wants start - pad - 1 - 3 .. end + pad + 3) -- from the bytes of a FASTA file resident
 * on the device.  intervals_dev: DEVICE int64[n_intervals][7] = start, len, output offset, and the
 * chromosome's .fai fields: length, byte offset of its first base, bases per line, bytes per
 * line.  Positions outside the chromosome read 'N'. */
int fpt_seq_gather_dev(fpt_ctx *ctx, const uint8_t *fasta_dev, int64_t fasta_bytes, const int64_t *intervals_dev,
                       int64_t n_intervals, uint8_t *seq_out_dev);

/* ---- statistics tracks (host only): indexed region access to the
 * bgzip-compressed bedGraph files `ftd detect` writes (cli/utils.py:119-144), which the posterior
 * caller reads back per interval through pysam.TabixFile.fetch (cli/post.py:52-87).  The file is
 * mapped; `<path>.tbi` (tabix's linear index) is used when present, else an index of the same shape
 * is built by one pass at open.  Rows are selected by their start column: start <= row start < end.
 * fpt_track_fetch serves a batch of intervals on a team of threads: for every row of interval i,
 * out[c][out_off[i] + (row start - starts[i])] = the value of file column cols[c] (0-based; nan
 * where it does not parse) and present[...] = 1 -- the loop of `_load_data` (cli/post.py:70-83);
 * the caller pre-fills the defaults.  fpt_track_fetch_rows returns the rows of one region
 * (*n_out = rows found; only the first `cap` are stored). */
typedef struct fpt_track fpt_track;
int fpt_track_open(const char *path, fpt_track **out);
int fpt_track_close(fpt_track *t);
int fpt_track_n_refs(fpt_track *t, int32_t *n_out, int32_t *indexed_out /* 1 = a .tbi was used */);
int fpt_track_ref(fpt_track *t, int32_t i, char *name_out, int32_t cap);
int fpt_track_fetch(fpt_track *t, int64_t n_intervals, const char *const *chroms, const int64_t *starts,
                    const int64_t *ends, const int64_t *out_off, int32_t n_cols, const int32_t *cols,
                    double *const *out, double *present);
int fpt_track_fetch_rows(fpt_track *t, const char *chrom, int64_t start, int64_t end, int32_t n_cols,
                         const int32_t *cols, int64_t cap, int64_t *pos_out, double *vals_out, int64_t *n_out);

/* The other direction: bedGraph text (position-sorted lines <chrom> TAB <start> TAB <end> TAB ...,
 * what fpt_format_stats / write_stats_to_output produce, cli/utils.py:119-144; '#' lines are
 * header) written as a bgzip-compressed file `path` and its tabix index `path`.tbi (BED preset) --
 * what the reference's workflow gets from the external `bgzip` and `tabix -p bed` before
 * cli/post.py reads the track back (cli/post.py:52-55).  Text arrives in pieces of any size (a
 * line may be split between two calls); members are deflated on a team of threads
 * (FPT_TRACK_THREADS).  fpt_track_writer_close writes the end-of-file member and the index, frees
 * the writer and reports the first error (unsorted lines, a malformed line, a failed write). */
typedef struct fpt_track_writer fpt_track_writer;
int fpt_track_writer_open(const char *path, fpt_track_writer **out);
/* zlib level of the members written from now on (0..9; 6 = what bgzip uses, the default; 1 is ~3x
 * faster for ~20 % more bytes -- on a per-base track deflate is what the writer waits for) */
int fpt_track_writer_set_level(fpt_track_writer *w, int32_t level);
int fpt_track_writer_write(fpt_track_writer *w, const char *text, int64_t n_bytes);
int fpt_track_writer_close(fpt_track_writer *w);

/* ---- the one collective of the sharded job (BASELINE.json north_star: "a
 * single RCCL all-gather over xGMI at the end to reassemble the per-base statistics track").
 * Intervals shard across GPUs with no communication during the scan (one process and one
 * context per GPU); afterwards every rank contributes its slice of a per-base track and receives
 * all slices in rank order.  RCCL is bound directly (librccl.so through dlopen at the first
 * call; FPT_RCCL_LIB names another library with the same entry points -- the test suite's stand-in):
 * rank 0 makes an id with fpt_comm_unique_id, the host program carries the 128 bytes to
 * the other ranks (file, socket, environment), every rank calls fpt_comm_init.
 * Reference counterpart: the worker processes of cli/detect.py:380-411 hand their per-interval
 * statistics to one writer; nothing in the reference exchanges arrays between devices. */
#define FPT_COMM_ID_BYTES 128
typedef struct fpt_comm fpt_comm;
int fpt_comm_unique_id(uint8_t id_out[FPT_COMM_ID_BYTES]);
int fpt_comm_init(fpt_ctx *ctx, const uint8_t id[FPT_COMM_ID_BYTES], int world_size, int rank, fpt_comm **out);
int fpt_comm_destroy(fpt_comm *comm);
/* What the communicator says about the job it belongs to: the arguments of fpt_comm_init beside what
 * RCCL itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice; -1 where the bound library
 * lacks the symbol) and the PCI bus id of this rank's device.  A launcher's line can then prove that
 * RCCL saw N ranks on N different devices (bench.py gathers these with the tiny all-gather).
 * Reference counterpart: none (cli/detect.py:394 counts worker processes). */
typedef struct fpt_comm_info_t {
    int32_t world_size, rank, device;                   /* as given to fpt_comm_init / the context's device */
    int32_t rccl_count, rccl_user_rank, rccl_device;    /* as RCCL reports them */
    char pci_bus_id[32];                                /* "0000:05:00.0" */
} fpt_comm_info_t;
int fpt_comm_info(fpt_comm *comm, fpt_comm_info_t *out);
/* counts[r] = doubles contributed by rank r (world_size entries, the same on every rank); send =
 * this rank's counts[rank] doubles, recv = sum(counts) doubles (device pointers; send may lie
 * inside recv at its own offset).  Equal counts: one ncclAllGather; ragged: one ncclBroadcast per
 * shard inside a group.  Enqueued on the context's stream; does not synchronise. */
int fpt_allgather_track(fpt_ctx *ctx, fpt_comm *comm, const double *send, const int64_t *counts, double *recv);
/* The same shards to ONE rank -- the rank that writes the track, as the reference's single writer
 * thread does (cli/detect.py:396-408): grouped ncclSend / ncclRecv.  `recv` (sum(counts) doubles) is
 * read on the root only and may be NULL elsewhere; the root's own shard may already lie in place.
 * 1/world_size of the all-gather's traffic; all of it arrives on the root's links. */
int fpt_gather_track(fpt_ctx *ctx, fpt_comm *comm, const double *send, const int64_t *counts, double *recv, int root);
/* Both collectives beside the scan of the NEXT batch: enqueued on the communicator's own stream,
 * ordered behind everything the context's stream holds at the call (an event), so the track of batch k
 * travels while batch k + 1 is scanned into another buffer.  fpt_comm_wait makes the context's stream
 * wait for one of them -- back = 0: the last one enqueued, 1: the one before it, up to 3 (with two
 * track buffers in turn, wait with back = 1 before scanning into a buffer again: its collective was
 * the one before the last); fpt_comm_synchronize blocks the host until the last one is done.  Neither
 * is needed for the plain forms. */
int fpt_allgather_track_async(fpt_ctx *ctx, fpt_comm *comm, const double *send, const int64_t *counts, double *recv);
int fpt_gather_track_async(fpt_ctx *ctx, fpt_comm *comm, const double *send, const int64_t *counts, double *recv, int root);
int fpt_comm_wait(fpt_ctx *ctx, fpt_comm *comm, int back);
int fpt_comm_synchronize(fpt_comm *comm);

/* Diagnostics of the most recent fpt_scan_dev in memo mode (synchronises): tiles launched, tiles
 * the first pass handed to the general kernel, and the largest (exp, obs) pair that missed the
 * first-level table (-1, -1: none; what sized the second-level table). */
int fpt_scan_stats(fpt_ctx *ctx, int64_t *tiles_out, int64_t *redone_out, int32_t miss_max_out[2]);

/* Heavy-tailed variant of the synthetic workload: adds hotspot bursts to counts made by
 * fpt_synth_dev (in place).  Interval iv = position / padded_len carries a hotspot with probability
 * per_mille / 1000: a triangular bump 80..199 positions wide with a peak of 100..499 cuts per
 * strand (observed counts up to ~1000, far outside the first-level (exp, obs) table), decided by
 * a hash of (seed, iv) so that both strands, every rank and the CPU oracle agree. */
int fpt_synth_hotspots_dev(fpt_ctx *ctx, uint64_t seed, int64_t pos0_counts, int64_t n_counts,
                           int32_t padded_len, int32_t per_mille, double *counts_plus,
                           double *counts_minus);

/* 64-bit order-independent checksum (sum of value bit patterns mod 2^64) of a device track,
 * written to *host_out after synchronising; for size-independent parity checks. */
int fpt_checksum_dev(fpt_ctx *ctx, const double *dev, int64_t n, uint64_t *host_out);

/* The scan's HBM access pattern and nothing else (measurement; SURVEY.md 8d: "record the box's
 * measured stream-copy BW beside" the 8 TB/s peak).  Per interval the loads and stores of
 * fpt_scan_dev -- 2 x (L + 2*pad + 1) doubles and L + 2*pad + 7 sequence bytes in, n_tracks
 * tracks of L doubles total_bases apart out, one workgroup per interval -- with one add per load
 * between them.  Same input layout as fpt_scan_dev (interval_len for uniform batches, or
 * interval_off_dev + max_interval_len for ragged ones); `out` holds n_tracks * total_bases doubles
 * and is overwritten.  Runs one warm-up launch and `reps` timed ones on the context's stream,
 * synchronises, and writes the mean milliseconds per launch (HIP events) to *ms_out.
 * Reference counterpart: none -- the memory side of cli/detect.py:120-130 for a batch. */
int fpt_stream_pattern_dev(fpt_ctx *ctx, int64_t n_intervals, int32_t interval_len, const int64_t *interval_off_dev,
                           int32_t max_interval_len, int32_t pad, int32_t n_tracks, const double *counts_plus,
                           const double *counts_minus, const uint8_t *seq, double *out, int64_t total_bases,
                           int32_t reps, float *ms_out);

/* device memory helpers for hosts that have no allocator of their own (ctypes callers) */
int fpt_dev_alloc(fpt_ctx *ctx, int64_t bytes, void **dev_out);
int fpt_dev_free(fpt_ctx *ctx, void *dev);
/* ---- output text (host only) -------------------------------------------------
 * The bedGraph lines of `write_stats_to_output` (cli/utils.py:119-163): for every selected row i of
 * the n_rows x n_cols matrix `stats` (row-major),
 *     chrom <d> start+i <d> start+i+1 <d> "{:0.<precision>f}".format(v) of each column joined by <d> "\n"
 * into `buf` (cap bytes; *len_out = bytes written; FPT_ERR_INVALID when it does not fit -- a value
 * needs at most 1 + 17 + 1 + precision bytes while |v| < 1e17, up to 311 + precision beyond).
 * rows = NULL selects every row (the reference's filter_fn is applied by the caller).  The decimal
 * expansion is the correctly rounded one Python prints; nan / inf / -inf as Python spells them. */
int fpt_format_stats(const char *chrom, int64_t start, const double *stats, int64_t n_rows, int32_t n_cols,
                     const int64_t *rows, int64_t n_sel, char delim, int32_t precision, char *buf, int64_t cap,
                     int64_t *len_out);

/* The same lines for a whole batch of intervals (what one step of detect's batch_iter yields,
 * cli/detect.py:399-411 writes them interval by interval): `stats` holds the rows of interval j at
 * [row_off[j], row_off[j+1]) (n_intervals + 1 offsets), its lines start at position start[j] of
 * chromosome chrom_names[chrom_id[j]].  Intervals are formatted on a team of threads
 * (FPT_TEXT_THREADS) and joined in order.  *len_out is set to the size of the text also when it
 * does not fit `cap` (FPT_ERR_INVALID then: come back with a buffer of that size). */
int fpt_format_stats_batch(int64_t n_intervals, const char *const *chrom_names, int32_t n_chroms,
                           const int32_t *chrom_id, const int64_t *start, const int64_t *row_off,
                           const double *stats, int32_t n_cols, char delim, int32_t precision, char *buf,
                           int64_t cap, int64_t *len_out);
/* ... and straight into a track writer (TAB-delimited), without the text passing through the host
 * program: what `detect` followed by bgzip + tabix comes to. */
int fpt_track_writer_write_stats(fpt_track_writer *w, int64_t n_intervals, const char *const *chrom_names,
                                 int32_t n_chroms, const int32_t *chrom_id, const int64_t *start,
                                 const int64_t *row_off, const double *stats, int32_t n_cols, int32_t precision);

/* bytes of device memory set to zero on the context's stream (not synchronised) */
int fpt_dev_zero(fpt_ctx *ctx, void *dev, int64_t bytes);
int fpt_memcpy_h2d(fpt_ctx *ctx, void *dev, const void *host, int64_t bytes);
int fpt_memcpy_d2h(fpt_ctx *ctx, void *host, const void *dev, int64_t bytes);

/* timing of the most recent fpt_scan_dev launch sequence measured with HIP events on the
 * context's stream (valid after fpt_ctx_synchronize): milliseconds of the fused kernel. */
int fpt_last_scan_ms(fpt_ctx *ctx, float *ms_out);

/* Per-launch timing for benchmarks: after fpt_timing_enable(ctx, n) each fpt_scan_dev records
 * HIP events on the context's stream (up to n scans; further scans fall back to the single
 * pair behind fpt_last_scan_ms).  fpt_timing_read synchronises the stream and writes TWO floats
 * per recorded scan (at most cap scans): the milliseconds of the whole launch sequence (memo
 * table build + scan passes) and of the dominant pass alone (the memo-only k_scan_fused
 * instance in memo mode, the full instance in direct mode); it resets the record count. */
int fpt_timing_enable(fpt_ctx *ctx, int max_records);
int fpt_timing_read(fpt_ctx *ctx, float *ms_out, int cap, int *n_out);
/* Marks for benchmarks: fpt_mark records a HIP event on the context's stream (where every kernel of the
 * library is launched) and returns its number; fpt_mark_elapsed waits for mark `to` and gives the
 * milliseconds between two marks -- the duration of whatever calls were enqueued between them (an
 * fpt_fdr_dev or fpt_posterior_dev launch sequence), measured on the device.  fpt_marks_clear frees
 * the events.  No reference counterpart (measurement only). */
int fpt_mark(fpt_ctx *ctx, int32_t *id_out);
int fpt_mark_elapsed(fpt_ctx *ctx, int32_t from, int32_t to, float *ms_out);
int fpt_marks_clear(fpt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* FPT_H */
