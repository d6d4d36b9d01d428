/*
 * fpt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, float64, single-threaded per call) of the
 * footprint-tools per-nucleotide expected-cleavage / deviation-statistics path.
 * It exists to CHECK the HIP path; nothing in footprint_tools_amd/ may link,
 * import or call it.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py use it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py)
 * against golden vectors produced by the real reference (Cython/C build of
 * /root/reference, see oracle/pyref/ and tests/golden/make_golden.py) and, when
 * oracle/_ref/libfpt_ref.so is present, against the reference's own C sources
 * compiled where they lie.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef FPT_ORACLE_H
#define FPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- special functions (hcephes v0.4.1 subset reachable from footprint_tools) */
double orc_polevl(double x, const double *c, int n);   /* hcephes/src/polyn/polevl.c:3-17  */
double orc_p1evl(double x, const double *c, int n);    /* hcephes/src/polyn/polevl.c:19-33 */
double orc_gamma(double x);                            /* hcephes/src/cprob/gamma.c:51-127 */
double orc_lgam(double x);                             /* hcephes/src/cprob/gamma.c:147-235 */
double orc_incbet(double a, double b, double x);       /* hcephes/src/cprob/incbet.c:12-94 */
double orc_ndtr(double a);                             /* hcephes/src/cprob/ndtr.c:34-59   */
double orc_erf(double x);                              /* hcephes/src/cprob/ndtr.c:79-87   */
double orc_erfc(double a);                             /* hcephes/src/cprob/ndtr.c:89-132  */
double orc_erfce(double x);                            /* hcephes/src/cprob/ndtr.c:65-77   */
double orc_expx2(double x, int sign);                  /* hcephes/src/cprob/expx2.c:6-34   */
double orc_ndtri(double y0);                           /* hcephes/src/cprob/ndtri.c:48-88  */
double orc_log1p(double x);                            /* hcephes/src/cprob/unity.c:29-37  */
double orc_igam(double a, double x);                   /* hcephes/src/cprob/igam.c:76-100  */
double orc_igamc(double a, double x);                  /* hcephes/src/cprob/igam.c:6-63    */
double orc_chdtrc(double df, double x);                /* hcephes/src/cprob/chdtr.c:3-10   */

/* vectorised helpers for lattice tests: op = 0 gamma,1 lgam,2 ndtr,3 ndtri,4 log1p,5 erf,6 erfc */
void orc_map1(int op, const double *x, int64_t n, double *out);
void orc_incbet_v(const double *a, const double *b, const double *x, int64_t n, double *out);
void orc_chdtrc_v(const double *df, const double *x, int64_t n, double *out);

/* ---- 6-mer bias lookup: modeling/bias.py:88-111,16-17; modeling/predict.pyx:47-61,150-153 */
/* seq has seq_len bytes (ASCII, any case); writes l = seq_len-6 entries per strand.
 * idx_fwd/idx_rev may be NULL; index -1 = k-mer contains a non-ACGT byte (-> dflt). */
void orc_kmer_probs(const uint8_t *seq, int64_t seq_len, const double *table4096, double dflt,
                    double *fwd, double *rev, int32_t *idx_fwd, int32_t *idx_rev);

/* ---- expected cleavage: modeling/predict.h:23-74 + modeling/smoothing.h:11-133 */
void orc_fast_predict(const double *obs, const double *probs, int l, int hw, int shw,
                      double clip, double *exp_out, double *win_out);

/* ---- dispersion model: modeling/dispersion.pyx:26-57,127-163 */
double orc_fit_mu(const double *mu_par9, double x);
/* returns 0 ok, 1 = ZeroDivisionError (piecewise value exactly 0) */
int orc_fit_r(const double *r_par15, double x, double *r_out);

/* ---- NB scalars: stats/distributions/nbinom.pyx:82-138 */
double orc_nb_logpmf(int32_t k, double p, double r);
double orc_nb_pmf(int32_t k, double p, double r);
double orc_nb_cdf(int32_t k, double p, double r);
int32_t orc_c_int(double v); /* `<int>obs[i]` on x86-64 (cvttsd2si) */

/* what: 0 = p_values (dispersion.pyx:291-316), 1 = log_pmf_values (:170-196), 2 = pmf_values (:199-225)
 * returns 0 ok, 1 = ZeroDivisionError raised at some element (output undefined from there) */
int orc_nb_values(int what, const double *mu_par9, const double *r_par15, const double *exp,
                  const double *obs, int64_t n, double *out);

/* ---- sliding windows: stats/windowing.h:11-122 + stats/windowing.pyx:34-58,132-158 */
enum { ORC_WIN_SUM = 0, ORC_WIN_PRODUCT = 1, ORC_WIN_FISHER = 2, ORC_WIN_STOUFFER = 3,
       ORC_WIN_WSTOUFFER = 4 };
void orc_window(int op, const double *x, const double *w, int n, int hw, double *out);

/* ---- stats/utils.pyx:52-79, stats/fdr/__init__.py:12-33, stats/utils.pyx:15-50 */
void orc_bisect(const double *a, int na, const double *b, int nb, double *out);
void orc_emperical_fdr(const double *pvals_null, int64_t n_null, const double *pvals, int n,
                       double *out);
/* returns number of segments written (pairs in seg[2*i], seg[2*i+1]); cap = max pairs */
int orc_segment(const double *x, int n, double threshold, int w, int decreasing, int32_t *seg,
                int cap);

/* ---- cli/learn_dm.py:276-287: the (expected, observed) histogram; returns the pairs that are not finite
 * (the reference raises on those) */
int64_t orc_hist2d(const double *ex, const double *ob, int64_t n, int rows, int cols, int64_t *hist);

/* ---- composite per-interval path: cli/detect.py:120-130 (predict -> merge -> p_values -> stouffers_z)
 * counts_* have l = L + 2*(hw+shw) + 1 entries, seq has l + 6 bytes.
 * Outputs have L entries; winp has n_scales rows of L.  returns 0 ok, 1 ZeroDivisionError. */
int orc_detect_interval(const double *counts_plus, const double *counts_minus, const uint8_t *seq,
                        int L, int hw, int shw, double clip, const double *table4096, double dflt,
                        const double *mu_par9, const double *r_par15, const int32_t *scales,
                        int n_scales, double *exp_out, double *obs_out, double *p_out,
                        double *winp_out);

/* batch of equal-length intervals laid out back to back; n_threads > 1 uses OpenMP if built with it */
int orc_detect_batch(const double *counts_plus, const double *counts_minus, const uint8_t *seq,
                     int64_t n_iv, int L, int hw, int shw, double clip, const double *table4096,
                     double dflt, const double *mu_par9, const double *r_par15,
                     const int32_t *scales, int n_scales, double *exp_out, double *obs_out,
                     double *p_out, double *winp_out, int n_threads);

/* ---- posterior: stats/posterior.py:93-121 (one dataset row) */
int orc_log_likelihood_row(const double *mu_par9, const double *r_par15, const double *obs,
                           const double *exp, const double *delta, int n, int w, double *out);

/* ---- empirical FDR with the library's reproducible (Philox, inverse-CDF) null sampler */
void orc_philox_raw(uint32_t *c4, uint32_t k0, uint32_t k1); /* Philox4x32-10 */
double orc_philox_uniform(uint64_t seed, uint64_t base, uint32_t sample);
int orc_null_alias_row(const double *mu_par9, const double *r_par15, double ex, int table_k, uint32_t *entry_out,
                       double *cdf_out);
void orc_null_draws(const double *mu_par9, const double *r_par15, double ex, const double *u, int64_t n,
                    int table_exp, int table_k, int32_t *k_out, double *p_out);
void orc_fdr_null(const double *mu_par9, const double *r_par15, const double *exp_, const double *winp,
                  int L, int hw, int times, uint64_t seed, int64_t base0, const double *uniforms,
                  int table_exp, int table_k, double *efdr_out, double *null_out);

/* ---- synthetic inputs (bench-defined): splitmix64 counter hash */
uint64_t orc_splitmix64(uint64_t x);
void orc_synth_hotspots(uint64_t seed, int64_t pos0, int64_t n, int stream, int padded_len, int per_mille,
                        double *counts);
void orc_synth_fill(uint64_t seed, int64_t pos0, int64_t n, int stream, double *counts,
                    uint8_t *bases);

#ifdef __cplusplus
}
#endif
#endif
