/*
 * ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin exporter around the REFERENCE's own native sources, compiled where they
 * lie under /root/reference (nothing is copied): it #includes the header-only C
 * files footprint_tools/modeling/predict.h (+smoothing.h) and
 * footprint_tools/stats/windowing.h by include path and is linked with the
 * vendored hcephes sources.  Output goes to oracle/_ref/libfpt_ref.so
 * (git-ignored).  Used to validate oracle/fpt_oracle.c and as the "reference"
 * CPU timing of the native kernels.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "predict.h"   /* /root/reference/footprint_tools/modeling/predict.h */
#include "windowing.h" /* /root/reference/footprint_tools/stats/windowing.h  */

/* modeling/predict.pyx:23-45 does exactly this copy-out around fast_predict */
void ref_fast_predict(const double *obs, const double *probs, int l, int hw, int shw, double clip,
                      double *exp_out, double *win_out) {
    result_t *res = fast_predict(obs, probs, l, hw, shw, clip);
    memcpy(exp_out, res->exp, (size_t)l * sizeof(double));
    memcpy(win_out, res->win, (size_t)l * sizeof(double));
    free_result_t(res);
}

/* stats/windowing.pyx:34-58 / :132-158: ones outside [hw, n-hw) */
void ref_window(int op, const double *x, const double *w, int n, int hw, double *out) {
    double *res;
    if (op == 4)
        res = fast_weighted_windowing_func(x, w, n, hw, fast_weighted_stouffers_z);
    else
        res = fast_windowing_func(x, n, hw,
                                  op == 0 ? fast_sum
                                  : op == 1 ? fast_product
                                  : op == 2 ? fast_fishers_combined
                                            : fast_stouffers_z);
    for (int i = 0; i < n; i++) out[i] = 1.0;
    for (int i = hw; i < n - hw; i++) out[i] = res[i];
    free(res);
}

/* vector forms of the hcephes entry points the path reaches */
void ref_map1(int op, const double *x, long n, double *out) {
    for (long i = 0; i < n; i++) {
        double v = x[i];
        out[i] = op == 0   ? hcephes_gamma(v)
                 : op == 1 ? hcephes_lgam(v)
                 : op == 2 ? hcephes_ndtr(v)
                 : op == 3 ? hcephes_ndtri(v)
                 : op == 4 ? hcephes_log1p(v)
                 : op == 5 ? hcephes_erf(v)
                           : hcephes_erfc(v);
    }
}

void ref_incbet_v(const double *a, const double *b, const double *x, long n, double *out) {
    for (long i = 0; i < n; i++) out[i] = hcephes_incbet(a[i], b[i], x[i]);
}

void ref_chdtrc_v(const double *df, const double *x, long n, double *out) {
    for (long i = 0; i < n; i++) out[i] = hcephes_chdtrc(df[i], x[i]);
}
