// Test-only: compiles footprint_tools_amd/csrc/fpt_math.hpp with the HOST compiler so the
// device special functions can be checked against the oracle on a machine without a GPU.
// Not part of the product library.
#include "fpt_math.hpp"

extern "C" {
void hm_map1(int op, const double *x, long n, double *out) {
    for (long i = 0; i < n; i++) {
        double v = x[i];
        out[i] = op == 0   ? fptm::gamma_fn(v)
                 : op == 1 ? fptm::lgam(v)
                 : op == 2 ? fptm::ndtr(v)
                 : op == 3 ? fptm::ndtri(v)
                 : op == 4 ? fptm::log1p_fn(v)
                 : op == 5 ? fptm::erf_fn(v)
                 : op == 6 ? fptm::erfc_fn(v)
                 : op == 8 ? fptm::log_pos_fast(v)
                 : op == 9 ? fptm::log1p_unit_fast(v)
                           : fptm::ndtr_window(v);
    }
}
void hm_incbet(const double *a, const double *b, const double *x, long n, double *out) {
    for (long i = 0; i < n; i++) out[i] = fptm::incbet(a[i], b[i], x[i]);
}
void hm_chdtrc(const double *df, const double *x, long n, double *out) {
    for (long i = 0; i < n; i++) out[i] = fptm::chdtrc(df[i], x[i]);
}
unsigned hm_uniform_word(double u) { return fptm::uniform_word(u); }
// what: 0 cdf, 1 logpmf, 2 pmf; returns 1 if a zero division was flagged
int hm_nb_values(int what, const double *mu9, const double *r15, const double *e, const double *o,
                 long n, double *out) {
    bool zd = false;
    for (long i = 0; i < n; i++) {
        double r = fptm::fit_r(r15, e[i], &zd);
        double mu = fptm::fit_mu(mu9, e[i]);
        int32_t k = fptm::c_int(o[i]);
        double p = r / (r + mu);
        out[i] = what == 0 ? fptm::nb_cdf(k, p, r)
                 : what == 1 ? fptm::nb_logpmf(k, p, r)
                             : exp(fptm::nb_logpmf(k, p, r));
    }
    return zd ? 1 : 0;
}
}
