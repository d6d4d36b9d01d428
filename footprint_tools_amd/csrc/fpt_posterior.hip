// fpt_posterior.hip -- the multi-dataset posterior caller as ONE kernel (gfx950 / CDNA4, wave64).
//
// The reference computes, per interval and with numpy loops over the datasets (cli/post.py:98-124):
//   prior  = compute_prior_weighted(fdr, w)                         stats/posterior.py:12-42
//   delta  = compute_delta_prior(obs, exp, fdr, betas)              stats/posterior.py:45-90
//   ll_on  = log_likelihood(obs, exp, dm, delta=delta, w=3)         stats/posterior.py:93-121
//   ll_off = log_likelihood(obs, exp, dm, w=3)
//   post   = -posterior(prior, ll_on, ll_off); post[post <= 0] = 0  stats/posterior.py:124-149, post.py:121-122
// Here a workgroup owns a tile of one interval, a lane one base of it (plus `hw` halo bases either
// side for the likelihood windows), and loops over the datasets twice:
//   1. the two priors of its base -- counts over datasets, the Beta posterior mean and variance in
//      closed form (scipy.stats.beta.stats is a/(a+b) and ab/((a+b)^2 (a+b+1))), the latter only for the
//      datasets that are called at the base, each lane working through its own;
//   2. per chunk of datasets, the occupied log-pmf (exp x delta) of the bases that have a called dataset, the
//      (base, dataset) pairs dealt out densely over the lanes; then per dataset: both NB log-pmfs of the base
//      (dispersion.pyx:170-226 -> nbinom.pyx:82-100: lgam(k+r) - lgam(k+1) - lgam(r) + r log p + k log1p(-p);
//      the unoccupied one from a table) into LDS, one barrier, the 2*hw+1 window sums left to right like
//      windowing.h:11-23 (edges 1.0, windowing.pyx:51), log-sum-exp like numpy's logaddexp, clamp, store.
// The kernel runs by the number of wavefronts a CU holds (long dependent fp64 chains, a barrier per dataset): LDS
// per one-wavefront workgroup is kept at 5.4 KB for that (the posterior kernel).
// Tracks are dataset-major (D rows of sum(L) bases: lanes read consecutive doubles); the result
// is base-major (sum(L) rows of D values), which is what the reference's record holds
// (`post.T`) and what its writer prints per base.
//
// Cost: per dataset and base 5 lgam, 2 log, 2 log1p, 4 piecewise fits, exp + log1p -- vector fp64
// work, not HBM traffic (40 bytes per dataset-base): see DESIGN.md for the measured rate.
#include "fpt_kernels.hpp"

#include <cmath>
#include <cstdlib>

#include "fpt_device.hpp"

using namespace fptd;

namespace {

struct post_args {
    int64_t n_intervals;
    int32_t interval_len;
    const int64_t *interval_off;
    int64_t total_bases;
    int32_t n_datasets, hw;
    double cutoff, pseudocount;
    const double *obs, *exp, *fdr, *w;  // (D, total_bases)
    const double *models;               // D x 24 (mu 9, r 15)
    const double *betas;                // D x 2
    double *post_out;                   // (total_bases, D)
    double *prior_out, *delta_out, *ll_on_out, *ll_off_out;  // optional: (D, total) / (total) / (D, total) x 2
    int32_t *status_out;                // optional, per interval: 1 = a dispersion fit divided by zero
    // tables made by k_posterior_tables before the launch (or nullptr): the unoccupied log-pmf of every
    // integer (exp, k) pair below kTabExp x kTabObs per dataset, and lgam(k + 1) for k < kTabLgam
    const double *off_table;
    const double *lgam_table;
    // ragged batches: a workgroup takes ONE chunk of kPostChunkTiles tiles of one interval.  Launch A: workgroup b = chunk 0
    // of interval b (chunk_list null); launch B: the further chunks of the intervals longer than that, listed by
    // k_posterior_plan -- (interval, chunk) pairs, *chunk_count of them (the launch is sized by a bound: workgroups
    // beyond the count leave at once).  Uniform batches: blockIdx.y is the chunk.
    const int2 *chunk_list;
    const int32_t *chunk_count;
};

constexpr int kPostChunkTiles = 8;

// The unoccupied likelihood is the NB log-pmf at the expected count itself, an integer (the scan's
// exp track is a sum of two rounded values), so per dataset it is a function of the integer pair
// (exp, k): a table entry made by the very expression the kernel evaluates holds the same bits.
// lgam(k + 1) is a function of k alone.  Three of the five lgam evaluations, a log, a log1p and
// two piecewise fits per dataset-base become two gathers (L2-resident: 512 KB per dataset).
// A dispersion fit that divides by zero leaves kTabDirect in its row: the kernel evaluates those
// itself and reports them.
constexpr int kTabExp = 256, kTabObs = 256, kTabLgam = 4096;
constexpr long long kTabDirectBits = 0x7ff8000000abcdefll;  // a NaN no arithmetic produces

__device__ __forceinline__ double nb_logpmf_terms(double lg_kr, double lg_k1, double lg_r, double r, double p, int32_t k) {
#pragma clang fp contract(off)
    return ((lg_kr - lg_k1) - lg_r) + r * log(p) + (double)k * fptm::log1p_fn(-p);
}

// The same log-pmf with lgam(k + r) - lgam(r) taken as the logarithm of the rising factorial
// r (r + 1) ... (r + k - 1): for the counts of a footprint track (k of a few dozen at most) that is k
// multiplications and ONE logarithm where two lgam evaluations are two logarithms, two rational
// functions with a division each and two argument reductions whose trip counts differ from lane to
// lane (a wavefront runs the longest).  The factors carry q = mu / (r + mu) with them, so that
// k log1p(-p) = k log q is part of the same logarithm:
//     log pmf = log( prod_{j<k} (r + j) q ) - lgam(k + 1) + r log p,     p = r / (r + mu)
// Every factor lies between q r and min(r, mu) + k (with r in [1e-3, 1e7] and mu in [1e-3, 1e12] nothing
// over- or underflows for k <= kProdMax), the
// product is good to k ulps, and where gamma.c's two lgam values cancel (r large) this form is the
// more accurate one.  Against the reference's expression it differs by ~1e-14 absolute per value
// (the contract on the posterior is 1e-6 relative; tests/test_gpu_parity.py).  Counts beyond
// kProdMax and arguments that are not finite positive numbers take the reference's expression.
constexpr int kProdMax = 48;
__device__ __forceinline__ double nb_logpmf_any(double r, double mu, int32_t k, double lg_k1) {
    const double d = r + mu;
    // (r <= 1e7, round 6: beyond it the reference's own lgam(k + r) - lgam(r) is the difference of two values of 1e9 and
    // more -- an absolute error of 1e-6 and up that the product form does not have -- and parity means the reference's
    // value: tests/test_gpu_parity.py::test_posterior_large_r_and_infinite_prior, r = 1e9 .. 1e12)
    if ((uint32_t)k <= (uint32_t)kProdMax && r >= 1e-3 && r <= 1e7 && mu >= 1e-3 && mu <= 1e12) {
        double inv = __builtin_amdgcn_rcp(d);  // 2^-24, two Newton steps: below an ulp
        inv = fma(fma(-d, inv, 1.0), inv, inv);
        inv = fma(fma(-d, inv, 1.0), inv, inv);
        // p = r / d correctly rounded (one residual step on the refined reciprocal): r log p multiplies p's last
        // bit by r, and a quotient an ulp off the reference's r / (r + mu) would move the term by 1.1e-16 r
        double p = r * inv;
        p = fma(fma(-p, d, r), inv, p);
        const double q = mu * inv, c = r * q;
        double prod = 1.0, fj = 0.0;
        int j = 0;
        for (; j + 3 < k; j += 4) {  // four factors a trip: (r + j) q = fma(j, q, r q), the next ones that + q each
            const double t0 = fma(fj, q, c), t1 = t0 + q, t2 = t1 + q, t3 = t2 + q;
            prod *= (t0 * t1) * (t2 * t3);
            fj += 4.0;
        }
        for (; j < k; ++j) {
            prod *= fma(fj, q, c);
            fj += 1.0;
        }
        // (the product lies in [1e-150, 1e150] and p in [1e-15, 1): normal positive numbers, what log_pos_fast asks for)
        return (fptm::log_pos_fast(prod) - lg_k1) + r * fptm::log_pos_fast(p);
    }
    const double p = r / d;
    return nb_logpmf_terms(fptm::lgam((double)k + r), lg_k1, fptm::lgam(r), r, p, k);
}

__global__ void __launch_bounds__(256) k_posterior_tables(const double *__restrict__ models, int n_datasets, int d0,
                                                          double *__restrict__ off_table, double *__restrict__ lgam_table) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int d = d0 + (int)blockIdx.y;  // (a launch covers at most 32,768 datasets: gridDim.y is 16 bits)
    if (d == n_datasets) {  // the extra row of workgroups: lgam(k + 1)
        if (i < kTabLgam) lgam_table[i] = fptm::lgam((double)(i + 1));
        return;
    }
    if (i >= kTabExp * kTabObs) return;
    const double *mu9 = models + (size_t)d * 24, *r15 = mu9 + 9;
    const int32_t k = i % kTabObs;
    const double x = (double)(i / kTabObs);
    bool zd = false;
    const double r = fptm::fit_r(r15, x, &zd);
    const double mu = fptm::fit_mu(mu9, x);
    const double v = nb_logpmf_any(r, mu, k, fptm::lgam((double)fptm::wrap_inc(k)));
    off_table[(size_t)d * kTabExp * kTabObs + i] = zd ? __longlong_as_double(kTabDirectBits) : v;
}

// The piecewise fits of dispersion.pyx:26-57 are sums of mask x (y + k x) over the segments.  With finite
// parameters, ascending breakpoints and a finite x exactly one mask is 1 and every other term is an exact zero, so
// the sum IS the active segment's y + k x (the same two roundings): the segment's index is the number of
// breakpoints x has reached, and its two parameters are read by that index -- 14 instructions where the sum over
// five segments is ~70.  A NaN x reaches no breakpoint and gives NaN through segment 0, as the sum does.  Whether
// the datasets' parameters allow this is decided once per launch, on the host (`posterior_model_simple`); an x
// that is not finite is settled in fit_r_mu.
template <int NSEG>
__device__ __forceinline__ double piecewise_active(const double *par, double x) {
#pragma clang fp contract(off)
    int s = 0;
#pragma unroll
    for (int i = 0; i + 1 < NSEG; ++i) s += x >= par[i] ? 1 : 0;
    return par[NSEG + s] + par[2 * NSEG + s] * x;
}

// np.max(np.vstack([a, b]), axis=0) of two values: NaN wins (posterior.py:72)
__device__ __forceinline__ double np_max2(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }

// numpy's logaddexp (npy_logaddexp): the branch on the sign of x - y keeps exp's argument <= 0
// (one exp and one log1p for both signs: lanes of a wavefront fall on both sides, and as two branches each side
// ran its own ~80 instructions; the values are the branches')
__device__ __forceinline__ double np_logaddexp(double x, double y) {
    const double t = x - y;
    const bool pos = t > 0;
    const double r = (pos ? x : y) + fptm::log1p_unit_fast(exp(pos ? -t : t));
    if (x == y) return x + 0.6931471805599453094;  // also +-inf == +-inf
    return t == t ? r : t;  // NaN
}

// datasets whose results are staged in LDS before they are stored (one 32-byte piece of a base's row; eight
// cost 2 KB more LDS per wavefront and with it 8 % of the rate: three workgroups fewer on a CU)
constexpr int kPostChunk = 4;

// The workgroup's barrier.  A one-wavefront workgroup needs none: its LDS instructions are carried out in the order
// they were issued, for all lanes at once -- only the compiler has to keep that order.  (`__syncthreads()` also
// waits for every global load AND store of the wavefront still in flight, which is what a workgroup-wide release
// asks for; measured: 2 % of this kernel.)
template <int NT>
__device__ __forceinline__ void tile_sync() {
    if (NT == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// fit_r / fit_mu (dispersion.pyx:127-163) on x for one dataset.  SIMPLE: by the active segment; an x that is not
// finite makes every term of the reference's sums NaN (0 x inf in the segments it is not in), which its clamps
// turn into r = 1e-6 and mu = 0.1 without a ZeroDivisionError.
template <bool SIMPLE>
__device__ __forceinline__ void fit_r_mu(const double *mu9, const double *r15, double x, double *r, double *mu, bool *zero_div) {
    if (SIMPLE) {
        const double vr = piecewise_active<5>(r15, x);
        const double ir = 1.0 / vr;
        *r = ir > 0.0 ? ir : 1e-6;
        if (vr == 0.0) {
            *zero_div = true;
            *r = NAN;
        }
        const double vm = piecewise_active<3>(mu9, x);
        *mu = vm > 0.0 ? vm : 0.1;
        if (!(fabs(x) < fptm::kInf)) *r = 1e-6, *mu = 0.1;
    } else {
        *r = fptm::fit_r(r15, x, zero_div);
        *mu = fptm::fit_mu(mu9, x);
    }
}

// SIMPLE: every dataset's parameters are finite and its breakpoints ascending (posterior_model_simple, decided by the host
// for the launch): the fits are their active segment, and the reference's sums over the segments -- 16 more
// registers, a wavefront per SIMD -- are not part of the kernel.
template <int NT, bool SIMPLE>
__global__ void __launch_bounds__(NT) k_posterior(const post_args a) {
    extern __shared__ double smem[];
    const int D = a.n_datasets, hw = a.hw;
    double *lp = smem;                              // [2 buffers][on, off][NT]
    double *stage = lp + 4 * NT;                    // [NT][kPostChunk + 1]: the occupied log-pmfs, then the posteriors of a chunk of datasets
    double *dl = stage + NT * (kPostChunk + 1);     // [NT]: delta of every lane's base
    int *blist = reinterpret_cast<int *>(dl + NT);  // [NT] the lanes whose base has delta != 1, then [NT] their count
    // (the models and the Beta priors are read where they lie -- one address per wavefront in the passes over the
    // datasets, L1 hits in the dense ones: staged in LDS they were 1.7 KB of a one-wavefront workgroup's 9, and
    // this kernel runs by the number of wavefronts a CU holds: DESIGN.md section 4)
    const double *par = a.models;
    const double *beta = a.betas;
    const int tid = threadIdx.x;
    int64_t iv = blockIdx.x;
    int chunk = blockIdx.y;
    if (a.chunk_list) {  // launch B of a ragged batch: the listed chunks
        typedef const __attribute__((address_space(4))) int32_t kint;
        if ((int)blockIdx.x >= *(kint *)a.chunk_count) return;
        const int2 e = a.chunk_list[blockIdx.x];
        iv = e.x;
        chunk = e.y;
    }
    int64_t off;
    int L;
    if (a.interval_off) {
        off = a.interval_off[iv];
        L = (int)(a.interval_off[iv + 1] - off);
    } else {
        L = a.interval_len;
        off = iv * (int64_t)L;
    }
    const int64_t T = a.total_bases;
    const int TL = NT - 2 * hw;  // output bases per tile
    // this workgroup's chunk of the interval: kPostChunkTiles tiles from t_first on
    const int64_t t_first64 = (int64_t)chunk * kPostChunkTiles * TL;
    if (t_first64 >= L) return;  // (uniform batches whose last chunk is empty; the same for all lanes, ahead of any barrier)
    const int t_first = (int)t_first64, t_end = min(L, t_first + kPostChunkTiles * TL);

    bool zero_div = false;
    int round = 0;               // parity of the LDS buffer across tiles and datasets
    for (int t0 = t_first; t0 < t_end; t0 += TL) {
        const int u = t0 - hw + tid;  // this lane's base (halo lanes included)
        const bool valid = u >= 0 && u < L;
        const int64_t g = off + (valid ? u : 0);
        // ---- 1: the priors of this base (both are reductions over the datasets)
        double pr = 1.0, delta = 1.0;
        if (valid) {
            double k_called = 0.0, n_cov = 0.0, sw = 0.0, swm = 0.0;
            // Beta(k + beta_a, n - k + beta_b) with n = max(exp, obs): mean a / s and variance a b / (s^2 (s + 1)),
            // s = a + b, by the reference's own operations -- delta = sum(w mu) / sum(w) must come out as the
            // reference's to the bit: with one dataset called it is exactly that dataset's mean (w mu / w), often
            // a short fraction, and exp x delta then falls ON a breakpoint of the piecewise dispersion fits,
            // which jump there (a closed form one division shorter moved delta by an ulp and the posterior by
            // 1e-2: test_posterior_driver_against_the_reference_driver).  What IS left out: a dataset that is not
            // called here (fdr above the cutoff: weight 0) adds exactly nothing, so its three divisions and
            // square root are skipped -- and the called ones are few and scattered (a twentieth of the pairs of a
            // null-like track, yet in nearly every wavefront for every dataset), so a lane notes ITS called
            // datasets in a mask and then works through them on its own, in ascending order like the reference's
            // sum: a wavefront makes as many trips as its busiest lane has called datasets (two or three of eight),
            // not one per dataset.  scipy returns NaN outside the domain, and 0 x NaN stays NaN (posterior.py:72-88):
            // a NaN term makes the sums NaN wherever it stands, so those are added at once.
            for (int d0 = 0; d0 < D; d0 += 64) {
                unsigned long long called = 0ull;
                const int dn = D - d0 < 64 ? D - d0 : 64;
                for (int dd = 0; dd < dn; ++dd) {
                    const int d = d0 + dd;
                    const int64_t j = (int64_t)d * T + g;
                    const double f = a.fdr[j], ww = a.w[j], o = a.obs[j], e = a.exp[j];
                    if (f <= a.cutoff) k_called += 1.0;
                    n_cov += ww;
                    const double al = o + beta[2 * d], be = (np_max2(e, o) - o) + beta[2 * d + 1];
                    // (al = +inf: the reference's mean al / (al + be) is inf / inf, NaN even for a dataset that is
                    // not called -- 0 x NaN; be = +inf with a finite al gives mean 0 and adds nothing there)
                    if (!(al > 0.0 && be > 0.0 && al < fptm::kInf)) {
                        swm += NAN;
                        sw += f > a.cutoff ? 0.0 : NAN;
                    } else if (!(f > a.cutoff)) {
                        called |= 1ull << dd;
                    }
                }
                while (called) {
#pragma clang fp contract(off)
                    const int d = d0 + __ffsll((long long)called) - 1;
                    called &= called - 1ull;
                    const int64_t j = (int64_t)d * T + g;
                    const double o = a.obs[j], e = a.exp[j];
                    const double al = o + beta[2 * d], be = (np_max2(e, o) - o) + beta[2 * d + 1];
                    const double s = al + be, mu = al / s, var = al * be / ((s * s) * (s + 1.0));
                    const double wt = 1.0 / sqrt(var);
                    swm += wt * mu;
                    sw += wt;
                }
            }
            const double unocc = n_cov - k_called + a.pseudocount, occ = k_called + a.pseudocount;
            pr = unocc / (unocc + occ);
            delta = swm / sw;
            if (delta != delta) delta = 1.0;
            if (a.delta_out && tid >= hw && tid < NT - hw) a.delta_out[g] = delta;
        }
        // the lanes whose base has a called dataset (delta != 1): the occupied form of every dataset is evaluated
        // for those alone, DENSELY -- below, a chunk of datasets at a time
        dl[tid] = delta;
        if (tid == 0) blist[NT] = 0;
        tile_sync<NT>();
        const bool busy = valid && delta != 1.0;
        if (busy) blist[atomicAdd(&blist[NT], 1)] = tid;
        tile_sync<NT>();
        const int n_busy = blist[NT];
        const bool mine = valid && tid >= hw && tid < NT - hw;  // an output base of this tile
        const bool inside = mine && u >= hw && u < L - hw;       // its window fits the interval
        const double log_pr = log(pr), log_1mpr = log(1.0 - pr);
        // ---- 2: per dataset, both log-pmfs -> LDS -> window sums -> posterior
        for (int d = 0; d < D; ++d, ++round) {
            if (d % kPostChunk == 0) {
                // The occupied form (exp x delta: no table) of this chunk of datasets, for the busy lanes' bases: the
                // (base, dataset) pairs are dealt out to ALL lanes, a pair each -- a third of the bases of a null-like
                // track are busy, scattered over every wavefront, and evaluated in place each dataset's pass ran
                // the fits, the product loop and two logarithms for a third of its lanes.  The values wait in `stage`
                // at the place the base's posterior of that dataset goes to afterwards (same lane, read before
                // written).  (A barrier first: the rows of the chunk before are being stored from there.)
                tile_sync<NT>();
                const int nd = D - d < kPostChunk ? D - d : kPostChunk;
                for (int q = tid; q < n_busy * nd; q += NT) {
                    const int bi = q / nd, dd = q - bi * nd;
                    const int b = blist[bi], dq = d + dd;
                    const int64_t j = (int64_t)dq * T + off + (t0 - hw + b);
                    const double o = a.obs[j], x = a.exp[j] * dl[b];
                    const double *mu9 = par + dq * 24, *r15 = mu9 + 9;
                    const int32_t k = fptm::c_int(o);
                    const double lg_k1 = (a.lgam_table && (uint32_t)k < (uint32_t)kTabLgam)
                                             ? a.lgam_table[k] : fptm::lgam((double)fptm::wrap_inc(k));
                    double r, mu;
                    fit_r_mu<SIMPLE>(mu9, r15, x, &r, &mu, &zero_div);
                    stage[b * (kPostChunk + 1) + dd] = nb_logpmf_any(r, mu, k, lg_k1);
                }
                tile_sync<NT>();
            }
            double *lp_on = lp + (size_t)(round & 1) * 2 * NT, *lp_off = lp_on + NT;
            double v_on = 0.0, v_off = 0.0;
            if (valid) {
                const int64_t j = (int64_t)d * T + g;
                const double o = a.obs[j], e = a.exp[j];
                const int32_t k = fptm::c_int(o);
                // the unoccupied form from the table where (exp, k) is an integer pair inside it
                const int ei = (int)e;
                bool need_off = true;
                if (a.off_table && e >= 0.0 && e < (double)kTabExp && (double)ei == e && (uint32_t)k < (uint32_t)kTabObs) {
                    v_off = a.off_table[((size_t)d * kTabExp + ei) * kTabObs + k];
                    need_off = __double_as_longlong(v_off) == kTabDirectBits;
                }
                if (need_off) {  // (rare with the tables: a count or an expectation beyond them, a fit that divided by zero)
                    const double *mu9 = par + d * 24, *r15 = mu9 + 9;
                    const double lg_k1 = (a.lgam_table && (uint32_t)k < (uint32_t)kTabLgam)
                                             ? a.lgam_table[k] : fptm::lgam((double)fptm::wrap_inc(k));
                    double r, mu;
                    fit_r_mu<SIMPLE>(mu9, r15, e, &r, &mu, &zero_div);
                    v_off = nb_logpmf_any(r, mu, k, lg_k1);
                }
                // delta is exactly 1 wherever no dataset is called at the base (most of a real track): the occupied
                // form is then the unoccupied one, value for value -- the same expression on the same arguments
                v_on = busy ? stage[tid * (kPostChunk + 1) + (d % kPostChunk)] : v_off;
            }
            lp_on[tid] = v_on;
            lp_off[tid] = v_off;
            tile_sync<NT>();  // the other buffer is free again once every lane is past the NEXT barrier
            if (mine) {
                double ll_on = 1.0, ll_off = 1.0;  // windowing.pyx:51: edges keep the 1.0 of np.ones
                if (inside) {
                    ll_on = ll_off = 0.0;
                    for (int j = tid - hw; j <= tid + hw; ++j) {
                        ll_on += lp_on[j];
                        ll_off += lp_off[j];
                    }
                }
                const int64_t j = (int64_t)d * T + g;
                // (w is read again rather than remembered from the first loop: a bit per dataset in a
                // register capped the datasets at 64, and the reference's tutorials run hundreds)
                const bool unc = a.w[j] == 0.0;  // the prior of a dataset without a hotspot here is 1
                // posterior.py:140-149 with prior = 1 where the dataset has no hotspot (log 0 = -inf)
                const double p_off = (unc ? 0.0 : log_pr) + ll_off;
                const double p_on = (unc ? -fptm::kInf : log_1mpr) + ll_on;
                double post = -(p_off - np_logaddexp(p_on, p_off));
                if (post <= 0.0) post = 0.0;  // post.py:122 (a NaN stays)
                stage[tid * (kPostChunk + 1) + (d % kPostChunk)] = post;
                if (a.prior_out) a.prior_out[j] = unc ? 1.0 : pr;
                if (a.ll_on_out) a.ll_on_out[j] = ll_on;
                if (a.ll_off_out) a.ll_off_out[j] = ll_off;
            }
            // The record is base-major -- (sum(L), D), what the reference's writer prints per base -- so a
            // lane's own store would touch one 64-byte piece per lane and dataset (64 lines per store
            // instruction).  A chunk of kPostChunk datasets is staged instead and stored with the lanes
            // ALONG the rows: four lanes write 32 contiguous bytes of a base, a wavefront sixteen bases.
            if ((d % kPostChunk) == kPostChunk - 1 || d == D - 1) {
                tile_sync<NT>();
                const int d0 = d - (d % kPostChunk), nd = d - d0 + 1;
                const int n_out = min(TL, L - t0);          // output bases of this tile: lanes hw .. hw + n_out - 1
                for (int i = tid; i < n_out * kPostChunk; i += NT) {
                    const int b = i / kPostChunk, e = i % kPostChunk;
                    if (e < nd)
                        a.post_out[(off + t0 + b) * D + d0 + e] = stage[(hw + b) * (kPostChunk + 1) + e];
                }
                // (the next chunk's first write to `stage` comes after a barrier of its own)
            }
        }
    }
    if (zero_div && a.status_out) atomicOr(&a.status_out[iv], 1);
}

// The chunks beyond the first of every interval longer than kPostChunkTiles tiles, as (interval, chunk) pairs in the
// order the atomics hand out (a chunk's result does not depend on its place).  count must be zero.
__global__ void __launch_bounds__(256) k_posterior_plan(const int64_t *__restrict__ off, int64_t n_intervals, int tl,
                                                        int2 *__restrict__ list, int32_t *__restrict__ count, int32_t capacity) {
    const int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (iv >= n_intervals) return;
    const int64_t L = off[iv + 1] - off[iv];
    const int64_t span = (int64_t)kPostChunkTiles * tl;
    const int extra = L > span ? (int)((L - 1) / span) : 0;
    if (!extra) return;
    const int at = atomicAdd(count, extra);
    for (int c = 0; c < extra && at + c < capacity; ++c) list[at + c] = make_int2((int)iv, c + 1);
}

}  // namespace

namespace fptk {

// the extra chunks of a ragged batch are at most total_bases / (kPostChunkTiles x the SMALLEST tile any instance
// uses): every chunk beyond an interval's first has a full chunk of bases before it
size_t posterior_plan_bytes(int64_t total_bases, int hw) {
    const int tl = 64 - 2 * hw > 0 ? 64 - 2 * hw : 1;
    return (size_t)(total_bases / ((int64_t)kPostChunkTiles * tl) + 2) * sizeof(int2) + 16;
}

bool posterior_model_simple(const double *par24) {
    bool ok = true;
    for (int i = 0; i < 24; ++i) ok = ok && std::fabs(par24[i]) < HUGE_VAL;
    ok = ok && par24[0] <= par24[1];                                                        // mu: x0 <= x1 (x2 unused)
    ok = ok && par24[9] <= par24[10] && par24[10] <= par24[11] && par24[11] <= par24[12];   // r: x0 .. x3 (x4 unused)
    return ok;
}

size_t posterior_table_bytes(int n_datasets) {
    return ((size_t)n_datasets * kTabExp * kTabObs + kTabLgam) * sizeof(double);
}

size_t posterior_lds_bytes(int n_datasets, int nt) {
    (void)n_datasets;
    return (size_t)(4 * nt + nt * (kPostChunk + 1) + nt + nt / 2 + 1) * sizeof(double);
}

hipError_t launch_posterior(hipStream_t st, const posterior_launch &pl) {
    post_args a;
    a.n_intervals = pl.n_intervals;
    a.interval_len = pl.interval_len;
    a.interval_off = pl.interval_off;
    a.total_bases = pl.total_bases;
    a.n_datasets = pl.n_datasets;
    a.hw = pl.hw;
    a.cutoff = pl.cutoff;
    a.pseudocount = pl.pseudocount;
    a.obs = pl.obs;
    a.exp = pl.exp;
    a.fdr = pl.fdr;
    a.w = pl.w;
    a.models = pl.models;
    a.betas = pl.betas;
    a.post_out = pl.post_out;
    a.prior_out = pl.prior_out;
    a.delta_out = pl.delta_out;
    a.ll_on_out = pl.ll_on_out;
    a.ll_off_out = pl.ll_off_out;
    a.status_out = pl.status_out;
    a.off_table = pl.off_table;
    a.lgam_table = pl.lgam_table;
    a.chunk_list = nullptr;
    a.chunk_count = nullptr;
    if (pl.off_table && pl.lgam_table)
        for (int d0 = 0; d0 <= pl.n_datasets; d0 += 32768) {
            const int ny = pl.n_datasets + 1 - d0 < 32768 ? pl.n_datasets + 1 - d0 : 32768;
            hipLaunchKernelGGL(k_posterior_tables, dim3(kTabExp * kTabObs / 256, ny), dim3(256), 0, st, pl.models,
                               pl.n_datasets, d0, pl.off_table, pl.lgam_table);
        }
    // Batches of short intervals (the whole-genome hotspot set averages 162 bases) run one WAVEFRONT per
    // workgroup, which walks its interval in tiles of 64 - 2 hw bases: 162 + 6 positions fill 3 x 64 lanes
    // to 88 %, where one 256-lane tile was 66 % full (SQ_THREAD_CYCLES_VALU said 39 % of the lanes idle);
    // long intervals keep 256 lanes (a halo of 2 hw lanes per tile is 2 % there, 9 % of a wavefront);
    // an interval longer than 8 tiles is spread over gridDim.y
    const int64_t mean_len = pl.n_intervals > 0 ? pl.total_bases / pl.n_intervals : pl.max_len;
    int nt = mean_len + 2 * pl.hw <= 320 && pl.hw <= 8 ? 64 : 256;
    if (const char *e = getenv("FPT_POSTERIOR_NT")) nt = atoi(e) == 64 ? 64 : (atoi(e) == 128 ? 128 : 256);
    if (nt - 2 * pl.hw < 16) nt = 256;  // (a tile must hold more than its halo)
    const int tl = nt - 2 * pl.hw;
    const int64_t tiles = ((int64_t)pl.max_len + tl - 1) / tl;
    // Chunks of kPostChunkTiles tiles.  A uniform batch has the same number in every interval: gridDim.y.  A ragged
    // one (round 6): ONE workgroup per interval for its first chunk -- nine in ten intervals of the whole-genome
    // set have no other -- and a second launch over the listed further chunks of the long ones (k_posterior_plan);
    // rounds 3 - 5 sized gridDim.y by the LONGEST interval of the batch, and four workgroups in five left at once.
    int64_t gy64 = pl.interval_off ? 1 : (tiles + kPostChunkTiles - 1) / kPostChunkTiles;
    if (gy64 > 65535) return hipErrorInvalidValue;  // (an interval of > 30 million bases in a uniform batch)
    const int gy = (int)gy64;
    int2 *chunk_list = nullptr;
    int32_t *chunk_count = nullptr;
    int64_t chunk_cap = 0;
    if (pl.interval_off && (tiles > kPostChunkTiles || pl.max_len_unknown)) {
        if (!pl.plan_ws) return hipErrorInvalidValue;
        chunk_count = (int32_t *)pl.plan_ws;
        chunk_list = (int2 *)((char *)pl.plan_ws + 16);
        chunk_cap = pl.total_bases / ((int64_t)kPostChunkTiles * tl) + 1;
        if (chunk_cap > 0x7fffff00) return hipErrorInvalidValue;
        hipError_t e0 = hipMemsetAsync(chunk_count, 0, 16, st);
        if (e0 != hipSuccess) return e0;
        hipLaunchKernelGGL(k_posterior_plan, dim3((unsigned)((pl.n_intervals + 255) / 256)), dim3(256), 0, st, pl.interval_off,
                           pl.n_intervals, tl, chunk_list, chunk_count, (int32_t)chunk_cap);
    }
    const size_t lds = posterior_lds_bytes(pl.n_datasets, nt);
    void (*kern)(const post_args) = pl.all_simple ? (nt == 64 ? k_posterior<64, true> : (nt == 128 ? k_posterior<128, true> : k_posterior<256, true>))
                                                   : (nt == 64 ? k_posterior<64, false> : (nt == 128 ? k_posterior<128, false> : k_posterior<256, false>));
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    for (int64_t done = 0; done < pl.n_intervals; done += 0x7fffff00) {
        post_args b = a;
        const int64_t n = pl.n_intervals - done < 0x7fffff00 ? pl.n_intervals - done : 0x7fffff00;
        if (done) {  // a later chunk of a very large batch: shift the interval view
            if (b.interval_off) b.interval_off += done;
            else {
                const int64_t shift = done * (int64_t)b.interval_len;
                b.obs += shift, b.exp += shift, b.fdr += shift, b.w += shift;
                b.post_out += shift * b.n_datasets;
                if (b.prior_out) b.prior_out += shift;
                if (b.delta_out) b.delta_out += shift;
                if (b.ll_on_out) b.ll_on_out += shift;
                if (b.ll_off_out) b.ll_off_out += shift;
            }
            if (b.status_out) b.status_out += done;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)n, gy), dim3(nt), lds, st, b);
    }
    if (chunk_list) {  // launch B: sized by the bound, the count decides
        post_args b = a;
        b.chunk_list = chunk_list;
        b.chunk_count = chunk_count;
        hipLaunchKernelGGL(kern, dim3((unsigned)chunk_cap, 1), dim3(nt), lds, st, b);
    }
    return hipSuccess;
}

}  // namespace fptk
