// fpt_kernels.hip -- HIP kernels (gfx950 / CDNA4, wave64) of the footprint scan.
//
// Kernels
//   k_kmer_probs      6-mer bias lookup, table staged in LDS            (bias.py / predict.pyx)
//   k_predict_rows    window sums + trimmed-mean smoothing + expected   (predict.h / smoothing.h)
//   k_nb_values       per-base NB cdf / logpmf / pmf                    (dispersion.pyx / nbinom.pyx)
//   k_nb_scalar       nbinom.{cdf,logpmf,pmf}(k,p,r)
//   k_window_rows     sliding-window reducers                           (windowing.h)
//   k_special         element-wise special functions (diagnostics)
//   k_scan_fused<NT>  the whole per-interval path in one pass           (cli/detect.py:120-130)
//   k_synth, k_checksum  synthetic workload + parity checksum
//
// Reference citations are paths under vierstralab/footprint-tools v1.3.7.
#include "fpt_kernels.hpp"

#include <type_traits>

#include <cstdlib>

#include "fpt_device.hpp"

using namespace fptd;

// ===========================================================================
// k_kmer_probs
// ===========================================================================
__global__ void __launch_bounds__(256) k_kmer_probs(const uint8_t *__restrict__ seq, int64_t n_out,
                                                    const double *__restrict__ table,
                                                    double *__restrict__ fwd,
                                                    double *__restrict__ rev) {
    __shared__ double tbl[kTable + 1];
    for (int i = threadIdx.x; i <= kTable; i += blockDim.x) tbl[i] = table[i];
    __syncthreads();
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += stride) {
        uint8_t codes[7];
#pragma unroll
        for (int m = 0; m < 7; ++m) codes[m] = (uint8_t)base_code(seq[j + m]);
        int fi, ri;
        kmer_indices(codes, fi, ri);
        if (fwd) fwd[j] = tbl[fi];
        if (rev) rev[j] = tbl[ri];
    }
}

// ===========================================================================
// k_predict_rows: one block = one tile of TL output positions of one row.
//   W[u]  = sum_{j=-hw}^{hw-1} obs[u+j]   u in [hw, l-hw), else 0       predict.h:41-48
//   W'[u] = trimmed mean of W[u-shw..u+shw], u in [shw, l-shw), else 0  smoothing.h:107-133
//   E[u]  = round(probs[u]/Q[u] * W'[u]),   u in [hw, l-hw), else 0     predict.h:60-63
// ===========================================================================
__global__ void __launch_bounds__(256) k_predict_rows(const double *__restrict__ obs,
                                                      const double *__restrict__ probs, int l,
                                                      int hw, int shw, int k_trim, int tile_len, int n_slots,
                                                      double *__restrict__ exp_out,
                                                      double *__restrict__ win_out) {
    extern __shared__ double s_w[];  // tile_len + 2*shw window sums | n_slots windows | 2 words
    const int64_t row = blockIdx.y;
    const double *o = obs + row * (int64_t)l;
    const double *p = probs + row * (int64_t)l;
    const int i0 = blockIdx.x * tile_len;
    const int nw = tile_len + 2 * shw;
    const int w = 2 * shw + 1;
    double *scratch = s_w + nw;
    unsigned long long *wmax_bits = reinterpret_cast<unsigned long long *>(scratch + (size_t)n_slots * w);
    int *counter = reinterpret_cast<int *>(wmax_bits + 1);
    if (threadIdx.x == 0) *wmax_bits = 0ull;
    __syncthreads();
    double mx = 0.0;
    for (int v = threadIdx.x; v < nw; v += blockDim.x) {
        int u = i0 - shw + v;
        double acc = 0.0;
        if (u >= hw && u < l - hw)
            for (int j = -hw; j < hw; ++j) acc += o[u + j];
        s_w[v] = acc;
        mx = acc != acc ? __longlong_as_double(0x7ff8000000000000ll) : fmax(mx, fabs(acc));
    }
    // largest |window sum| of the tile for the tolerance below (non-negative doubles order like
    // their bit patterns; a NaN lands on top and switches the re-evaluation off)
    atomicMax(wmax_bits, (unsigned long long)__double_as_longlong(mx));
    __syncthreads();
    const double w_max = __longlong_as_double((long long)*wmax_bits);
    const double w_div = (double)(w - 2 * k_trim);
    for (int first = 0; first < tile_len; first += blockDim.x) {  // same trip count for every lane: votes inside
#pragma clang fp contract(off)
        const int v = first + threadIdx.x;
        const int u = i0 + v;
        const bool live = u < l;
        const bool smoothed = live && shw > 0 && u >= shw && u < l - shw;
        double ws = 0.0, q = 0.0, e = 0.0;
        bool need = false;
        if (live) {
            if (shw > 0)
                ws = smoothed ? trimmed_mean(&s_w[v], w, k_trim) : 0.0;
            else
                ws = s_w[v];
            if (u >= hw && u < l - hw) {
                for (int j = -hw; j < hw; ++j) q += p[u + j];
                const double pq = p[u] / q, x = pq * ws;
                e = round(x);
                // Where the reference's rounding noise around the exact trimmed sum could decide
                // round(), the window goes through its order of operations (fpt_device.hpp); the
                // bound is the one of tie_ctx::tol_x with the summation noise always counted in.
                const double tol_t = w_max * ((double)((w + 256) * 256) * 1.1102230246251565e-16);
                need = smoothed && near_rounding_tie(x, fabs(pq) * tol_t / w_div + fabs(x) * 1e-13);
            }
        }
        if (shw > 0) {
            const double t = trimmed_sum_rounds(need, &s_w[need ? v : 0], w, k_trim, scratch, n_slots, counter, threadIdx.x);
            if (need) {
                ws = t / w_div;
                e = round((p[u] / q) * ws);
            }
        }
        if (live) {
            exp_out[row * (int64_t)l + u] = e;
            win_out[row * (int64_t)l + u] = ws;
        }
    }
}

// ===========================================================================
// k_nb_values / k_nb_scalar / k_special
// ===========================================================================
__global__ void __launch_bounds__(256, 4) k_nb_values(int what, const double *__restrict__ model,
                                                   const double *__restrict__ ex,
                                                   const double *__restrict__ ob, int64_t n,
                                                   double *__restrict__ out, int *__restrict__ flags) {
    __shared__ double par[24];
    if (threadIdx.x < 24) par[threadIdx.x] = model[threadIdx.x];
    __syncthreads();
    // one element per thread: a grid-stride loop around incbet makes LICM hoist every
    // polynomial coefficient into registers (VGPRs 118 -> 200+)
    bool zd = false;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double e = ex[i];
        double r = fptm::fit_r(par + 9, e, &zd);
        double mu = fptm::fit_mu(par, e);
        int32_t k = fptm::c_int(ob[i]);
        double pp = r / (r + mu);
        double v;
        if (what == 0) v = fptm::nb_cdf(k, pp, r);
        else {
            v = fptm::nb_logpmf(k, pp, r);
            if (what == 2) v = exp(v);
        }
        out[i] = v;
    }
    if (zd) atomicOr(flags, 1);
}

__global__ void __launch_bounds__(256, 4) k_nb_scalar(int what, const int32_t *__restrict__ k,
                                                   const double *__restrict__ p,
                                                   const double *__restrict__ r, int64_t n,
                                                   double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double v;
        if (what == 0) v = fptm::nb_cdf(k[i], p[i], r[i]);
        else {
            v = fptm::nb_logpmf(k[i], p[i], r[i]);
            if (what == 2) v = exp(v);
        }
        out[i] = v;
    }
}

// the table of fptm::ndtr_fast_tab (tools/fit_ndtr_gtab.py), for the special-function entry point that
// checks it; the scan kernels keep their own copy and stage it in LDS
__device__ const double g_ndtr_gtab[4 * FPT_NDTR_GTAB_N + 1] = {FPT_NDTR_GTAB_LIST};

__global__ void __launch_bounds__(256) k_special(int fn, const double *__restrict__ a,
                                                 const double *__restrict__ b,
                                                 const double *__restrict__ x, int64_t n,
                                                 double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double v = a[i], res;
        switch (fn) {
        case 0: res = fptm::gamma_fn(v); break;
        case 1: res = fptm::lgam(v); break;
        case 2: res = fptm::ndtr(v); break;
        case 3: res = fptm::ndtri(v); break;
        case 4: res = fptm::log1p_fn(v); break;
        case 5: res = fptm::erf_fn(v); break;
        case 6: res = fptm::erfc_fn(v); break;
        case 7: res = fptm::incbet(v, b[i], x[i]); break;
        case 9: res = fptm::ndtr_window(v); break;
        case 10: res = fabs(v) < fptm::kNdtrFastLimit ? fptm::ndtr_fast_tab(v, g_ndtr_gtab) : fptm::ndtr(v); break;
        case 11: res = fptm::log_pos_fast(v); break;
        case 12: res = fptm::log1p_unit_fast(v); break;
        default: res = fptm::chdtrc(v, x[i]); break;
        }
        out[i] = res;
    }
}

// ===========================================================================
// k_window_rows: out[i] = f(x[i-hw..i+hw]), i in [hw, n-hw); 1.0 elsewhere.
// The per-element transform (ndtri(1-x), log x) is evaluated once per element into LDS
// (the reference re-evaluates it for every window), then each output sums its window
// left to right like windowing.h:11-67 does.
// ===========================================================================
__global__ void __launch_bounds__(256) k_window_rows(int op, const double *__restrict__ x,
                                                     const double *__restrict__ w, int n, int hw,
                                                     int tile_len, double *__restrict__ out) {
    extern __shared__ double s_t[];  // [tile_len + 2*hw] (+ same again for weights^2)
    const int64_t row = blockIdx.y;
    const double *xr = x + row * (int64_t)n;
    const double *wr = w ? w + row * (int64_t)n : nullptr;
    const int i0 = blockIdx.x * tile_len;
    const int nt = tile_len + 2 * hw;
    double *s_w2 = s_t + nt;
    for (int v = threadIdx.x; v < nt; v += blockDim.x) {
        int i = i0 - hw + v;
        double t = 0.0, w2 = 0.0;
        if (i >= 0 && i < n) {
            double xv = xr[i];
            switch (op) {
            case 0: case 1: t = xv; break;
            case 2: t = log(xv); break;
            case 3: t = fptm::ndtri(1.0 - xv); break;
            default: {
                double wv = wr[i];
                t = wv * fptm::ndtri(1.0 - xv);
                w2 = wv * wv;
            }
            }
        }
        s_t[v] = t;
        if (op == 4) s_w2[v] = w2;
    }
    __syncthreads();
    const int k = 2 * hw + 1;
    for (int v = threadIdx.x; v < tile_len; v += blockDim.x) {
        int i = i0 + v;
        if (i >= n) break;
        double res = 1.0;
        if (i >= hw && i < n - hw) {
            const double *win = &s_t[v];
            if (op == 1) {
                double pr = 1.0;
                for (int j = 0; j < k; ++j) pr *= win[j];
                res = pr;
            } else {
                double s = 0.0;
                for (int j = 0; j < k; ++j) s += win[j];
                if (op == 0) res = s;
                else if (op == 2) res = fptm::chdtrc(2.0 * (double)k, s * -2.0);
                else if (op == 3) res = fptm::ndtr(-(s / sqrt((double)k)));
                else {
                    double sw = 0.0;
                    for (int j = 0; j < k; ++j) sw += s_w2[v + j];
                    res = fptm::ndtr(-(s / sqrt(sw)));
                }
            }
        }
        out[row * (int64_t)n + i] = res;
    }
}

// ===========================================================================
// k_scan_fused: the whole path for one tile of one interval per workgroup.
//
// A tile is `tl` consecutive output bases [t0, t0+tl) of interval iv.  With
// pad = hw+shw and H = the largest Stouffer half-width, the block stages into LDS
//   counts +/-            padded positions u in [ta, tb+2*pad+1)        (ta = max(0,t0-H),
//   sequence codes        u in [ta, tb+2*pad+7)                          tb = min(L,t0+tl+H))
// and runs, separated by workgroup barriers,
//   B  6-mer lookup from the LDS table, 2*hw-wide window sums of the counts, strand-merged obs
//   C  trimmed-mean smoothing, window sum of the propensities, expected = round(P/Q*W')
//   D  strand merge, NB lower-tail p-value, z = ndtri(1-p)      (one base per thread)
//   E  per-wavefront prefix scan of z (wave64 __shfl_up) published per 64-base tile, and
//      per-scale Stouffer windows assembled from tile prefixes / totals
// One tile per workgroup and one base per thread in D on purpose: any loop around the
// incbet body makes the compiler hoist its ~150 fp64 coefficients into registers.
// The hardware workgroup dispatcher balances ragged tiles.
//
// LDS (doubles unless noted): table[4098] par[24] | cE+[nc] cE-[nc] P+[nc] P-[nc] W+[nc] W-[nc]
//   xA[nc] xB[nc] | codes u8[nc+8] (nc rounded up to 64);  expected counts overwrite the counts
//   (dead after B), z / non-finite prefix arrays overwrite the window sums (dead after C); xA/xB
//   and the dead count arrays are the scratch of the fast smoothing path.
// ===========================================================================
struct scan_args {
    int64_t n_intervals;
    int32_t interval_len;        // uniform mode when interval_off == nullptr
    const int64_t *interval_off; // ragged: output offsets
    const int32_t *tile_iv;      // ragged: tile table
    const int32_t *tile_t0;
    const int32_t *tile_tl;
    int64_t tile_first;          // first tile of this launch
    int32_t tiles_per_interval;  // uniform mode
    int32_t tile_len;
    int32_t hw, shw, k_trim;
    int32_t n_scales;
    int32_t scales[FPT_MAX_SCALES];
    double scale_sqrt[FPT_MAX_SCALES];
    int32_t max_scale;
    int32_t min_scale;
    int32_t nc_max;              // LDS capacity per array (padded positions)
    int64_t total_bases;
    const double *counts_plus;
    const double *counts_minus;
    const uint8_t *seq;
    const double *table;         // kTable + 1
    const double *model;         // 24 doubles
    double *exp_out, *obs_out, *pval_out, *winp_out;
    int32_t *status_out;
    const double2 *memo;         // (p, z) per (exp, obs) pair, or nullptr = direct evaluation
    int32_t memo_exp, memo_obs;
    int32_t ablate;              // timing-only diagnostics, honoured only in -DFPT_ABLATE builds
    int32_t counts_only;         // FPT_NB_NONE: stop after the expected counts
    int32_t fast_trim;           // k_trim == 1 && shw >= 32 && nc_max <= 3*NT: tile-scan smoothing
    int32_t *redo;               // per tile: memo-only pass flags a miss, full pass redoes flagged tiles
    const int32_t *redo_list;    // second pass: the flagged tiles of the launch (relative to tile_first) ...
    int32_t *redo_cursor;        // ... [0] how many, [1] the next one to hand out (k_redo_compact fills, workgroups take)
    const int32_t *dm_ids;       // per interval: dispersion-model slot relative to `model` (or nullptr)
    // second-level (exp, obs) table of the redo pass: rows of memo2_stride entries, valid for
    // exp <= memo2_max[0] and obs <= memo2_max[1] (device values: the largest pair the first pass
    // missed, found without a host round trip), or nullptr
    const double2 *memo2;
    const int32_t *memo2_max;
    const int32_t *memo2_have;  // ... or held already (kept across calls): valid up to the larger of the two
    int32_t memo2_rows, memo2_stride;
};
#ifdef FPT_ABLATE
#define ABL(bit) (a.ablate & (bit))
#else
#define ABL(bit) 0
#endif

// Direct evaluation of (p, z) for one base (dispersion.pyx:311-314 + the z of windowing.h:61).
// Build note: this library is compiled with `-mllvm -disable-machine-licm`; with machine LICM
// on, any loop around this body makes the compiler hoist the ~150 fp64 polynomial
// coefficients of incbet / ndtri into registers (70 -> 170+ VGPRs, or spills).
__device__ __forceinline__ double2 nb_pz_direct(double r, double mu, int32_t k) {
    const double pv = fptm::nb_cdf(k, r / (r + mu), r);
    const double z = fptm::ndtri(1.0 - pv);
    return make_double2(pv, z);
}

// z of a table row whose dispersion fit divides by zero (dispersion.pyx:160-161): a quiet NaN
// with a payload, so that a lookup learns about the ZeroDivisionError from the entry itself
// instead of re-evaluating the five-segment fit per base (p of such a row is NaN as before)
constexpr long long kZeroDivZBits = 0x7ff800005a440000ll;

// (p, z) for every integer pair (exp, obs) of the table: the same device functions the
// direct path calls, so a lookup returns bit-identical values.
// It is the first kernel of a scan call, so it also resets the call's device state (fewer stream
// operations per call: launch gaps are what a small batch costs): the per-tile redo flags, the
// largest missed pair (-1, -1) and the cursors of the second pass.
__global__ void __launch_bounds__(256, 4) k_nb_memo(const double *__restrict__ models, int memo_exp,
                                                    int memo_obs, double2 *__restrict__ memos,
                                                    int32_t *__restrict__ clear, int64_t n_clear,
                                                    int32_t *__restrict__ state, int32_t *__restrict__ have,
                                                    int have_state, int rows, int stride) {
    if (blockIdx.y == 0) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_clear; i += (int64_t)gridDim.x * blockDim.x)
            clear[i] = 0;
        if (state && blockIdx.x == 0 && threadIdx.x < 8) {
            // bounds of the kept second-level table: what the last call's k_nb_memo2 filled it up to
            // (that call's largest missed pair, still in state[0..1]) becomes part of them
            if (have && threadIdx.x < 2) {
                const int cap = (threadIdx.x == 0 ? rows : stride) - 1;
                const int h = have[threadIdx.x];
                have[threadIdx.x] = have_state == 2 ? -1 : (have_state == 1 ? max(h, min(state[threadIdx.x], cap)) : h);
            }
            state[threadIdx.x] = threadIdx.x < 2 ? -1 : 0;
        }
    }
    // blockIdx.y = model slot: one table per dispersion model in use
    __shared__ double par[24];
    const double *model = models + (size_t)blockIdx.y * 24;
    double2 *memo = memos + (size_t)blockIdx.y * memo_exp * memo_obs;
    if (threadIdx.x < 24) par[threadIdx.x] = model[threadIdx.x];
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < memo_exp * memo_obs) {
        const double ex = (double)(idx / memo_obs);
        const int32_t k = idx % memo_obs;
        bool zd = false;
        double r = fptm::fit_r(par + 9, ex, &zd);
        double mu = fptm::fit_mu(par, ex);
        double pv = fptm::nb_cdf(k, r / (r + mu), r);
        double z = fptm::ndtri(1.0 - pv);
        if (zd) z = __longlong_as_double(kZeroDivZBits);
        memo[idx] = make_double2(pv, z);
    }
}

// Second-level table for the redo pass of memo mode: the (exp, obs) pairs up to the largest pair
// the first pass missed (maxima left on the device by that pass: no host round trip), except the
// corner the first-level table already holds.  Heavy-tailed data (hotspots with counts in the
// hundreds) would otherwise send every base of a flagged tile through the direct incbet.
__global__ void __launch_bounds__(256, 4) k_nb_memo2(const double *__restrict__ models, const int32_t *__restrict__ mx,
                                                     const int32_t *__restrict__ have, int memo_exp, int memo_obs,
                                                     int rows, int stride, double2 *__restrict__ memos) {
    __shared__ double par[24];
    const double *model = models + (size_t)blockIdx.y * 24;
    double2 *memo = memos + (size_t)blockIdx.y * rows * stride;
    if (threadIdx.x < 24) par[threadIdx.x] = model[threadIdx.x];
    __syncthreads();
    // the table is kept across calls (fpt_capi.cpp): filled up to `have` already; this call reaches
    // max(have, missed) and computes what lies between
    const int he = have[0] + 1, hk = have[1] + 1;
    if (mx[0] < he && mx[1] < hk) return;  // nothing missed beyond what is there
    const int ne = min(max(mx[0] + 1, he), rows), nk = min(max(mx[1] + 1, hk), stride);
    if (ne <= 0 || nk <= 0) return;
    // one entry per thread, then the next block row: no loop around the incbet body (see k_nb_values)
    const long long n = (long long)ne * nk;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int ei = (int)(idx / nk), k = (int)(idx % nk);
        if (ei < memo_exp && k < memo_obs) continue;  // first-level table
        if (ei < he && k < hk) continue;              // kept from an earlier call
        bool zd = false;
        const double ex = (double)ei;
        double r = fptm::fit_r(par + 9, ex, &zd);
        double mu = fptm::fit_mu(par, ex);
        double pv = fptm::nb_cdf(k, r / (r + mu), r);
        double z = fptm::ndtri(1.0 - pv);
        if (zd) z = __longlong_as_double(kZeroDivZBits);
        memo[(size_t)ei * stride + k] = make_double2(pv, z);
    }
}

// HWC / SHWC: compile-time half window widths (0 = take them from the arguments); the
// `detect` defaults 5 / 50 get their own instance so the window loops unroll.
// TBLG: the 4097-entry bias table is gathered through the L1/L2 caches (it is 32 KB, read by
// every workgroup, and stays cache resident) instead of being copied into LDS by every
// workgroup.  Measured on config 2: 2.07 ms vs 2.75 ms with the LDS copy, because the copy
// costs 33 KB of LDS (2 instead of 3 workgroups per CU) and ~0.4 ms of staging.
// Fast smoothing + expected counts of one tile (phase C of k_scan_fused) for k == 1 and
// w = 2*shw+1 >= 65 (the `detect` default is w = 101): the trimmed sum of a window is
// S - min - max unless the 2nd smallest equals the 2nd largest element (smoothing.h:61-69 then
// applies only one weight).  Every 64-lane wavefront owns an aligned 64-position tile, scans it
// on the DPP path (prefix sums, prefix and suffix minima / maxima) and publishes the result in
// LDS; a window is then the suffix of its first tile + whole middle tiles + the prefix of its
// last tile, so no carry has to cross wavefronts.  Near-constant windows (at most 4 value
// changes between neighbours) are the only ones that can hit the equal-order-statistics rule and
// are re-done element by element, so every case keeps the reference's value.
// T = int when every window sum of the tile is an integer of magnitude <= 2^24 (cut counts
// are): the same scans on int32 are exact and cost a third of the instructions; T = double
// otherwise.  Each thread owns padded positions v = tid + i*NT.
// threadIdx.x through an empty asm: inside a loop over tiles the optimiser would otherwise hoist
// every piece of lane-index arithmetic (window bounds, LDS addresses...) out of the loop and
// keep ~100 registers of it alive across the whole tile body.
__device__ __forceinline__ int opaque_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// What phase C needs to send a window through the reference's own order of operations
// (fpt_device.hpp: trimmed_sum_rounds).  w_max = the largest |2*hw window sum| staged by the
// workgroup, or 0 when the window sums are small non-negative integers: every fast form then holds
// the exact trimmed sum T and the reference's value is T(1 + 1.2e-14) at worst.  Otherwise the
// scans and the reference both carry summation noise, bounded through w_max: at most
// (w + 256) additions of magnitude <= 256 * w_max each side.
struct tie_ctx {
    double *scratch;      // LDS free in phase C once the scans have been read (xA, xB)
    int scratch_doubles;
    int *counter;         // LDS word: slot hand-out
    double w_max;
    bool flag_only;       // the memo-only instance: flag the tile for the full instance instead
    __device__ __forceinline__ int slots(int w) const { return scratch_doubles / w; }
    // bound on |x_reference - x| for x = pq * T / div
    __device__ __forceinline__ double tol_x(double pq, double x, int w, double div) const {
        const double tol_t = w_max * ((double)((w + 256) * 256) * 1.1102230246251565e-16);
        return fabs(pq) * tol_t / div + fabs(x) * 1e-13;
    }
};

template <int NT, typename T>
__device__ __forceinline__ void smooth_expected_fast(const double *wP, const double *wM, const double *pP,
                                                     const double *pM, double *cP, double *cM, double *xA,
                                                     double *xB, int nc, int ncr, int nc_max, int pad,
                                                     int hw, int shw, bool skip_trim, int tid_in,
                                                     const tie_ctx &tc) {
    constexpr int MAXI = 3;
    const int tid = tid_in;
    const int lane = tid & (kWave - 1);
    const int w = 2 * shw + 1;
    const double w_div = (double)(w - 2), w_rdiv = 1.0 / w_div;
    const int ni = (ncr + NT - 1) / NT;  // <= MAXI (checked on the host)
    T *b0 = reinterpret_cast<T *>(cP), *b1 = reinterpret_cast<T *>(cM);
    T *b2 = reinterpret_cast<T *>(xA), *b3 = reinterpret_cast<T *>(xB);
    int *chP = reinterpret_cast<int *>(xA);
    int *chM = chP + nc_max;
    double winS[2][MAXI];
    int winC[2][MAXI];
    // C1: tile prefix sums of W and of the neighbour-change flags, both strands
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int v = i * NT + tid;
        if (i < ni && v < ncr) {  // uniform per wavefront
            T v0 = 0, v1 = 0;
            int c0 = 0, c1 = 0;
            if (v < nc) {
                const double d0 = wP[v], d1 = wM[v];
                v0 = (T)d0;
                v1 = (T)d1;
                if (v + 1 < nc) {
                    c0 = wP[v + 1] != d0;
                    c1 = wM[v + 1] != d1;
                }
            }
            b0[v] = scan_add(v0);
            b1[v] = scan_add(v1);
            chP[v] = scan_add(c0);
            chM[v] = scan_add(c1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int v = i * NT + tid;
        winS[0][i] = winS[1][i] = 0.0;
        winC[0][i] = winC[1][i] = 0;
        if (i < ni && v >= pad && v < nc - pad) {
            const int lo = v - shw, hi = v + shw;
            winS[0][i] = (double)tile_range_sum(b0, lo, hi);
            winS[1][i] = (double)tile_range_sum(b1, lo, hi);
            winC[0][i] = tile_range_sum(chP, lo, hi - 1);
            winC[1][i] = tile_range_sum(chM, lo, hi - 1);
        }
    }
    __syncthreads();
    // C2: per strand, tile prefix / suffix extrema, then the expected counts
    double eOut[2][MAXI];
    unsigned near = 0;  // bit strand * MAXI + i: that window goes through the reference's order
#pragma unroll
    for (int strand = 0; strand < 2; ++strand) {
        const double *ws = strand ? wM : wP;
        const double *ps = strand ? pM : pP;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int v = i * NT + tid;
            if (i < ni && v < ncr) {
                // forward order for the prefixes, reversed order within the tile for the
                // suffixes (a suffix scan is a prefix scan of the mirrored tile)
                const int vr = (v & ~(kWave - 1)) + (kWave - 1 - lane);
                const bool okf = v < nc, okr = vr < nc;
                const T xf = okf ? (T)ws[v] : (T)0;
                const T xr = okr ? (T)ws[vr] : (T)0;
                b0[v] = scan_min(okf ? xf : scan_lim<T>::hi());
                b2[v] = scan_max(okf ? xf : scan_lim<T>::lo());
                b1[vr] = scan_min(okr ? xr : scan_lim<T>::hi());
                b3[vr] = scan_max(okr ? xr : scan_lim<T>::lo());
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int v = i * NT + tid;
            eOut[strand][i] = 0.0;
            if (i < ni && v >= pad && v < nc - pad) {
#pragma clang fp contract(off)
                const int lo = v - shw, hi = v + shw;
                double t;
                const int nchg = winC[strand][i];
                if (nchg > 4) {
                    const double lo1 = (double)tile_range_min(b0, b1, lo, hi);
                    const double hi1 = (double)tile_range_max(b2, b3, lo, hi);
                    t = (winS[strand][i] - lo1) - hi1;
                } else if (nchg == 0) {
                    t = (double)(w - 1) * ws[lo];
                } else {
                    t = trimmed_sum_k1(ws + lo, w);
                }
                const double wsm = skip_trim ? ws[v] : div_invariant(t, w_div, w_rdiv);
                double q = 0.0;
                for (int j = -hw; j < hw; ++j) q += ps[v + j];
                const double pq = ps[v] / q, x = pq * wsm;
                eOut[strand][i] = round(x);
                if (!skip_trim && near_rounding_tie(x, tc.tol_x(pq, x, w, w_div))) near |= 1u << (strand * MAXI + i);
            }
        }
        __syncthreads();  // extrema buffers are reused by the other strand / overwritten by E
    }
    // windows whose rounding the reference's order of operations decides: once more, in that order
    // (xA / xB are free now; the first vote inside is a barrier)
#pragma unroll
    for (int strand = 0; strand < 2; ++strand) {
        const double *ws = strand ? wM : wP;
        const double *ps = strand ? pM : pP;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
#pragma clang fp contract(off)
            const int v = i * NT + tid;
            const bool need = !tc.flag_only && ((near >> (strand * MAXI + i)) & 1u);
            const double t = tc.flag_only ? 0.0 : trimmed_sum_rounds(need, ws + (need ? v - shw : 0), w, 1, tc.scratch, tc.slots(w), tc.counter, tid);
            if (need) {
                double q = 0.0;
                for (int j = -hw; j < hw; ++j) q += ps[v + j];
                eOut[strand][i] = round((ps[v] / q) * (t / w_div));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int v = i * NT + tid;
        if (i < ni && v >= pad && v < nc - pad) {
            cP[v] = eOut[0][i];
            cM[v] = eOut[1][i];
        }
    }
}

// Integer smoothing with ONE barrier inside: every scan of both strands (prefix sums of W,
// packed change counts, prefix / suffix minima and maxima) is published at once -- the extrema as
// uint16, which is what makes them fit next to the prefix sums in the four scratch arrays and
// limits this path to window sums <= 65535 -- and after the barrier each lane combines the two
// windows of ITS OWN output base (strand '+' at padded position pad+1+t, strand '-' at pad+t,
// detect.py:121-122), so the expected counts never go through LDS.
// Returns E+ and E- of output base tid (tid < nt).
template <int NT>
__device__ __forceinline__ void smooth_expected_fused(const double *wP, const double *wM, const double *pP,
                                                      const double *pM, double *cP, double *cM, double *xA,
                                                      double *xB, int nc, int ncr, int nc_max, int nt, int pad,
                                                      int hw, int shw, bool skip_trim, double &e_plus,
                                                      double &e_minus, int tid_in, const tie_ctx &tc,
                                                      bool &flag_tile) {
    typedef unsigned short u16;
    const int tid = tid_in;
    const int lane = tid & (kWave - 1);
    const int w = 2 * shw + 1;
    const double w_div = (double)(w - 2), w_rdiv = 1.0 / w_div;
    int *psP = reinterpret_cast<int *>(cP), *psM = psP + nc_max;
    int *chB = reinterpret_cast<int *>(cM);
    u16 *mnP = reinterpret_cast<u16 *>(chB + nc_max), *mnPs = mnP + nc_max;   // prefix / suffix min '+'
    u16 *mxP = reinterpret_cast<u16 *>(xA), *mxPs = mxP + nc_max;             // prefix / suffix max '+'
    u16 *mnM = mxPs + nc_max, *mnMs = mnM + nc_max;
    u16 *mxM = reinterpret_cast<u16 *>(xB), *mxMs = mxM + nc_max;
    for (int v = tid; v < ncr; v += NT) {  // whole 64-position tiles: wave-uniform
        const int vr = (v & ~(kWave - 1)) + (kWave - 1 - lane);
        const bool okf = v < nc, okr = vr < nc;
        int f0 = 0, f1 = 0, c0 = 0, c1 = 0;
        if (okf) {
            const double d0 = wP[v], d1 = wM[v];
            f0 = (int)d0;
            f1 = (int)d1;
            if (v + 1 < nc) {
                c0 = wP[v + 1] != d0;
                c1 = wM[v + 1] != d1;
            }
        }
        const int r0 = okr ? (int)wP[vr] : 0, r1 = okr ? (int)wM[vr] : 0;
        psP[v] = scan_add(f0);
        psM[v] = scan_add(f1);
        chB[v] = scan_add(c0 | (c1 << 16));
        mnP[v] = (u16)scan_min_nonneg(f0, okf);
        mxP[v] = (u16)scan_max_nonneg(f0, okf);
        mnM[v] = (u16)scan_min_nonneg(f1, okf);
        mxM[v] = (u16)scan_max_nonneg(f1, okf);
        // positions beyond the data scan as "no value": min = kNonnegTop there, never read back
        mnPs[vr] = (u16)scan_min_nonneg(r0, okr);
        mxPs[vr] = (u16)scan_max_nonneg(r0, okr);
        mnMs[vr] = (u16)scan_min_nonneg(r1, okr);
        mxMs[vr] = (u16)scan_max_nonneg(r1, okr);
    }
    __syncthreads();
    e_plus = e_minus = 0.0;
    flag_tile = false;
    const bool le3 = shw <= 64;
    auto range_min = [&](const u16 *p, const u16 *s, int lo, int hi) {  // spans >= 2 tiles (w > 64)
        int m = min((int)s[lo], (int)p[hi]);
        for (int q = (lo >> 6) + 1; q < (hi >> 6); ++q) m = min(m, (int)p[(q << 6) + 63]);
        return m;
    };
    auto range_max = [&](const u16 *p, const u16 *s, int lo, int hi) {
        int m = max((int)s[lo], (int)p[hi]);
        for (int q = (lo >> 6) + 1; q < (hi >> 6); ++q) m = max(m, (int)p[(q << 6) + 63]);
        return m;
    };
    bool near[2] = {false, false};
    if (tid < nt) {
#pragma unroll
        for (int strand = 0; strand < 2; ++strand) {
#pragma clang fp contract(off)
            const int v = pad + tid + (strand ? 0 : 1);
            const double *ws = strand ? wM : wP, *pr = strand ? pM : pP;
            const int lo = v - shw, hi = v + shw;
            const int S = le3 ? tile_range_sum3(strand ? psM : psP, lo, hi) : tile_range_sum(strand ? psM : psP, lo, hi);
            const int wc = le3 ? tile_range_sum3(chB, lo, hi - 1) : tile_range_sum(chB, lo, hi - 1);
            const int nchg = strand ? (wc >> 16) : (wc & 0xffff);
            double t;
            if (nchg > 4) {
                const int mn = range_min(strand ? mnM : mnP, strand ? mnMs : mnPs, lo, hi);
                const int mx = range_max(strand ? mxM : mxP, strand ? mxMs : mxPs, lo, hi);
                t = ((double)S - (double)mn) - (double)mx;
            } else if (nchg == 0) {
                t = (double)(w - 1) * ws[lo];
            } else {
                t = trimmed_sum_k1(ws + lo, w);
            }
            const double wsm = skip_trim ? ws[v] : div_invariant(t, w_div, w_rdiv);
            double q = 0.0;
            for (int j = -hw; j < hw; ++j) q += pr[v + j];
            const double pq = pr[v] / q, x = pq * wsm;
            const double e = round(x);
            near[strand] = !skip_trim && near_rounding_tie(x, tc.tol_x(pq, x, w, w_div));
            if (strand) e_minus = e; else e_plus = e;
        }
    }
    if (tc.flag_only) {  // the full instance, launched behind this one, settles it
        flag_tile = near[0] | near[1];
        return;
    }
    // t is the exact integer; a window whose rounding the reference's noise around it could decide
    // is evaluated once more the reference's way (the scan buffers in xA / xB are free after the
    // first vote inside)
#pragma unroll
    for (int strand = 0; strand < 2; ++strand) {
#pragma clang fp contract(off)
        const int v = pad + tid + (strand ? 0 : 1);
        const double *ws = strand ? wM : wP, *pr = strand ? pM : pP;
        const bool need = near[strand];
        const double t = trimmed_sum_rounds(need, ws + (need ? v - shw : 0), w, 1, tc.scratch, tc.slots(w), tc.counter, tid);
        if (need) {
            double q = 0.0;
            for (int j = -hw; j < hw; ++j) q += pr[v + j];
            const double e = round((pr[v] / q) * (t / w_div));
            if (strand) e_minus = e; else e_plus = e;
        }
    }
}

// MO ("memo only"): the instance used first in memo mode.  It has no direct incbet/ndtri body,
// so it needs 56 instead of 76 VGPRs and 8 wavefronts per SIMD fit; the model parameters are
// read with scalar loads and the sequence codes share LDS with a scratch array, so a 500-base
// tile takes exactly 40 KB (4 workgroups per CU) and a 1 kb tile 72 KB (2 per CU).  A tile in
// which some base misses the table (pair outside it, non-integer or non-finite exp) is flagged
// in `redo` and computed again by the full instance, launched right behind on the same stream
// over the same tiles with an early exit for unflagged ones.
#define FPT_SCAN_WAVES(NT, TBLG, MO) ((MO) ? 8 : (((TBLG) && (NT) < 1024) ? 6 : 4))
typedef const __attribute__((address_space(4))) scan_args kernarg_scan_args;
// LOOPED: the caller runs the body inside a loop over tiles (see opaque_tid)
template <int NT, int HWC, int SHWC, bool TBLG, bool MO, bool LOOPED, typename Args>
__device__ __forceinline__ void scan_tile(Args &a, const int64_t tile) {
    extern __shared__ double smem[];
    const double *tbl = TBLG ? a.table : smem;  // kTable + 1 (+1 pad to keep 16-B alignment)
    double *par_lds = smem + (TBLG ? 0 : (kTable + 2));  // 24 + 2 (unused when MO)
    double *aux = par_lds + 24;               // [0] largest |window sum| of the tile, [1] slot counter (tie_ctx)
    double *cP = par_lds + (MO ? 0 : 26);     // counts '+', scratch in C, expected '+' for D
    double *cM = cP + a.nc_max;
    double *pP = cM + a.nc_max;               // propensities
    double *pM = pP + a.nc_max;
    double *wP = pM + a.nc_max;               // window sums, later z prefix
    double *wM = wP + a.nc_max;               // window sums, later non-finite prefix (int)
    double *xA = wM + a.nc_max;               // scratch of the fast smoothing path
    double *xB = xA + a.nc_max;
    // sequence codes: nc_max + 8 bytes; with MO they share xB (first written in phase C)
    uint8_t *sq = reinterpret_cast<uint8_t *>(MO ? xB : xB + a.nc_max);

    const int tid = LOOPED ? opaque_tid() : (int)threadIdx.x;
    const int lane = tid & (kWave - 1);

    const int hw = HWC ? HWC : a.hw, shw = SHWC ? SHWC : a.shw, pad = hw + shw;
    const int H = a.max_scale;

    int64_t iv;
    int t0, L, tl;
    int64_t out_off;
    if (a.interval_off) {
        iv = a.tile_iv[tile];
        t0 = a.tile_t0[tile];
        tl = a.tile_tl[tile];
        out_off = a.interval_off[iv];
        L = (int)(a.interval_off[iv + 1] - out_off);
    } else {
        if (a.tiles_per_interval == 1) {  // the common case; a 64-bit scalar division costs ~150 instructions
            iv = tile;
            t0 = 0;
        } else if ((tile >> 32) == 0) {
            const uint32_t q = (uint32_t)tile / (uint32_t)a.tiles_per_interval;
            iv = q;
            t0 = (int)((uint32_t)tile - q * (uint32_t)a.tiles_per_interval) * a.tile_len;
        } else {
            iv = tile / a.tiles_per_interval;
            t0 = (int)(tile % a.tiles_per_interval) * a.tile_len;
        }
        L = a.interval_len;
        out_off = iv * (int64_t)L;
        tl = min(a.tile_len, L - t0);
    }
    const int ta = max(0, t0 - H);
    const int tb = min(L, t0 + tl + H);
    const int nt = tb - ta;              // positions needing p / z   (<= NT by construction)
    const int nc = nt + 2 * pad + 1;     // padded positions staged   (<= nc_max)
    const int ncr = (nc + kWave - 1) & ~(kWave - 1);  // rounded up to whole 64-position tiles
    const int64_t cbase = out_off + iv * (int64_t)(2 * pad + 1) + ta;
    const int64_t sbase = out_off + iv * (int64_t)(2 * pad + 7) + ta;

    // ---- A: stage table, model, counts and sequence codes
    // (raised wave priority until the inputs are staged, as in k_scan_lean: a new tile's loads go out ahead of the
    // arithmetic of the older wavefronts of the SIMD)
    __builtin_amdgcn_s_setprio(3);
    if (!TBLG)
        for (int i = tid; i <= kTable; i += NT) smem[i] = a.table[i];
    // dispersion model of this interval (dm_ids: one slot index per interval, else slot 0 of
    // the launch): parameters staged in LDS (scalar loads from the table when MO), memo table
    // of that model
    const int dm = a.dm_ids ? a.dm_ids[iv] : 0;
    const double *model = a.model + (size_t)dm * 24;
    const double2 *memo = a.memo ? a.memo + (size_t)dm * a.memo_exp * a.memo_obs : nullptr;
    const double *par = MO ? model : par_lds;
    if (!MO && tid < 24) par_lds[tid] = model[tid];
    if (ABL(16384)) {  // inputs made up in registers: what the kernel costs without its HBM reads
        for (int v = tid; v < nc; v += NT) {
            cP[v] = (double)((v * 7 + (int)tile) % 20);
            cM[v] = (double)((v * 13 + (int)tile) % 20);
        }
        for (int v = tid; v < nc + 6; v += NT) sq[v] = (uint8_t)((v * 5 + (int)tile) & 3);
    } else {
        // all loads of a lane are issued before the first one is consumed: a lane that owns two or
        // three staged positions pays the HBM latency once, not once per position
        constexpr int kStage = 3;
        double ldp[kStage], ldm[kStage];
        uint8_t lds_[kStage];
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int v = tid + i * NT;
            if (v < nc) {
                ldp[i] = a.counts_plus[cbase + v];
                ldm[i] = a.counts_minus[cbase + v];
            }
            if (v < nc + 6) lds_[i] = a.seq[sbase + v];
        }
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int v = tid + i * NT;
            if (v < nc) {
                cP[v] = ldp[i];
                cM[v] = ldm[i];
            }
            if (v < nc + 6) sq[v] = (uint8_t)base_code(lds_[i]);
        }
        for (int v = tid + kStage * NT; v < nc; v += NT) {  // very wide padding only
            cP[v] = a.counts_plus[cbase + v];
            cM[v] = a.counts_minus[cbase + v];
        }
        for (int v = tid + kStage * NT; v < nc + 6; v += NT) sq[v] = (uint8_t)base_code(a.seq[sbase + v]);
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (ABL(32)) return;

    // ---- B: bias lookup (bias.py:101-111), count window sums (predict.h:41-48),
    //         strand merge of the observed counts (detect.py:121; kept in a register: the
    //         thread that owns padded position v = tid also owns output base t' = tid)
    double ob = 0.0;
    if (tid < nt) ob = cP[pad + 1 + tid] + cM[pad + tid];
    int small_int = 1;
    const double int_lim = 65535.0;  // what smooth_expected_fused stores its extrema in
    // The two table gathers of every staged position of a lane are issued first and stored last,
    // so the window sums run while they are in flight (a lane owns up to kStageB positions in
    // the common geometries; more only with very wide padding).
    constexpr int kStageB = 3;
    double tfs[kStageB], trs[kStageB];
#pragma unroll
    for (int i = 0; i < kStageB; ++i) {
        const int v = tid + i * NT;
        if (v < nc) {
            int fi, ri;
            if (ABL(8)) {
                fi = v & 4095;
                ri = (v * 7) & 4095;
            } else {
                kmer_indices(sq + v, fi, ri);
            }
            tfs[i] = tbl[fi];
            trs[i] = tbl[ri];
        }
    }
#pragma unroll
    for (int i = 0; i < kStageB; ++i) {
        const int v = tid + i * NT;
        if (v >= nc) break;
        double sp = 0.0, sm = 0.0;
        if (v >= hw && v < nc - hw) {
            for (int j = -hw; j < hw; ++j) {
                sp += cP[v + j];
                sm += cM[v + j];
            }
        }
        wP[v] = sp;
        wM[v] = sm;
        small_int &= (int)(sp >= 0.0) & (int)(sm >= 0.0) & (int)(sp <= int_lim) & (int)(sm <= int_lim) &
                     (int)((double)(int)sp == sp) & (int)((double)(int)sm == sm);
        pP[v] = tfs[i];
        pM[v] = trs[i];
    }
    for (int v = tid + kStageB * NT; v < nc; v += NT) {
        int fi, ri;
        kmer_indices(sq + v, fi, ri);
        const double tf = tbl[fi], tr = tbl[ri];
        double sp = 0.0, sm = 0.0;
        if (v >= hw && v < nc - hw) {
            for (int j = -hw; j < hw; ++j) {
                sp += cP[v + j];
                sm += cM[v + j];
            }
        }
        wP[v] = sp;
        wM[v] = sm;
        // small-integer test for the int32 smoothing scans: integral and small enough that a
        // window of w values (and a 64-position tile prefix) stays below 2^30
        small_int &= (int)(sp >= 0.0) & (int)(sm >= 0.0) & (int)(sp <= int_lim) & (int)(sm <= int_lim) &
                     (int)((double)(int)sp == sp) & (int)((double)(int)sm == sm);
        pP[v] = tf;
        pM[v] = tr;
    }
    const bool all_small_int = __syncthreads_and(small_int) != 0;
    if (ABL(64)) return;
    // what a window needs to go through the reference's own order of operations where that order
    // decides round() (tie_ctx).  The memo-only instance hands such tiles to the full one.
    tie_ctx tc;
    tc.scratch = xA;
    tc.scratch_doubles = 2 * a.nc_max;
    tc.counter = reinterpret_cast<int *>(aux + 1);
    tc.w_max = 0.0;
    tc.flag_only = MO;
    if (MO) {
        if (!all_small_int && tid == 0) a.redo[tile] = 1;
    } else if (!all_small_int) {
        // fractional, negative or very large counts: the tolerance needs the largest |window sum|
        // (non-negative doubles order like their bit patterns; a NaN ends up on top and switches
        // the re-evaluation off -- the reference's selection is undefined on NaN)
        unsigned long long *wmax_bits = reinterpret_cast<unsigned long long *>(aux);
        if (tid == 0) *wmax_bits = 0ull;
        __syncthreads();
        double mx = 0.0;
        for (int v = tid; v < nc; v += NT) mx = fmax(mx, fmax(fabs(wP[v]), fabs(wM[v])));
        for (int v = tid; v < nc; v += NT)
            if (wP[v] != wP[v] || wM[v] != wM[v]) mx = __longlong_as_double(0x7ff8000000000000ll);
        atomicMax(wmax_bits, (unsigned long long)__double_as_longlong(mx));
        __syncthreads();
        tc.w_max = __longlong_as_double((long long)*wmax_bits);
    }

    // ---- C: smoothing (smoothing.h:107-133) + expected counts (predict.h:60-63);
    //         E overwrites the counts, which nobody reads any more
    double ex_plus = 0.0, ex_minus = 0.0;  // the integer path hands E of the lane's base over in registers
    if (a.fast_trim) {
        // T = int when the whole tile's window sums are small integers (decided block-wide at the
        // barrier that ends phase B), else double
        if (all_small_int) {
            bool flag_tile = false;
            smooth_expected_fused<NT>(wP, wM, pP, pM, cP, cM, xA, xB, nc, ncr, a.nc_max, nt, pad, hw, shw, ABL(1),
                                      ex_plus, ex_minus, tid, tc, flag_tile);
            if (MO && flag_tile) a.redo[tile] = 1;
        } else {
            smooth_expected_fast<NT, double>(wP, wM, pP, pM, cP, cM, xA, xB, nc, ncr, a.nc_max, pad, hw, shw, ABL(1), tid, tc);
        }
    } else {
        const int ne = nt + 1;  // padded positions [pad, nc-pad) per strand
        const int w = 2 * shw + 1;
        const bool smooth = shw > 0 && !ABL(1);
        const double w_div = (double)(w - 2 * a.k_trim);
        for (int first = 0; first < 2 * ne; first += NT) {  // the same trip count for every lane: votes inside
#pragma clang fp contract(off)
            const int idx = first + tid;
            const bool live = idx < 2 * ne;
            const bool minus = idx >= ne;
            const int v = live ? pad + (minus ? idx - ne : idx) : pad;
            const double *ws = minus ? wM : wP;
            const double *ps = minus ? pM : pP;
            double q = 0.0, pq = 0.0, e = 0.0;
            bool need = false;
            if (live) {
                const double wsm = smooth ? trimmed_mean(ws + v - shw, w, a.k_trim) : ws[v];
                for (int j = -hw; j < hw; ++j) q += ps[v + j];
                pq = ps[v] / q;
                const double x = pq * wsm;
                e = round(x);
                need = smooth && near_rounding_tie(x, tc.tol_x(pq, x, w, w_div));
            }
            if (MO) {
                if (need) a.redo[tile] = 1;
            } else if (smooth) {
                const double t = trimmed_sum_rounds(need, ws + v - shw, w, a.k_trim, tc.scratch, tc.slots(w), tc.counter, tid);
                if (need) e = round(pq * (t / w_div));
            }
            if (live) (minus ? cM : cP)[v] = e;
        }
    }
    __syncthreads();
    if (ABL(128)) return;

    // ---- D: expected merge (detect.py:122), p-value (dispersion.pyx:311-314), z = ndtri(1-p)
    double *zb = wP;  // window sums are dead now
    double zv = 0.0;
    int zc = 0;
    // expected counts: the integer smoothing path keeps them in registers, the others in cP / cM
    const bool e_in_reg = a.fast_trim && all_small_int;
    const double *eP = cP, *eM = cM;
    double *fA = xA, *fB = xB;  // scratch for the direct mode's lane regrouping
    double ex = 0.0, pv = 0.0, z = 0.0;
    bool zd = false;
    if (tid < nt) ex = e_in_reg ? ex_plus + ex_minus : eP[pad + 1 + tid] + eM[pad + tid];
    if (!MO && a.counts_only) {  // expected / observed tracks only (learn_dm)
        const int t = ta + tid;
        if (tid < nt && t >= t0 && t < t0 + tl) {
            if (a.exp_out) a.exp_out[out_off + t] = ex;
            if (a.obs_out) a.obs_out[out_off + t] = ob;
        }
        return;
    }
    const int32_t k = fptm::c_int(ob);
    if (!MO && !memo && !ABL(2)) {
        // Direct mode: every base evaluates incbet itself.  Lanes are regrouped first so that a
        // wavefront mostly runs ONE of incbet's expansions (power series / continued fraction 1 /
        // continued fraction 2) on similar observed counts: a counting sort of the tile's bases
        // by (expansion, obs/4) through LDS, evaluation in sorted order, results handed back
        // through LDS.  Values are unchanged; only who computes them moves.
        int *bins = reinterpret_cast<int *>(sq);       // 64 counters, then bases (sq is dead)
        int *order = reinterpret_cast<int *>(pP);      // propensities are dead after C
        double *s_r = pM, *s_mu = fB;
        int *s_k = reinterpret_cast<int *>(fA);
        if (tid < 64) bins[tid] = 0;
        __syncthreads();
        int key = 0;
        if (tid < nt) {
            const double r = fptm::fit_r(par + 9, ex, &zd);
            const double mu = fptm::fit_mu(par, ex);
            const double xx = r / (r + mu), bb = (double)fptm::wrap_inc(k);
            int cls = 0;  // trivial / domain
            if (r > 0.0 && bb > 0.0 && xx > 0.0 && xx < 1.0) {
                const bool direct = (bb * xx) <= 1.0 && xx <= 0.95;
                const bool flipped = !direct && xx > (r / (r + bb));
                const double ia = flipped ? bb : r, ib = flipped ? r : bb, ix = flipped ? 1.0 - xx : xx;
                const bool series = direct || (flipped && (ib * ix) <= 1.0 && ix <= 0.95);
                cls = series ? 1 : ((ix * (ia + ib - 2.0) - (ia - 1.0)) < 0.0 ? 2 : 3);
            }
            key = cls * 16 + min(max(k, 0) >> 2, 15);
            s_r[tid] = r;
            s_mu[tid] = mu;
            s_k[tid] = k;
            atomicAdd(&bins[key], 1);
        }
        __syncthreads();
        if (tid < kWave) {  // exclusive prefix of the 64 bins
            const int c = bins[tid];
            const int incl = wave_scan_i32(c);
            bins[tid] = incl - c;
        }
        __syncthreads();
        if (tid < nt) order[atomicAdd(&bins[key], 1)] = tid;
        __syncthreads();
        if (tid < nt) {
            const int src = order[tid];
            const double2 pz = nb_pz_direct(s_r[src], s_mu[src], s_k[src]);
            s_r[src] = pz.x;
            s_mu[src] = pz.y;
        }
        __syncthreads();
        if (tid < nt) {
            pv = s_r[tid];
            z = s_mu[tid];
        }
    } else if (tid < nt) {
        const int ei = (int)ex;
        if (ABL(2)) {
            pv = 0.5;
            z = ob * 0.01;
        } else if (memo && ex >= 0.0 && ex < (double)a.memo_exp && (double)ei == ex && k >= 0 &&
                   k < a.memo_obs) {
            const double2 pz = memo[ei * a.memo_obs + k];
            pv = pz.x;
            z = pz.y;
            zd = __double_as_longlong(z) == kZeroDivZBits;
        } else if (MO) {
            pv = z = NAN;
            a.redo[tile] = 1;  // the full instance recomputes this tile
        } else if (a.memo2 && ex >= 0.0 && ex < (double)a.memo2_rows && (double)ei == ex && k >= 0 &&
                   k < a.memo2_stride && ei <= max(a.memo2_max[0], a.memo2_have[0]) &&
                   k <= max(a.memo2_max[1], a.memo2_have[1])) {
            // second-level table, built between the passes for the pairs the first pass missed
            const double2 pz = a.memo2[((size_t)dm * a.memo2_rows + ei) * a.memo2_stride + k];
            pv = pz.x;
            z = pz.y;
            zd = __double_as_longlong(z) == kZeroDivZBits;
        } else {
            const double r = fptm::fit_r(par + 9, ex, &zd);
            const double mu = fptm::fit_mu(par, ex);
            const double2 pz = nb_pz_direct(r, mu, k);
            pv = pz.x;
            z = pz.y;
        }
    }
    if (tid < nt) {
        const int tp = tid;
        bool fin = isfinite(z);
        zv = fin ? z : 0.0;
        zc = fin ? 0 : 1;
        int t = ta + tp;
        if (t >= t0 && t < t0 + tl) {
            int64_t g = out_off + t;
            if (a.exp_out) a.exp_out[g] = ex;
            if (a.obs_out) a.obs_out[g] = ob;
            if (a.pval_out) a.pval_out[g] = pv;
        }
        if (zd && a.status_out) atomicOr(&a.status_out[iv], 1);
    }
    if (a.n_scales == 0 || ABL(256)) return;

    // ---- E: Stouffer windows (windowing.h:53-84): out = ndtr(-(sum of z over 2*hs+1 bases) / sqrt(2*hs+1)),
    //         NaN when the window holds a non-finite z, 1.0 within hs of the interval's ends.
    constexpr int kDirectWin = 8;
    double *zraw = pP;  // propensities are dead after C
    if (a.n_scales == 1 && a.max_scale <= kDirectWin) {
        // One narrow scale (the reference's only one is hw = 3): summed directly from the raw z in
        // LDS, left to right like the reference; a non-finite z makes the sum non-finite.
        if (tid < nt) zraw[tid] = z;
        __syncthreads();
        const int hs = a.scales[0];
        const double rk = a.scale_sqrt[0];
        double *dst = a.winp_out + out_off;
        for (int v = tid; v < tl; v += NT) {
            const int t = t0 + v;
            double res = 1.0;  // edges are 1.0 (windowing.pyx:51)
            if (t >= hs && t < L - hs) {
                double sv = 0.0;
                for (int j = t - ta - hs; j <= t - ta + hs; ++j) sv += zraw[j];
                res = !isfinite(sv) ? NAN : (ABL(4) ? sv : fptm::ndtr_window(-(sv * rk)));
            }
            dst[t] = res;
        }
        return;
    }
    // Several scales (or a wide one): every window sum is a difference of two entries of ONE
    // workgroup-wide prefix sum of z, built in two levels -- rows of 16 lanes on the DPP path, the
    // NT/16 row totals scanned by the first wavefront -- so a scale costs two LDS reads and a
    // subtraction whatever its width.  Tiles holding a non-finite z (p = 0, p >= 1, p below 2^-53:
    // the NaN rule) are rare and take the tile-prefix path below, which counts them per window.
    constexpr int NROW = NT / 16;
    double *rowtot = pM, *rowcar = pM + NROW;  // propensities are dead after C
    {
        const double zr = row_scan_f64(zv);  // lanes beyond nt hold 0
        zb[tid] = zr;
        if ((lane & 15) == 15) rowtot[tid >> 4] = zr;
    }
    const bool any_nonfinite = __syncthreads_or(zc) != 0;
    if (!any_nonfinite) {
        if (tid < kWave) {
            const double tv = tid < NROW ? rowtot[tid] : 0.0;
            const double inc = wave_scan_f64(tv, 0.0, op_add());
            if (tid < NROW) rowcar[tid] = inc - tv;
        }
        __syncthreads();
        const int t = t0 + tid, idx = t - ta;  // tl <= nt <= NT: one output base per lane
        for (int s = 0; s < a.n_scales; ++s) {
            const int hs = a.scales[s];
            const double rk = a.scale_sqrt[s];
            double *dst = a.winp_out + (int64_t)s * a.total_bases + out_off;
            const bool inside = tid < tl && t >= hs && t < L - hs;
            const int hi = inside ? idx + hs : 0, lo = inside ? idx - hs - 1 : -1;
            const int lo_c = lo < 0 ? 0 : lo;
            const double p_hi = zb[hi] + rowcar[hi >> 4];
            const double p_lo = zb[lo_c] + rowcar[lo_c >> 4];
            const double sv = p_hi - (lo < 0 ? 0.0 : p_lo);
            const double pw = ABL(4) ? sv : fptm::ndtr_window(-(sv * rk));
            if (tid < tl) dst[t] = inside ? pw : 1.0;  // edges are 1.0 (windowing.pyx:51)
        }
        return;
    }
    // tile-prefix path: per-wavefront prefix sums of (z with non-finite values as 0, count of
    // non-finite values) published per 64-base tile; narrow windows directly from the raw z
    int *nf = reinterpret_cast<int *>(wM);
    if (a.min_scale <= kDirectWin && tid < nt) zraw[tid] = z;
    if (a.max_scale > kDirectWin) {
        wave_scan(zv, zc, lane);
        if (tid < ((nt + kWave - 1) & ~(kWave - 1))) {
            zb[tid] = zv;
            nf[tid] = zc;
        }
    }
    __syncthreads();
    for (int s = 0; s < a.n_scales; ++s) {
        const int hs = a.scales[s];
        const double rk = a.scale_sqrt[s];
        double *dst = a.winp_out + (int64_t)s * a.total_bases + out_off;
        if (hs <= kDirectWin) {
            for (int v = tid; v < tl; v += NT) {
                const int t = t0 + v;
                double res = 1.0;  // edges are 1.0 (windowing.pyx:51)
                if (t >= hs && t < L - hs) {
                    double sv = 0.0;
                    for (int j = t - ta - hs; j <= t - ta + hs; ++j) sv += zraw[j];
                    res = !isfinite(sv) ? NAN : (ABL(4) ? sv : fptm::ndtr_window(-(sv * rk)));
                }
                dst[t] = res;
            }
            continue;
        }
        for (int v = tid; v < tl; v += NT) {
            int t = t0 + v;
            double res = 1.0;  // edges are 1.0 (windowing.pyx:51)
            if (t >= hs && t < L - hs) {
                const int lo = t - ta - hs, hi = t - ta + hs;
                const double sv = hs <= 64 ? tile_range_sum3(zb, lo, hi) : tile_range_sum(zb, lo, hi);
                const int sc = hs <= 64 ? tile_range_sum3(nf, lo, hi) : tile_range_sum(nf, lo, hi);
                res = (sc > 0) ? NAN : (ABL(4) ? sv : fptm::ndtr_window(-(sv * rk)));
            }
            dst[t] = res;
        }
    }
}

// The tiles the first pass of memo mode flagged, as a list (order does not matter): one lane per
// tile, one atomic per wavefront.
__global__ void __launch_bounds__(256) k_redo_compact(const int32_t *__restrict__ flags, int64_t n,
                                                      int32_t *__restrict__ list, int32_t *__restrict__ cursor) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool f = i < n && flags[i] != 0;
    const unsigned long long m = __ballot(f);
    if (m == 0) return;
    int base = 0;
    if ((threadIdx.x & (kWave - 1)) == 0) base = atomicAdd(&cursor[0], __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (f) list[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = (int32_t)i;
}

// One tile per workgroup, except in the second pass of memo mode (REDO: the full instance over the
// tiles the memo-only pass flagged): there as many workgroups as the GPU holds at once take the
// flagged tiles off k_redo_compact's list one by one (an atomic cursor), so a pass in which
// nothing is flagged costs one short launch (it used to be one workgroup per 16 tiles: 0.31 ms
// of wavefront launches per 10^6 tiles), and a pass with hotspots is balanced dynamically.
template <int NT, int HWC, int SHWC, bool TBLG, bool MO, bool REDO>
__global__ void __launch_bounds__(NT, FPT_SCAN_WAVES(NT, TBLG, MO)) k_scan_fused(const scan_args a) {
    if constexpr (!REDO) {
        scan_tile<NT, HWC, SHWC, TBLG, MO, false>(a, a.tile_first + blockIdx.x);
    } else {
        // the arguments through a reference into the kernarg segment (constant address space:
        // scalar loads) whose pointer is re-laundered every iteration, so that the argument loads
        // are not hoisted out of the loop either
        kernarg_scan_args *ap = (kernarg_scan_args *)__builtin_amdgcn_kernarg_segment_ptr();
        __shared__ int next_tile;
        for (;;) {
            asm volatile("" : "+s"(ap));  // argument loads stay inside the iteration that needs them
            __syncthreads();              // the previous tile of this workgroup is done with LDS
            if (threadIdx.x == 0) next_tile = atomicAdd(&ap->redo_cursor[1], 1);
            __syncthreads();
            const int j = __builtin_amdgcn_readfirstlane(next_tile);
            if (j >= ((const __attribute__((address_space(4))) int32_t *)ap->redo_cursor)[0]) break;
            const int64_t tile = ap->tile_first + ((const __attribute__((address_space(4))) int32_t *)ap->redo_list)[j];
            scan_tile<NT, HWC, SHWC, TBLG, false, true>(*ap, tile);
        }
    }
}

#define FPT_SCAN_INSTANCES(X) X(256, 0, 0) X(512, 0, 0) X(1024, 0, 0) X(256, 5, 50) X(512, 5, 50) X(1024, 5, 50)
#define FPT_INST(NT, H_, S_)                                                                \
    template __global__ void k_scan_fused<NT, H_, S_, false, false, false>(const scan_args); \
    template __global__ void k_scan_fused<NT, H_, S_, true, false, false>(const scan_args);  \
    template __global__ void k_scan_fused<NT, H_, S_, true, true, false>(const scan_args);   \
    template __global__ void k_scan_fused<NT, H_, S_, true, false, true>(const scan_args);
FPT_SCAN_INSTANCES(FPT_INST)
#undef FPT_INST

// ===========================================================================
// k_fdr_null: empirical FDR of one interval per workgroup (cli/detect.py:132-135).
//   1. the interval's observed window p-values are sorted in LDS (filed and ranked, or a bitonic network; NaN
//      last) and translated into thresholds for the null windows' raw sums
//   2. `times` null tracks: per base an NB draw from the row's alias table (k_nb_alias; the direct inverse cdf
//      off the tables) whose z is read off a table, Stouffer window like phase E of the scan, and every null
//      window is ranked among the thresholds through a guide + a short search, into an LDS histogram
//   3. a prefix sum of the histogram gives #{null <= observed} for every base
// The null values are never stored: 100 draws per base stay on chip.
// With the `detect` width the steps are launches of their own (MODE below), and long intervals' draws are
// sliced (k_fdr_slice).
// ===========================================================================
__host__ __device__ inline int fdr_guide_slices(int n2) { return 4 * n2 < 4096 ? 4 * n2 : 4096; }
// the lanes of a slice of an interval of L bases: 128, 192 or 256, whichever covers the interval's positions
// (lanes - 6 outputs per slice) with the fewest lanes in all; of equals the widest (fewer halo positions)
__host__ __device__ inline int fdr_slice_lanes(int L) {
    int best = 256, cost = ((L + 249) / 250) * 256;
    const int c192 = ((L + 185) / 186) * 192, c128 = ((L + 121) / 122) * 128;
    if (c192 < cost) best = 192, cost = c192;
    if (c128 < cost) best = 128;
    return best;
}

struct fdr_args {
    int64_t n_intervals;
    int32_t interval_len;
    const int64_t *interval_off;
    int64_t base_index0;
    int32_t hw, times;
    uint64_t seed;
    const double *model;
    const double2 *memo;
    const uint32_t *alias;  // alias tables (k_nb_alias): room for 2^alias_lg slots per (model, integer exp) row of the memo
    const double *zt;       // z of outcome k of the same rows
    const uint8_t *row_lg;  // the width of every row's table
    int32_t alias_lg;
    int32_t memo_exp, memo_obs;
    const double *exp;
    const double *winp;
    const double *obs;       // optional: observed counts -> the observed windows are ranked in y, ties exact
    double *efdr;
    const double *null_uniform;
    double *null_out;       // optional [base][times] null window p-values
    const int32_t *dm_ids;  // per interval model slot relative to `model`, or nullptr
    int32_t ablate;      // timing-only diagnostics (FPT_ABLATE builds)
    int32_t n2_max;      // buffer capacity: power of two >= longest interval of the launch
    double inv_sqrt_k, sqrt_k;  // y = -(sum of z) / sqrt(K) exactly as windowing.h:64 divides (div_invariant)
    int32_t dbuf;            // a second set of z buffers follows the first: one barrier per pass instead of two
    const int32_t *iv_list;  // interval of workgroup b is iv_list[b], or iv_first + b when null
    int64_t iv_first;
    char *gws;               // GWS instances: per-workgroup buffers in global memory
    int64_t gws_stride;
    // the hand-over between the set-up launch (MODE 1) and the draw launch (MODE 2): the interval's sorted
    // thresholds and their positions (at the interval's offset in the tracks), and m / rank_one per interval
    double *ws_key;
    uint16_t *ws_idx;
    int32_t *ws_misc;        // three per interval: m, rank_one, and "left to the full draw launch" (see MODE 3)
    int64_t ws_total;        // positions the hand-over arrays have room for (the host's total)
    int32_t redo_only;       // MODE 2: only the intervals the light draw launch left
};

// Alias tables of the null sampler (Walker / Vose): for every (model, integer exp) row of the memo the
// outcomes 0 .. n-2 (their probabilities the differences of the row's cdf) and "n-1 or more" (the
// rest) are spread over n = 2^lg slots of probability 1/n each; a slot holds one outcome up to a
// threshold and a second one (its alias) above it.  A draw is then one 4-byte gather -- slot from
// the top lg bits of the Philox word, threshold against the other 32 - lg -- and one 8-byte gather
// of the outcome's z, with no search and no walk.
// n is chosen PER ROW: the smallest power of two whose rest is at most 2^-32 (capped by the memo's
// width), so that a row of small counts is a few hundred bytes and the tables of an interval's rows
// stay in the vector L1 -- the gathers, not the arithmetic, bound the draws (FDR).
// The construction is Vose's with two QUEUES filled in index order (small: n p < 1, large: the
// rest), so that it is one fixed sequence of double operations -- include/fpt.h states it, and anything that
// repeats it gets the same table bit for bit.  One wavefront per row: the probabilities and the queues are made by
// all lanes, the pairing loop (n steps, each depending on the one before) on registers filled 64 steps at a time.
//   entry = threshold << lg | alias;   draw: slot = word >> (32 - lg), t = word & (2^(32-lg) - 1),
//   outcome = t < threshold ? slot : alias
// Every row's entries start at row << lg_max (its z at the same index of `zt`); row_lg[row] = lg.
// A row with a NaN in it gets the identity table (outcome = slot: its z is the NaN).
constexpr int kAliasLgMax = 11;
constexpr double kAliasRest = 1.0 / 4294967296.0, kAliasHeavyRest = 1.0 / 16777216.0;
__host__ __device__ inline int alias_lg_of(int memo_obs) {  // the cap: 2^lg <= memo_obs, 1 <= lg <= 11
    int lg = 1;
    while (lg < kAliasLgMax && (2 << lg) <= memo_obs) ++lg;
    return lg;
}

__global__ void __launch_bounds__(64) k_nb_alias(const double2 *__restrict__ memo, int memo_exp, int memo_obs, int lg_max,
                                                 uint32_t *__restrict__ alias, double *__restrict__ zt,
                                                 uint8_t *__restrict__ row_lg) {
    extern __shared__ double alias_smem[];
    const int lane = threadIdx.x;
    const size_t r = (size_t)blockIdx.y * memo_exp + blockIdx.x;
    const double2 *row = memo + r * memo_obs;
    // the row's width: lane i tries lg = i + 1 (cdf(2^lg - 2) exists: 2^lg_max <= memo_obs, or the memo is a single
    // column and the only outcome besides the rest is k = 0)
    int lg = lg_max;
    {
        const int idx = (2 << lane) - 2;
        const bool ok = lane < lg_max && idx < memo_obs && 1.0 - row[idx].x <= kAliasRest;
        const unsigned long long m = __ballot(ok);
        if (m) lg = __ffsll((long long)m);
    }
    const int n = 1 << lg;
    double *q = alias_smem;
    uint16_t *sq = reinterpret_cast<uint16_t *>(q + ((size_t)1 << lg_max)), *lq = sq + ((size_t)1 << lg_max),
             *al = lq + ((size_t)1 << lg_max);
    bool bad = false;
    int ns = 0, nl = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        double qq = 0.0;
        if (k < n) {
            const double2 e = k < memo_obs ? row[k] : make_double2(1.0, 0.0);
            const double below = k > 0 ? row[k - 1].x : 0.0;
            double pm = (k == n - 1 ? 1.0 : e.x) - below;
            bad |= !(pm == pm);
            pm = pm > 0.0 ? pm : 0.0;  // (a cdf that steps back by a rounding: probability 0)
            qq = pm * (double)n;
            q[k] = qq;
            al[k] = (uint16_t)k;
            zt[(r << lg_max) + k] = e.y;
        }
        const bool small = k < n && qq < 1.0, large = k < n && !small;
        const unsigned long long ms = __ballot(small), ml = __ballot(large);
        const unsigned long long below_me = (1ull << lane) - 1ull;
        if (small) sq[ns + __popcll(ms & below_me)] = (uint16_t)k;
        if (large) lq[nl + __popcll(ml & below_me)] = (uint16_t)k;
        ns += __popcll(ms);
        nl += __popcll(ml);
    }
    const bool any_bad = __ballot(bad) != 0ull;
    __syncthreads();
    if (!any_bad) {
        // The pairing is a chain -- every step needs the large outcome's running value of the step before --
        // but its reads need not be: the wavefront loads the next 64 entries of the small queue and their
        // probabilities at once (lane j the j-th), and the steps take them out of the registers with
        // v_readlane; the current large outcome and its value stay in registers until it turns small.  All
        // lanes run the steps on the same values (lane 0 stores); what is appended to the queue meanwhile is
        // picked up by the next load.  The same pairs in the same order as a plain loop over the queues.
        int si = 0, li = 0, se = ns, l = 0;
        double ql = 0.0;
        if (nl > 0) {
            l = lq[0];
            ql = q[l];
        }
        while (si < se && li < nl) {
            const int chunk = se - si < kWave ? se - si : kWave;
            const int sj = lane < chunk ? (int)sq[si + lane] : 0;
            const double qj = q[sj];
            const int qj_lo = __double2loint(qj), qj_hi = __double2hiint(qj);
            for (int j = 0; j < chunk && li < nl; ++j, ++si) {
                const int s = __builtin_amdgcn_readlane(sj, j);
                const double qs = __hiloint2double(__builtin_amdgcn_readlane(qj_hi, j), __builtin_amdgcn_readlane(qj_lo, j));
                if (lane == 0) al[s] = (uint16_t)l;
                ql = (ql + qs) - 1.0;
                if (ql < 1.0) {
                    if (lane == 0) {
                        q[l] = ql;
                        sq[se] = (uint16_t)l;
                    }
                    ++se;
                    ++li;
                    if (li < nl) {
                        l = lq[li];
                        ql = q[l];
                    }
                }
            }
        }
        if (lane == 0) {
            // what is left on either queue is 1 up to rounding: its own slot, whole
            while (li < nl) q[lq[li++]] = 1.0;
            while (si < se) {
                const int s = sq[si++];
                q[s] = 1.0;
                al[s] = (uint16_t)s;
            }
        }
    }
    __syncthreads();
    const uint32_t top = 0xffffffffu >> lg;  // the largest threshold: 32 - lg bits
    for (int k = lane; k < n; k += 64) {
        uint32_t th = top, ak = (uint32_t)k;
        if (!any_bad) {
            const double t = floor(ldexp(q[k], 32 - lg) + 0.5);
            th = t >= (double)top ? top : (uint32_t)t;
            ak = al[k];
        }
        alias[(r << lg_max) + k] = (th << lg) | ak;
    }
    // (0x80: a capped row whose rest is not negligible -- its intervals go to the full draw launch straight away)
    if (lane == 0) row_lg[r] = (uint8_t)(lg | ((lg == lg_max && !(1.0 - row[n - 2 < memo_obs ? n - 2 : 0].x <= kAliasHeavyRest)) ? 0x80 : 0));
}

// Draw beyond the table (or at a non-integer expected value): gallop, then bisect on the direct
// evaluation.  lo = largest k known to have cdf(k) < u.
__device__ __forceinline__ double2 nb_inverse_cdf_direct(const double *par, double ex, double u, int lo) {
    bool zd = false;
    const double r = fptm::fit_r(par + 9, ex, &zd);
    const double mu = fptm::fit_mu(par, ex);
    const double pr = r / (r + mu);
    int step = 1, hi = lo + 1;
    double chi = fptm::nb_cdf(hi, pr, r);
    while (chi < u && hi < (1 << 28)) {
        lo = hi;
        step <<= 1;
        hi = lo + step;
        chi = fptm::nb_cdf(hi, pr, r);
    }
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        const double cm = fptm::nb_cdf(mid, pr, r);
        if (cm >= u) {
            hi = mid;
            chi = cm;
        } else {
            lo = mid;
        }
    }
    return make_double2(chi, fptm::ndtri(1.0 - chi));
}

// ei: table row of the position's expected value, or -1 when it is not an integer inside the
// table (then *exp_ptr is read and the draw is evaluated directly).
__device__ __forceinline__ int table_row_of(double ex, int memo_exp) {
    const int ei = (int)ex;
    return (ex >= 0.0 && ex < (double)memo_exp && (double)ei == ex) ? ei : -1;
}
// the same with the width of the row's alias table above the row: row | lg << 16, or -1
__device__ __forceinline__ int alias_row_of(double ex, int memo_exp, const uint8_t *row_lg) {
    const int ei = table_row_of(ex, memo_exp);
    return ei >= 0 ? (ei | ((int)(row_lg[ei] & 0x1f) << 16)) : -1;
}

// (the tables are addressed base + 32-bit byte offset: one vector register and no 64-bit address
// arithmetic per gather; a model's alias table is at most 4096 x 2048 x 4 bytes, its z table twice that)
template <typename T>
__device__ __forceinline__ T table_at(const void *base, uint32_t byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
// N draws (2 or 4) at the same expected value, in step: the N alias entries go out together, then
// the N z values -- a draw is two dependent trips to the L1 / L2 and the draws of a Philox block share them.
// z[j] = ndtri(1 - cdf(k)) of the outcome k drawn with word w[j].  The outcome "n-1 or more" (and a
// position without a table row) is settled by the direct inverse cdf: the former with the uniform that
// the word's low bits pick inside the rest (1 - cdf(n-2)), the latter with u[j].
// rl: alias_row_of() of the position.
// "Which draws still need the direct evaluation" is kept as wavefront masks in scalar registers --
// ballots combined with scalar logic, turned back into a lane condition where a select needs one
// (inverse ballot: free); as bools the compiler materialises each as 0 / 1 in a vector register.
typedef unsigned long long lane_mask;
#define FPT_BALLOT(x) __builtin_amdgcn_ballot_w64(x)
#define FPT_LANE(m) __builtin_amdgcn_inverse_ballot_w64(m)
// LIGHT: the direct evaluation is left out (it is what the draw loop's 121 registers are for): the return
// value says that a draw needed it, and the caller hands the interval to the full instance.
template <int N, bool LIGHT>
__device__ __forceinline__ bool nb_draw_zn(const double2 *memo, const uint32_t *alias, const double *zt, int memo_obs,
                                           int lg_max, const double *par, int rl, const double *exp_ptr,
                                           const uint32_t (&w)[N], const double (&u)[N], double (&z)[N]) {
    lane_mask D[N];  // still to be evaluated directly
    // (a mask is the same for every lane and must not be assigned under a lane's condition: the few
    // lanes without a table row read row 0 along with the others and their result is dropped)
    const lane_mask tabled = FPT_BALLOT(rl >= 0), untabled = FPT_BALLOT(rl < 0);
    const uint32_t eiu = rl >= 0 ? (uint32_t)rl & 0xffffu : 0u, lg = rl >= 0 ? (uint32_t)rl >> 16 : 1u;
    const uint32_t arow = eiu << (lg_max + 2), zrow = eiu << (lg_max + 3), last = (1u << lg) - 1u;
    uint32_t e[N], slot[N], k[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        slot[j] = w[j] >> (32u - lg);
        e[j] = table_at<uint32_t>(alias, arow + 4u * slot[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) k[j] = (w[j] << lg) < (e[j] & ~last) ? slot[j] : (e[j] & last);
#pragma unroll
    for (int j = 0; j < N; ++j) z[j] = table_at<double>(zt, zrow + 8u * k[j]);
    lane_mask any_direct = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        D[j] = (FPT_BALLOT(k[j] == last) & tabled) | untabled;
        any_direct |= D[j];
    }
    if (LIGHT) return any_direct != 0;
    if (any_direct) {  // rare: beyond the table or a non-integer expected value
        const double ex = rl >= 0 ? (double)eiu : *exp_ptr;
        // the rest of the row: (cdf(n-2), 1]
        const double base = (rl >= 0 && last > 1u) ? memo[(size_t)eiu * memo_obs + (last - 1u)].x
                                                    : ((rl >= 0) ? memo[(size_t)eiu * memo_obs].x : 0.0);
        // one copy of the evaluation (it is ~8,000 instructions): a lane's draws that need it take
        // turns, picked with selects (an array indexed by the turn would live in scratch memory)
        for (;;) {
            int j = -1;
            double uj = 0.0;
            uint32_t wj = 0u, ej = 0u;
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                j = FPT_LANE(D[i]) ? i : j;
                uj = FPT_LANE(D[i]) ? u[i] : uj;
                wj = FPT_LANE(D[i]) ? w[i] : wj;
                ej = FPT_LANE(D[i]) ? e[i] : ej;
            }
            double zz = 0.0;
            if (j >= 0) {
                int lo = -1;
                if (rl >= 0) {
                    // where in its share of the slot the word fell: the slot's own part [0, threshold)
                    // or the alias part [threshold, 2^(32-lg))
                    const uint32_t t = (wj << lg) >> lg, th = ej >> lg, span = (0xffffffffu >> lg) + 1u;
                    const bool own = t < th;
                    const double frac = ((double)(own ? t : t - th) + 0.5) / (double)(own ? th : span - th);
                    uj = fma(1.0 - base, frac, base);
                    lo = (int)last - 1;
                }
                zz = nb_inverse_cdf_direct(par, ex, uj, lo).y;
            }
            any_direct = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                z[i] = j == i ? zz : z[i];
                D[i] &= ~FPT_BALLOT(j == i);
                any_direct |= D[i];
            }
            if (!any_direct) break;
        }
    }
    return false;
}
#undef FPT_BALLOT
#undef FPT_LANE

// A null window p-value is ndtr(y), y = -(sum of z) / sqrt(K), and only its RANK among the
// interval's observed values is needed.  ndtr is monotone, so "observed P < ndtr(y)" is
// "T(P) <= y" with T(P) the smallest double y whose ndtr(y) exceeds P.  Translating the L observed
// values once (a few ndtr evaluations each: start at ndtri(P), gallop and bisect over the
// ordered-integer image of the doubles) replaces `times` ndtr evaluations per base by comparisons
// of y.  Where the device ndtr is not monotone to the last bit, a null value within rounding of
// an observed one may land on the other side -- the tolerance ranks already have.
__device__ __forceinline__ long long ordered_bits(double y) {
    const long long b = __double_as_longlong(y);
    return b < 0 ? (long long)(0x8000000000000000ull - (unsigned long long)b) : b;
}
__device__ __forceinline__ double from_ordered_bits(long long k) {
    const long long b = k < 0 ? (long long)(0x8000000000000000ull - (unsigned long long)k) : k;
    return __longlong_as_double(b);
}
// ndtr(a), and for a > 0 the addend it is the sum of (ndtr(a) = base + t rounded once: fpt_math.hpp): one
// evaluation for both
__device__ __forceinline__ double ndtr_and_addend(double a, double &t, double &base, double &ec) {
    t = fptm::ndtr_addend_pos(a, base, &ec);  // (any sign: 0.5 erf in the centre, -Phi(-|a|) in the tails)
    return (base == 1.0 && a < 0.0) ? -t : base + t;
}
__device__ __forceinline__ double ndtr_threshold_from(double y, double P, bool have0 = false, double t0 = 0.0, double base0 = 0.0,
                                                      double ec0 = 1.0);
__device__ __forceinline__ double ndtr_threshold(double P) {
    if (!(P < 1.0)) return fptm::kInf;  // ndtr never exceeds 1
    if (P < 0.0) return -fptm::kInf;
    const double y0 = P > 0.0 ? fptm::ndtri(P) : -39.0;
    long long lo, hi, step = 1;  // ndtr(lo) <= P < ndtr(hi)
    const double p0 = fptm::ndtr(y0);
    // (in the upper tail a plateau is thousands of values wide and ndtri lands on it: the search that
    // starts from a point of the plateau knows where it ends)
    if (p0 == P && y0 > 0.0) return ndtr_threshold_from(y0, P);
    if (p0 > P) {
        hi = ordered_bits(y0);
        for (;;) {
            const long long c = hi - step;
            if (!(fptm::ndtr(from_ordered_bits(c)) > P)) {
                lo = c;
                break;
            }
            hi = c;
            step <<= 1;
        }
    } else {
        lo = ordered_bits(y0);
        for (;;) {
            const long long c = lo + step;
            if (fptm::ndtr(from_ordered_bits(c)) > P) {
                hi = c;
                break;
            }
            lo = c;
            step <<= 1;
        }
    }
    while (hi - lo > 1) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (fptm::ndtr(from_ordered_bits(mid)) > P) hi = mid; else lo = mid;
    }
    return from_ordered_bits(hi);
}

// The same threshold when a y with ndtr(y) == P is at hand (the observed window's own y).  The upper
// end of P's plateau is where the search has to go, and how far that is depends on y: the plateau is
// about ulp(P) / (phi(y) ulp(y)) values of y wide -- 1 for y < 0, tens to 10^12 in the upper tail
// (windows that are not depleted, p next to 1: most windows of most data), 2^40 next to 0.  A gallop
// from y by powers of two and a bisection are 2 log2(width) evaluations of ndtr
// (tools/micro/thr_evals.hip: 88 in the wavefront of the largest y when the windows sit at y ~ 6,
// 47 when the gallop starts from the estimated width).  For y > 0 the search therefore starts where
// the plateau must end: ndtr(a) = base + t(a) rounded once (base 0.5 or 1: the central and the tail
// branch of ndtr.c), t resolves a far finer than the sum does, the sum first rounds above P at
// t* ~ (P - base) + ulp(P)/2, and Newton's iteration on t (central branch: dt/da = phi(a); tail:
// on ln(-t), whose slope is the hazard sqrt(2/pi) / erfce, nearly constant over a plateau however
// wide) lands within a few values of the end in two to four steps.  The gallop and the bisection from
// there -- on ndtr itself, as before -- make it four to six evaluations in every wavefront, whatever
// the windows look like.  ndtr is not monotone to the last bit, so "the" end of a plateau is one of a
// few neighbouring values whichever way it is searched: this search and the plain one differ by a
// value or two of y in 0.3 % of the cases around y = 0 and in none in the tail.
// have0: the caller made P as ndtr(y) from y's own addend (ndtr_and_addend): t0, base0, ec0 are the first
// iteration's evaluation.
__device__ __forceinline__ double ndtr_threshold_from(double y, double P, bool have0, double t0, double base0, double ec0) {
    long long lo = ordered_bits(y), hi, step = 1;  // ndtr(lo) <= P
    long long k = lo + 1;
    if (y > 0.0) {
        const bool central = fptm::ndtr_is_central(y);
        const double up = __longlong_as_double(__double_as_longlong(P) + 1) - P;  // ulp(P), P in [0.5, 1)
        double a = y;
#pragma clang loop unroll(disable)
        for (int it = 0; it < 6; ++it) {
            if (fptm::ndtr_is_central(a) != central || !(a < 40.0)) break;
            double base = base0, ec = ec0, t = t0;
            if (it > 0 || !have0) t = fptm::ndtr_addend_pos(a, base, &ec);
            const double ts = (P - base) + 0.5 * up;
            const double d = central ? (ts - t) / (exp(-0.5 * a * a) * 0.3989422804014327)
                                     : log(t / ts) * ec * 1.2533141373155003;
            const double an = a + d;
            const double ua = __longlong_as_double(__double_as_longlong(a) + 1) - a;
            if (!(an > y)) break;
            a = an;
            // converged: the step was a few values of a -- or what a further step would correct, the second-order
            // term d^2 |t''/t'| / 2 ~ d^2 max(a, 1) / 2 (both branches: phi'/phi = -a, and the hazard's slope is
            // below 1), is under a value of a: no evaluation spent on seeing a step of nothing (the plateau is
            // 10^-13 wide in a where it is 300 values of a wide; only beyond y ~ 7 is it wide enough to bend)
            if (fabs(d) <= 4.0 * ua || d * d * fmax(a, 1.0) < 0.5 * ua) break;
        }
        if (a > y && a < 40.0) k = ordered_bits(a);
    }
    // The guess lies at or beyond the end of the plateau (then: DOWN to the first value still on it, not below
    // lo, which is) or on it (UP to the first value beyond).  One loop for both: a wavefront's lanes go either
    // way about evenly, and as two loops each direction waited for the other's evaluations of the exact cdf.
    const bool above = k > lo + 1 && fptm::ndtr(from_ordered_bits(k)) > P;
    long long cur = (above || k > lo + 1) ? k : lo, oth = lo;
    for (;;) {
        const long long c = above ? cur - step : cur + step;
        if (above && c <= lo) break;  // (oth = lo: lo itself is on the plateau)
        const bool ab = fptm::ndtr(from_ordered_bits(c)) > P;
        if (ab != above) {
            oth = c;
            break;
        }
        cur = c;
        step <<= 1;
    }
    hi = above ? cur : oth;
    lo = above ? oth : cur;
    while (hi - lo > 1) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (fptm::ndtr(from_ordered_bits(mid)) > P) hi = mid; else lo = mid;
    }
    return from_ordered_bits(hi);
}
// threshold of an observed value that is not below 1 (the edge positions' constant 1.0): above every
// finite y, below the +inf that stands for "not a number" in the sort
constexpr double kThresholdOfOne = 1e300;

// A null window's y is x / sqrt(K) rounded once, x = -(sum of z) (windowing.h:64), and that division
// is monotone in x: "T <= y" is "X(T) <= x" with X(T) the smallest x whose quotient reaches T.  With
// the thresholds translated once more, from y to x, a null window is ranked by its raw sum and the
// division per null value goes away.  (T sqrt(K) is within two ulps of X: a step or two.)
__device__ __forceinline__ double sum_threshold(double T, double sqrt_k, double inv_sqrt_k) {
    if (!(fabs(T) < 1e299)) return T;  // kThresholdOfOne and the infinities stay what they are
    long long k = ordered_bits(T * sqrt_k);
    while (div_invariant(from_ordered_bits(k - 1), sqrt_k, inv_sqrt_k) >= T) --k;
    while (!(div_invariant(from_ordered_bits(k), sqrt_k, inv_sqrt_k) >= T)) ++k;
    return from_ordered_bits(k);
}

typedef void (*fdr_kernel_t)(const fdr_args);

#ifdef FPT_FDR_MARKS
// -DFPT_FDR_MARKS builds: cycles of the first wavefront of every workgroup between the phase marks,
// summed over workgroups (FPT_FDR_PHASES=1 prints them after each launch)
__device__ unsigned long long g_fdr_phase[16];
#define FDR_MARK(n)                                                                  \
    if (threadIdx.x == 0) {                                                           \
        const unsigned long long now_ = wall_clock64();                               \
        atomicAdd(&g_fdr_phase[n], now_ - mark_);                                     \
        mark_ = now_;                                                                 \
    }
#define FDR_MARK_INIT unsigned long long mark_ = wall_clock64();
#else
#define FDR_MARK(n)
#define FDR_MARK_INIT
#endif

// GWS: the per-interval buffers live in global memory instead of LDS -- the same code for
// intervals too long for the 160 KB of a CU (one workgroup still owns one interval, and a
// workgroup's own global writes are visible to it after __syncthreads()).
// HSC: the half window width at compile time (3, the only one the reference uses: the window sums
// unroll into 14 LDS reads with immediate offsets), or 0 for any width.
// ONE: no interval of the launch is longer than the workgroup -- a lane has one base, and what
// belongs to the base (table row, Philox counter, addresses) is made once, not in every pass.
// MODE: 0 the whole pass; 1 the per-interval SET-UP alone (steps 0 - 1b: the observed windows re-made,
// sorted, translated into thresholds -- a chain of barrier-separated phases with ~4 exact normal cdfs
// per observed value, a third of a 50-draw call) with its results left in a workspace; 2 the DRAWS
// (steps 2 - 3) reading them back.  As a launch of its own the set-up has a third of the LDS and no
// part in the draw loop's 121 registers: twice as many workgroups cover each other's latencies.
// MODE 3: the draws WITHOUT the direct inverse cdf -- 58 registers instead of 127, eight wavefronts per SIMD
// instead of four.  An interval with a base off the tables (or on a capped row with a heavy rest) is
// marked by the set-up launch and skipped; one whose draw falls into a row's rest (2^-32 of the draws)
// marks itself and stores nothing.  The full instance (MODE 2 with redo_only) then does the marked ones.
template <int NT, bool GWS, int HSC, bool ONE, int MODE = 0>
__global__ void __launch_bounds__(NT, MODE == 3 ? (NT > 256 ? 4 : 8) : (MODE == 1 && NT <= 256 && !GWS) ? 7 : ((GWS || NT > 256) ? 2 : 4)) k_fdr_null(const fdr_args a) {
    extern __shared__ double smem[];
    constexpr bool DRAWS = MODE == 2 || MODE == 3, LIGHT = MODE == 3;
    const int n2 = a.n2_max;
    double *par = GWS ? reinterpret_cast<double *>(a.gws + (size_t)blockIdx.x * a.gws_stride) : smem;  // 24
    double *skey = par + 24;                         // n2 sorted observed values (NaN -> +inf)
    double *zb = skey + n2;                          // 4 x n2: z of the pass's four samples, the four of a position side by side
                                                     // (with wide windows: two arrays of n2 tile prefix sums)
    double *zalt = zb + (MODE == 1 ? 2 : 4) * n2;    // 4 x n2 more when a.dbuf (passes alternate between the sets); the
                                                     // set-up launch (MODE 1) uses two n2 of zb: the z, and the keys
    int *sidx = reinterpret_cast<int *>(zalt + (a.dbuf ? 4 * n2 : 0));  // n2 original positions
    int *nf = sidx + n2;                             // n2 tile prefix counts of non-finite z (16 bits per sample)
    int *hist = nf + n2;                             // n2 + 2 histogram / prefix
    int *misc = hist + n2 + 2;                       // [0] n_nan, [1] m
    // rank guide, fdr_guide_slices(n2) + 1 entries: #{sorted observed whose slice is < b}; 16-bit where the buffers
    // are in LDS (an interval there has at most 2,048 bases)
    typedef typename std::conditional<GWS, int, uint16_t>::type guide_t;
    guide_t *rguide = reinterpret_cast<guide_t *>(misc + 8);
    // (MODE 1 is launched with the buffers it uses -- par, skey, two n2 of zb, sidx, nf, hist, misc:
    // fdr_setup_lds_bytes -- and never touches rguide, whose address then lies beyond its allocation)

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int hs = HSC ? HSC : a.hw;
    const int64_t iv = a.iv_list ? (int64_t)a.iv_list[blockIdx.x] : a.iv_first + blockIdx.x;
    // (the full draw launch behind a light one finds nearly every interval done: out before anything else is read)
    if (MODE == 2 && a.redo_only && a.ws_misc[3 * iv + 2] == 0) return;
    int64_t off;
    int L;
    if (a.interval_off) {
        off = a.interval_off[iv];
        L = (int)(a.interval_off[iv + 1] - off);
    } else {
        L = a.interval_len;
        off = iv * (int64_t)L;
    }
    if (L <= 0) return;
    const int Lr = (L + kWave - 1) & ~(kWave - 1);
    int np2 = 1;
    while (np2 < L) np2 <<= 1;
    // The launch and its buffers were sized from the HOST copy of the offsets; these come from the device.
    // If the two disagree (a stale host array) the interval does not fit: it is not processed -- NaN in its
    // efdr says so -- rather than written past the buffers.
    if (np2 > n2 || (ONE && L > NT) || (MODE != 0 && off + L > a.ws_total)) {
        if (MODE != 1)
            for (int i = tid; i < L; i += NT) a.efdr[off + i] = NAN;
        return;
    }

    if (LIGHT && a.ws_misc[3 * iv + 2] != 0) return;  // left to the full launch by the set-up
    const int dm = a.dm_ids ? a.dm_ids[iv] : 0;
    const double2 *memo = a.memo + (size_t)dm * a.memo_exp * a.memo_obs;
    const uint32_t *alias = a.alias + ((size_t)dm * a.memo_exp << a.alias_lg);
    const double *zt = a.zt + ((size_t)dm * a.memo_exp << a.alias_lg);
    const uint8_t *row_lg = a.row_lg + (size_t)dm * a.memo_exp;
    if (tid < 24) par[tid] = a.model[(size_t)dm * 24 + tid];
    FDR_MARK_INIT
    // ---- 0. with the observed counts at hand, the observed window p-values are re-made here by the
    // very operations the null windows go through below -- z of the (exp, obs) pair from the same
    // table, summed left to right, and the SAME normal cdf that the thresholds are searched with --
    // so that "null <= observed" is decided as in the reference, which compares two p-values that
    // went through one function: a null window made of the same counts ties exactly, and so does
    // one whose z sum differs in the last bits only (the same counts in another order; the cdf is
    // flat to double precision near 1).  Taken from the p-value track of the scan, whose cdf is a
    // different (faster) evaluation, such ties fall either way by rounding -- and with sparse counts
    // ties are a large share of the null: an all-zero window is its own most likely null draw.
    if (!DRAWS && a.obs) {
        for (int t = tid; t < L; t += NT) {
            const double ex = a.exp[off + t];
            const int ei = table_row_of(ex, a.memo_exp);
            const int32_t k = fptm::c_int(a.obs[off + t]);
            double z;
            if (ei >= 0 && k >= 0 && k < a.memo_obs) {
                z = memo[(size_t)ei * a.memo_obs + k].y;
            } else {  // beyond the table: evaluated directly (rare)
                bool zd = false;
                const double *par_src = a.model + (size_t)dm * 24;
                const double r = fptm::fit_r(par_src + 9, ex, &zd), mu = fptm::fit_mu(par_src, ex);
                z = nb_pz_direct(r, mu, k).y;
            }
            zb[t] = z;
        }
        __syncthreads();
    }
    FDR_MARK(0)  // step 0: z of the observed counts
    // ---- 1. sort the observed window p-values (NaN compares as +inf and ends up last); on the way
    // in, count the values that are not NaN (m) and those below 1 (the rank of the edge positions'
    // constant 1.0 among the thresholds)
    int n_num = 0, n_below_one = 0;
    double v_one = fptm::kInf;  // ONE: this lane's key
    for (int i = tid; !DRAWS && i < (ONE ? L : np2); i += NT) {
        double v = fptm::kInf;
        int id = -1;
        if (i < L) {
            const double P = a.winp[off + i];
            id = i;
            if (isnan(P)) {
                // not a number: compares as +inf, ends up last, is not counted
            } else if (!a.obs) {
                v = P;  // becomes a threshold in y after the sort (ndtr_threshold)
                n_num += 1;
                n_below_one += (P < 1.0) ? 1 : 0;
            } else if (!(P < 1.0)) {
                v = kThresholdOfOne;
                n_num += 1;  // an edge position (or p = 1): above every null window
            } else {
                // this position's window p-value once more, from its own y through the function the
                // thresholds below are searched with
                double sm = NAN;
                if (i >= hs && i < L - hs) {
                    sm = 0.0;
                    for (int j = i - hs; j <= i + hs; ++j) sm += zb[j];
                }
                if (isfinite(sm)) {
                    // the sort key is this y: the order of the p-values (ndtr is monotone; the order
                    // among equal p-values does not matter), and what the threshold search starts from
                    v = -div_invariant(sm, a.sqrt_k, a.inv_sqrt_k);
                    n_num += 1;
                }  // else: the tracks disagree (a p-value where the counts give none): not a number
            }
        }
        if (ONE) {
            v_one = v;
            zb[n2 + i] = v;  // (beside the z of step 0, which other lanes may still be reading)
        } else {
            skey[i] = v;
            sidx[i] = id;
        }
    }
    if (DRAWS) {  // what the set-up launch left: sorted thresholds, their positions, m and rank_one
        for (int i = tid; i < (ONE ? L : np2); i += NT) {
            skey[i] = i < L ? a.ws_key[off + i] : fptm::kInf;
            sidx[i] = i < L ? (int)a.ws_idx[off + i] : -1;
        }
    }
    for (int i = tid; i < (ONE && !DRAWS && !GWS ? n2 : np2) + 2; i += NT) hist[i] = 0;
    if (tid < 4) misc[tid] = (DRAWS && tid >= 1 && tid <= 2) ? a.ws_misc[3 * iv + tid - 1] : 0;
    __syncthreads();
    FDR_MARK(1)  // keys
    if (n_num) atomicAdd(&misc[1], n_num);
    if (n_below_one) atomicAdd(&misc[2], n_below_one);
    if (ONE && !DRAWS) {
        // One key per lane: its place in the order is the number of keys that come before it (equal keys in the
        // order of their positions).  Counting them all -- a walk over the interval's keys, 4 L instructions per
        // wavefront -- was a quarter of the set-up's instructions; the keys are FILED first instead: n2
        // buckets by a monotone function of the key (window y values spread like a standard normal, p-values
        // over [0, 1]; the last bucket takes 1.0, the edges and the NaNs), a count per bucket by LDS atomics, their
        // prefix, every key written to its bucket's range, and a key then counts only the keys of its own
        // bucket that come before it: one or two where there were L.  (A bucket full of equal keys -- the all-zero
        // windows of sparse data -- is counted as before.)  The bitonic network below issues ~2,000 instructions
        // for 256 slots and meets at 36 barriers.
        double *tkey = zb;   // the z of step 0 are done with (every lane is past the barrier above)
        int *tidx = nf, *cnt = hist;  // (hist is zero; it is zeroed again below)
        const double bscale = a.obs ? (double)(n2 - 2) / 12.0 : (double)(n2 - 1);
        int b = n2 - 1;
        if (v_one < 1e299) {
            const double f = a.obs ? fma(v_one, bscale, 6.0 * bscale) : v_one * bscale;
            b = (int)fmin(fmax(f, 0.0), (double)(n2 - 2));
        }
        int slot = 0;
        if (tid < L) slot = atomicAdd(&cnt[b], 1);
        __syncthreads();
        if (tid < kWave) {  // exclusive prefix of the counts, in place
            int carry = 0;
            for (int base = 0; base < n2; base += kWave) {
                const int c = cnt[base + lane];
                const int incl = wave_scan_i32(c) + carry;
                cnt[base + lane] = incl - c;
                carry = __shfl(incl, kWave - 1, kWave);
            }
            if (lane == 0) cnt[n2] = carry;
        }
        __syncthreads();
        const int first = cnt[b], last = cnt[b + 1];
        if (tid < L) {
            tkey[first + slot] = v_one;
            tidx[first + slot] = tid;
        }
        __syncthreads();
        if (tid < L) {
            int rank = first;
            for (int j = first; j < last; ++j) {
                const double kj = tkey[j];
                rank += (kj < v_one || (kj == v_one && tidx[j] < tid)) ? 1 : 0;
            }
            skey[rank] = v_one;
            sidx[rank] = tid;
        }
        __syncthreads();
        for (int i = tid; i < n2 + 2; i += NT) hist[i] = 0;
    }
    // (barrier-free stages for partner distances below 64 -- a wavefront owns whole 64-element
    // blocks -- were measured: 6 instead of 36 barriers for 256 elements, no change in time)
    for (int k = 2; k <= ((ONE || DRAWS || ABL(16384)) ? 0 : np2); k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const double x = skey[i], y = skey[ixj];
                    const int xi = sidx[i], yi = sidx[ixj];
                    // ties broken by position so the order is total (pads, id -1 -> last)
                    const bool gt = (x > y) || (x == y && (unsigned)xi > (unsigned)yi);
                    const bool up = (i & k) == 0;
                    if (gt == up) {
                        skey[i] = y; skey[ixj] = x;
                        sidx[i] = yi; sidx[ixj] = xi;
                    }
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    FDR_MARK(2)  // sort
    const int m = misc[1];         // observed values that are not NaN (NaN / pads were mapped to +inf)
    // ---- 1b. the sorted observed values become thresholds in y (see ndtr_threshold)
    // (after the sort: neighbouring lanes then search neighbouring values, whose searches are about
    // equally long -- in the order of the positions a wavefront waits for its one value in the tail)
    if (DRAWS) {
    } else if (!a.obs) {
        for (int i = tid; i < m; i += NT)
            skey[i] = sum_threshold(ABL(8192) ? fptm::ndtri(skey[i]) : ndtr_threshold(skey[i]), a.sqrt_k, a.inv_sqrt_k);
        __syncthreads();
    } else {
        int below = 0;
        for (int i = tid; i < m; i += NT) {
            const double yo = skey[i];
            if (yo == kThresholdOfOne) continue;
            // the observed window p-value, by the function the search uses -- and for y > 0 as the sum the search
            // starts from: its first look at the addend is this one
            double t0 = 0.0, base0 = 0.0, ec0 = 1.0;
            const double pv = ndtr_and_addend(yo, t0, base0, ec0);
            if (pv < 1.0) {
                skey[i] = sum_threshold(ABL(8192) ? yo : ndtr_threshold_from(yo, pv, yo > 0.0, t0, base0, ec0), a.sqrt_k, a.inv_sqrt_k);
                below += 1;
            } else {
                skey[i] = kThresholdOfOne;
            }
        }
        if (below) atomicAdd(&misc[2], below);
        __syncthreads();
    }
    // rank guide: x / sqrt(K) of a null window is about standard normal; nb slices of [-kYR, kYR) (the
    // first and last reach to infinity) bracket #{X <= x}, so a rank needs a probe or two, not log2(L)
    FDR_MARK(3)  // thresholds
    const int rank_one = misc[2];  // thresholds of values below 1
    if (MODE == 1) {  // hand over and stop: the draw launch picks these up
        for (int i = tid; i < L; i += NT) {
            a.ws_key[off + i] = skey[i];
            a.ws_idx[off + i] = (uint16_t)sidx[i];
        }
        // a base off the tables (or on a capped row with a heavy rest) needs the direct inverse cdf
        int off_table = 0;
        for (int t = tid; t < L; t += NT) {
            const int ei = table_row_of(a.exp[off + t], a.memo_exp);
            off_table |= (ei < 0 || (row_lg[ei] & 0x80)) ? 1 : 0;
        }
        off_table = __syncthreads_or(off_table);
        if (tid == 0) {
            a.ws_misc[3 * iv] = m;
            a.ws_misc[3 * iv + 1] = rank_one;
            a.ws_misc[3 * iv + 2] = off_table;
        }
        return;
    }
    const double kYR = 4.5 * a.sqrt_k;
    // Four slices per observed value (of the power of two above the interval's length; at most 4,096): the largest
    // bracket of a wavefront's 256 null values -- which is how long its rank search runs -- holds three or four
    // thresholds instead of seven or eight.  (Round 3 measured the same at four wavefronts per SIMD and with the
    // guide as 32-bit entries, a quarter of the workgroups lost to LDS: slower then.)
    const int nb = fdr_guide_slices(np2);
    const double yscale = (double)nb / (2.0 * kYR);
    // rguide[b] = #{thresholds below 1 whose slice is < b}, with the slice of a threshold found by the
    // very expression a null window's is below -- it is monotone in x, so the thresholds <= x lie
    // between rguide[slice(x)] and rguide[slice(x) + 1] whatever the rounding does at a slice's
    // edge.  Filled from the sorted thresholds: the one at place i writes i to the slices after its
    // predecessor's up to its own (one or two on average) -- no search (a bisection per slice was
    // log2(L) dependent LDS reads, a quarter of an interval's set-up time).
    // (one fused multiply-add, two clamps, one conversion: any monotone function of x will do, and a
    // NaN -- whose rank is not used -- lands in slice 0)
    const double slice_c0 = kYR * yscale, slice_top = (double)(nb - 1);
    auto slice_of = [&](double y) { return (int)fmin(fmax(fma(y, yscale, slice_c0), 0.0), slice_top); };
    for (int i = tid; i <= rank_one; i += NT) {
        const int from = i == 0 ? 0 : slice_of(skey[i - 1]) + 1;
        const int to = i == rank_one ? nb : slice_of(skey[i]);  // (the last entry: every threshold below 1)
        for (int b = from; b <= to; ++b) rguide[b] = (guide_t)i;
    }
    __syncthreads();

    FDR_MARK(4)  // rank guide
    // ---- 2. null tracks, four samples per pass with narrow windows (one Philox block feeds the
    //         four), two with wide ones
    // Narrow windows (the reference only ever uses hw = 3) are summed directly from the raw z in
    // LDS, left to right like windowing.h:53-67: 2*hs+1 reads and adds are fewer instructions than
    // three prefix scans plus three tile-range sums, and a non-finite z shows up as a non-finite
    // sum, so no separate count is needed.  Wider windows take the prefix-scan path.
    const bool direct = hs <= 8;
    int ei_one = -1;  // ONE: the table row of this lane's base
    if (ONE) {
        if (tid < L) ei_one = alias_row_of(a.exp[off + tid], a.memo_exp, row_lg);
    } else if (direct) {
        for (int t = tid; t < L; t += NT) nf[t] = alias_row_of(a.exp[off + t], a.memo_exp, row_lg);
        // (each lane reads back only the entries it wrote: no barrier needed)
    }
    // the loop over a lane's bases: a single trip when ONE
    constexpr int kStride = ONE ? (1 << 30) : NT;
    // With a second set of z buffers a pass writes one set while slower wavefronts may still be
    // reading the other, so the barrier at the end of a pass is not needed (direct windows only).
    const bool alternate = a.dbuf && direct;
    const int spp = direct ? 4 : 2;  // samples per pass
    double *const zset0 = zb, *const zset1 = zalt;  // 4 (direct) or 2 buffers of n2 each
    int pass = 0;
    bool left_out = false;  // LIGHT: a draw needed the direct evaluation (wave-uniform)
    for (int s = 0; s < a.times; s += spp, ++pass) {
        const int ns = a.times - s < spp ? a.times - s : spp;  // samples of this pass
        double *const zq = (alternate && (pass & 1)) ? zset1 : zset0;
        for (int t = tid; t < Lr; t += kStride) {  // wave-uniform bound
            uint32_t o[4] = {0u, 0u, 0u, 0u};  // the Philox block of this base and pass (words = samples)
            int ei = -1;
            // (the caller's uniforms are a test hook: whether there are any is a scalar condition, not a lane's pointer)
            // (... and the light instance carries neither hook: fewer scalar registers live across the loop, where the
            // compiler was spilling them into a vector register's lanes -- v_readlane_b32 in every pass)
            const bool has_up = !LIGHT && a.null_uniform != nullptr;
            const double *up = LIGHT ? nullptr : a.null_uniform + (size_t)(off + (t < L ? t : 0)) * a.times + s;
            if (t < L) {
                if (ABL(32768)) {  // timing only: four well-mixed words for five multiplications instead of Philox's sixty instructions
                    uint32_t h = ((uint32_t)(a.base_index0 + off + t) * 2654435761u) ^ ((uint32_t)s * 0x9e3779b9u);
                    h ^= h >> 15;
                    o[0] = h * 0x85ebca6bu, o[1] = h * 0xc2b2ae35u, o[2] = h * 0x27d4eb2fu, o[3] = h * 0x165667b1u;
                } else if (!has_up && !ABL(512)) {
                    const uint64_t base = (uint64_t)(a.base_index0 + off + t);
                    philox4x32_10((uint32_t)base, (uint32_t)(base >> 32), (uint32_t)(s >> 2), 0x66707464u,
                                  (uint32_t)a.seed, (uint32_t)(a.seed >> 32), o);
                    if (s & 2) {  // a pass of two samples (wide windows) on the upper half of a block
                        o[0] = o[2];
                        o[1] = o[3];
                    }
                }
                // the table row of a position does not change from pass to pass: with direct
                // windows `nf` is free and holds it (filled above, before the first pass)
                ei = ONE ? ei_one : (direct ? nf[t] : alias_row_of(a.exp[off + t], a.memo_exp, row_lg));
            }
            // u = (word + 1/2) 2^-32 (philox_uniform4), or the caller's uniforms (tests)
            if (direct) {  // the four draws of the block in step, their z side by side in the buffer
                const uint32_t w4_[4] = {o[0], o[1], o[2], o[3]};
                uint32_t w4[4];
                double u4[4], z4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w4[j] = w4_[j];
                    u4[j] = fma((double)w4[j], 1.0 / 4294967296.0, 0.5 / 4294967296.0);
                    if (ABL(512)) u4[j] = 0.37 + 1e-3 * s + 0.04 * j, w4[j] = fptm::uniform_word(u4[j]);
                }
                if (has_up) {  // (tests)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        u4[j] = (t < L && j < ns) ? up[j] : 0.5;
                        w4[j] = fptm::uniform_word(u4[j]);
                    }
                }
                if (t < L) {
                    if (ABL(1024)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) z4[j] = u4[j] - 0.5;
                    } else {
                        left_out |= nb_draw_zn<4, LIGHT>(memo, alias, zt, a.memo_obs, a.alias_lg, par, ei, a.exp + off + t, w4, u4, z4);
                    }
                    // (two arrays of pairs, not one of quadruples: consecutive lanes then touch consecutive 16
                    // bytes, and a 16-byte access at a 32-byte stride is a two-way bank conflict)
                    if (ABL(65536)) {  // timing only: the z of a pass never reach LDS (kept alive through an impossible branch)
                        if (z4[0] + z4[1] + z4[2] + z4[3] == 12345.678) reinterpret_cast<double2 *>(zq)[t] = make_double2(z4[0], z4[1]);
                    } else {
                        reinterpret_cast<double2 *>(zq)[t] = make_double2(z4[0], z4[1]);
                        reinterpret_cast<double2 *>(zq + 2 * n2)[t] = make_double2(z4[2], z4[3]);
                    }
                }
            } else {  // wide windows: two draws per pass, through the scans
                uint32_t w2[2] = {o[0], o[1]};
                double u2[2], z2[2] = {0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u2[j] = fma((double)w2[j], 1.0 / 4294967296.0, 0.5 / 4294967296.0);
                    if (has_up && t < L) {
                        u2[j] = j < ns ? up[j] : 0.5;
                        w2[j] = fptm::uniform_word(u2[j]);
                    }
                    if (ABL(512)) u2[j] = 0.37 + 1e-3 * s + 0.04 * j, w2[j] = fptm::uniform_word(u2[j]);
                }
                if (t < L) {
                    if (ABL(1024)) z2[0] = u2[0] - 0.5, z2[1] = u2[1] - 0.5;
                    else left_out |= nb_draw_zn<2, LIGHT>(memo, alias, zt, a.memo_obs, a.alias_lg, par, ei, a.exp + off + t, w2, u2, z2);
                }
                const bool f0 = isfinite(z2[0]), f1 = isfinite(z2[1]);
                const int zc = (t < L) ? ((f0 ? 0 : 1) | (f1 ? 0 : 1 << 16)) : 0;
                zq[t] = scan_add(f0 ? z2[0] : 0.0);
                zq[n2 + t] = scan_add(f1 ? z2[1] : 0.0);
                nf[t] = scan_add(zc);
            }
        }
        FDR_MARK(7)  // a pass: Philox + draws
        if (!ABL(131072)) __syncthreads();  // (timing only: the windows then read whatever z are there)
        FDR_MARK(8)  // a pass: the barrier
        for (int t = tid; t < L; t += kStride) {
            // x = -(sum of z) of the null windows (y = x / sqrt(K) is what the thresholds were translated
            // from): NaN when a z in the window is not finite, +inf stands for the edges, whose window
            // p-value is the constant 1.0 (windowing.pyx:51) and which are part of the pooled null
            // (four named values: an array indexed by the pair would live in scratch memory)
            double x0 = fptm::kInf, x1 = fptm::kInf, x2 = fptm::kInf, x3 = fptm::kInf;
            if (direct) {
                if (t >= hs && t < L - hs) {
                    double sm[4] = {0.0, 0.0, 0.0, 0.0};
                    const double2 *zw01 = reinterpret_cast<const double2 *>(zq) + (t - hs);
                    const double2 *zw23 = reinterpret_cast<const double2 *>(zq + 2 * n2) + (t - hs);
                    if (ABL(2048)) {  // no window: the position's own four z
                        const double2 p01 = zw01[hs], p23 = zw23[hs];
                        sm[0] = p01.x, sm[1] = p01.y, sm[2] = p23.x, sm[3] = p23.y;
                    } else if constexpr (HSC > 0) {
                        // (the sum starts at the first z, not at 0.0 + z: the same value, but for the
                        // sign of a zero sum, which no comparison below sees)
                        const double2 f01 = zw01[0], f23 = zw23[0];
                        sm[0] = f01.x, sm[1] = f01.y, sm[2] = f23.x, sm[3] = f23.y;
#pragma unroll
                        for (int j = 1; j <= 2 * HSC; ++j) {
                            const double2 p01 = zw01[j], p23 = zw23[j];
                            sm[0] += p01.x, sm[1] += p01.y, sm[2] += p23.x, sm[3] += p23.y;
                        }
                    } else {
                        for (int j = 0; j <= 2 * hs; ++j) {
                            const double2 p01 = zw01[j], p23 = zw23[j];
                            sm[0] += p01.x, sm[1] += p01.y, sm[2] += p23.x, sm[3] += p23.y;
                        }
                    }
                    // -sum, or NaN where the sum is not finite: (s - s) is 0 or NaN, and two subtractions
                    // are half of a class test, a sign flip and two selects
                    x0 = (sm[0] - sm[0]) - sm[0];
                    x1 = (sm[1] - sm[1]) - sm[1];
                    x2 = (sm[2] - sm[2]) - sm[2];
                    x3 = (sm[3] - sm[3]) - sm[3];
                }
            } else if (t >= hs && t < L - hs) {
                const bool le3 = hs <= 64;
                const double s0 = le3 ? tile_range_sum3(zq, t - hs, t + hs) : tile_range_sum(zq, t - hs, t + hs);
                const double s1 = le3 ? tile_range_sum3(zq + n2, t - hs, t + hs) : tile_range_sum(zq + n2, t - hs, t + hs);
                const int sc = le3 ? tile_range_sum3(nf, t - hs, t + hs) : tile_range_sum(nf, t - hs, t + hs);
                x0 = (sc & 0xffff) ? NAN : -s0;
                x1 = (sc >> 16) ? NAN : -s1;
            }
            if (!LIGHT && a.null_out) {  // the p-values themselves only when somebody wants them
                double *np_ = a.null_out + (size_t)(off + t) * a.times + s;
#pragma clang loop unroll(disable)
                for (int w = 0; w < ns; ++w) {
                    const double xw = w == 0 ? x0 : (w == 1 ? x1 : (w == 2 ? x2 : x3));
                    np_[w] = xw == fptm::kInf ? 1.0 : fptm::ndtr(div_invariant(xw, a.sqrt_k, a.inv_sqrt_k));
                }
            }
            if (ABL(4096)) {
                if (x0 == 12345.0 || x1 == 12345.0 || x2 == 12345.0 || x3 == 12345.0) atomicAdd(&misc[0], 1);
                continue;
            }
            // rank = #{thresholds <= y}: bisect inside the guide's bracket, two samples in step
            // (for NaN the result is unused; +inf gets the precomputed rank of 1.0)
            auto rank_pair = [&](const double y0, const double y1, const bool second) {
                const int b0 = slice_of(y0), b1 = slice_of(y1);
                int l0 = rguide[b0], h0 = rguide[b0 + 1], l1 = rguide[b1], h1 = rguide[b1 + 1];
                // The first two probes go to the two ends of the bracket, then it is bisected: a bracket
                // of one or two thresholds costs what it did, and a bracket full of EQUAL thresholds -- the
                // windows of sparse data: hundreds of all-zero windows share one y, and most null windows
                // are that very window -- is settled by the ends instead of log2(size) probes (sparse
                // counts: 8.4e8 -> 1.0e9 bases/s, tools/diag_sparse_redo.py).  (skey[-1] is inside the
                // buffer: the slot before skey belongs to `par`.)
                // (measured and dropped: brackets of up to 8 thresholds counted front to back -- a third of a
                // bisection step's instructions per step, but a wavefront runs as many steps as its largest
                // bracket holds thresholds: 8.1 -> 8.5 ms per 100-draw call)
                for (int it = 0; l0 < h0 || l1 < h1; ++it) {
                    const int m0 = it == 0 ? h0 - 1 : (it == 1 ? l0 : (l0 + h0) >> 1);
                    const int m1 = it == 0 ? h1 - 1 : (it == 1 ? l1 : (l1 + h1) >> 1);
                    const bool g0 = skey[m0] <= y0, g1 = skey[m1] <= y1;
                    if (l0 < h0) {
                        l0 = g0 ? m0 + 1 : l0;
                        h0 = g0 ? h0 : m0;
                    }
                    if (l1 < h1) {
                        l1 = g1 ? m1 + 1 : l1;
                        h1 = g1 ? h1 : m1;
                    }
                }
                // (an edge position's +inf needs no special case: it falls in the last slice, whose bracket
                // ends at rank_one -- every threshold below 1 is <= +inf -- and the first probe settles it)
                atomicAdd(isnan(y0) ? &misc[0] : &hist[l0], 1);
                if (second) atomicAdd(isnan(y1) ? &misc[0] : &hist[l1], 1);
            };
            // (two copies of the search rather than a loop over the pairs: a loop picks its pair with four selects)
            rank_pair(x0, x1, ns > 1);
            if (ns > 2) rank_pair(x2, x3, ns > 3);
        }
        FDR_MARK(9)  // a pass: windows + ranks
        if (!alternate && !ABL(131072)) __syncthreads();
    }
    if (LIGHT) {  // one of this interval's draws fell into a row's rest: nothing is stored, the full launch does it
        if (__syncthreads_or(left_out ? 1 : 0)) {
            if (tid == 0) a.ws_misc[3 * iv + 2] = 1;
            return;
        }
    } else {
        __syncthreads();
    }

    FDR_MARK(5)  // the passes
    // ---- 3. counts: inclusive prefix of the histogram (one wavefront, carried over chunks)
    if (tid < kWave) {
        int carry = 0;
        for (int base = 0; base <= m; base += kWave) {
            const int i = base + lane;
            int v = (i <= m) ? hist[i] : 0;
            v = wave_scan_i32(v) + carry;
            if (i <= m) hist[i] = v;
            carry = __shfl(v, kWave - 1, kWave);
        }
    }
    __syncthreads();
    const int n_finite = hist[m];
    const int n_nan = misc[0];
    const double denom = (double)L * (double)a.times;
    for (int i = tid; i < (ONE ? L : np2); i += NT) {
        const int pos = sidx[i];
        if (pos < 0) continue;
        double f = 1.0;  // NaN observed: the two-pointer walk runs to the end (utils.pyx:76)
        if (i < m) {
            int cnt = hist[i];
            if (cnt == n_finite) cnt += n_nan;  // nothing finite above: the walk passes the NaNs too
            f = (double)cnt / denom;
            if (f > 1.0) f = 1.0;
        }
        a.efdr[off + pos] = f;
    }
    FDR_MARK(6)  // prefix and output
}

// ===========================================================================
// k_fdr_slice / k_fdr_slice_finish: the light draws of an interval of more than 256 bases as SLICES.
// An interval's null tracks are independent across positions but for the 2 hw + 1 window, and a draw
// is a function of its position (the Philox counter): so a long interval is drawn by several workgroups of
// the size that suits the short ones -- three wavefronts (in a uniform batch two to four, whichever leaves the
// fewest lanes idle over an interval: fdr_slice_lanes), one lane per position, lanes - 2 hw output positions each
// with hw halo positions either side drawn again -- every one holding the interval's
// thresholds, its rank guide and a private histogram that it adds to a global one at the end;
// k_fdr_slice_finish then turns the interval's counts into its efdr like step 3 of k_fdr_null.  As one
// workgroup an interval of 300 bases left five workgroups on a CU (its 512-entry buffers) and ran its lanes
// in two trips, one of 1,000 bases left two: 1.35 and 2.6 times the time per base of a short interval.
// Only the light form exists (split launches, the `detect` width): a marked interval is skipped here like in
// k_fdr_null<MODE 3>, one whose draw falls into a row's rest is marked by the slice that meets it, and the
// full launch that follows (MODE 2, redo_only, one workgroup per interval) does those.
// LDS: par 24 | skey n2 | z 4 x lanes | hist n2 + 2 (int) | misc 8 (int) | guide (16-bit).
// ===========================================================================
constexpr int kFdrSliceLanes = 192;  // ragged batches; a uniform batch takes 128, 192 or 256 (fdr_slice_lanes)
struct fdr_slice_args {
    fdr_args a;
    const int32_t *slice_iv;     // per workgroup: the interval ...
    const int32_t *slice_start;  // ... and the first of its output positions
    const int64_t *goff;         // per interval: where its counts start in ghist (low 40 bits) and how many there
                                 // is room for (above them: the HOST's L + 2)
    int32_t per_interval;        // uniform batches (the three above null): slices per interval, interval-major
    int64_t slice_first;         // ... and the slice of this launch's first workgroup
    int32_t *ghist;
    int32_t *gnan;               // per interval: null windows that are not a number
};

template <int NT>
__global__ void __launch_bounds__(NT, 8) k_fdr_slice(const fdr_slice_args sa) {
    extern __shared__ double smem[];
    constexpr int NTMAX = NT, HS = 3;
    const fdr_args &a = sa.a;
    const int n2 = a.n2_max;
    double *par = smem;                                   // 24
    double *skey = par + 24;                              // n2: the interval's sorted thresholds
    double *zq = skey + n2;                               // 2 x NT pairs: z of the pass's four samples
    int *hist = reinterpret_cast<int *>(zq + 4 * NTMAX);  // n2 + 2
    int *misc = hist + n2 + 2;                            // [0] n_nan
    uint16_t *rguide = reinterpret_cast<uint16_t *>(misc + 8);
    const int tid = threadIdx.x;
    const int64_t wg = sa.slice_first + blockIdx.x;
    const int64_t iv = sa.slice_iv ? (int64_t)sa.slice_iv[blockIdx.x] : wg / sa.per_interval;
    if (a.ws_misc[3 * iv + 2] != 0) return;  // left to the full launch
    const int64_t off = a.interval_off ? a.interval_off[iv] : iv * (int64_t)a.interval_len;
    const int L = a.interval_off ? (int)(a.interval_off[iv + 1] - off) : a.interval_len;
    const int s0 = sa.slice_iv ? sa.slice_start[blockIdx.x] : (int)(wg % sa.per_interval) * (NT - 2 * HS);
    int np2 = 1;
    while (np2 < L) np2 <<= 1;
    if (np2 > n2 || s0 >= L || off + L > a.ws_total) return;  // (a stale host copy of the offsets: the full launch's guard reports it)
    const int64_t goff = sa.goff ? sa.goff[iv] : (iv * (int64_t)(L + 2)) | ((int64_t)(L + 2) << 40);
    if ((goff >> 40) < L + 2) return;  // ... or k_fdr_slice_finish: the interval's counts have no room here
    const int dm = a.dm_ids ? a.dm_ids[iv] : 0;
    const double2 *memo = a.memo + (size_t)dm * a.memo_exp * a.memo_obs;
    const uint32_t *alias = a.alias + ((size_t)dm * a.memo_exp << a.alias_lg);
    const double *zt = a.zt + ((size_t)dm * a.memo_exp << a.alias_lg);
    const uint8_t *row_lg = a.row_lg + (size_t)dm * a.memo_exp;
    if (tid < 24) par[tid] = a.model[(size_t)dm * 24 + tid];
    for (int i = tid; i < L; i += NT) skey[i] = a.ws_key[off + i];
    const int m = a.ws_misc[3 * iv], rank_one = a.ws_misc[3 * iv + 1];
    for (int i = tid; i < m + 2; i += NT) hist[i] = 0;
    if (tid < 8) misc[tid] = 0;
    __syncthreads();
    // the rank guide, as in k_fdr_null
    const double kYR = 4.5 * a.sqrt_k;
    const int nb = fdr_guide_slices(np2);
    const double yscale = (double)nb / (2.0 * kYR);
    const double slice_c0 = kYR * yscale, slice_top = (double)(nb - 1);
    auto slice_of = [&](double y) { return (int)fmin(fmax(fma(y, yscale, slice_c0), 0.0), slice_top); };
    for (int i = tid; i <= rank_one; i += NT) {
        const int from = i == 0 ? 0 : slice_of(skey[i - 1]) + 1;
        const int to = i == rank_one ? nb : slice_of(skey[i]);
        for (int b = from; b <= to; ++b) rguide[b] = (uint16_t)i;
    }
    __syncthreads();
    // this lane's position (the first and last HS lanes are the halo) and whether it is an output of the slice
    const int t = s0 - HS + tid;
    const bool valid = t >= 0 && t < L;
    const bool out = tid >= HS && tid < NT - HS && t < L;
    const bool inside = out && t >= HS && t < L - HS;  // its window fits the interval
    const int rl = valid ? alias_row_of(a.exp[off + t], a.memo_exp, row_lg) : -1;
    constexpr bool has_up = false;  // (no test hooks in the light instances: fpt_fdr_dev sends such calls to the full one)
    bool left_out = false;
    double2 *z01 = reinterpret_cast<double2 *>(zq), *z23 = z01 + NTMAX;
    for (int s = 0; s < a.times; s += 4) {
        const int ns = a.times - s < 4 ? a.times - s : 4;
        uint32_t w4[4] = {0u, 0u, 0u, 0u};
        double u4[4], z4[4] = {0.0, 0.0, 0.0, 0.0};
        if (valid) {
            if (!has_up) {
                const uint64_t base = (uint64_t)(a.base_index0 + off + t);
                philox4x32_10((uint32_t)base, (uint32_t)(base >> 32), (uint32_t)(s >> 2), 0x66707464u,
                              (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) u4[j] = fma((double)w4[j], 1.0 / 4294967296.0, 0.5 / 4294967296.0);
            if (has_up) {  // (tests)
                const double *up = a.null_uniform + (size_t)(off + t) * a.times + s;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u4[j] = j < ns ? up[j] : 0.5;
                    w4[j] = fptm::uniform_word(u4[j]);
                }
            }
            left_out |= nb_draw_zn<4, true>(memo, alias, zt, a.memo_obs, a.alias_lg, par, rl, a.exp + off + t, w4, u4, z4);
        }
        z01[tid] = make_double2(z4[0], z4[1]);
        z23[tid] = make_double2(z4[2], z4[3]);
        __syncthreads();
        if (out) {
            double x0 = fptm::kInf, x1 = fptm::kInf, x2 = fptm::kInf, x3 = fptm::kInf;
            if (inside) {
                const double2 *zw01 = z01 + (tid - HS), *zw23 = z23 + (tid - HS);
                const double2 f01 = zw01[0], f23 = zw23[0];
                double sm[4] = {f01.x, f01.y, f23.x, f23.y};
#pragma unroll
                for (int j = 1; j <= 2 * HS; ++j) {
                    const double2 p01 = zw01[j], p23 = zw23[j];
                    sm[0] += p01.x, sm[1] += p01.y, sm[2] += p23.x, sm[3] += p23.y;
                }
                x0 = (sm[0] - sm[0]) - sm[0];
                x1 = (sm[1] - sm[1]) - sm[1];
                x2 = (sm[2] - sm[2]) - sm[2];
                x3 = (sm[3] - sm[3]) - sm[3];
            }
            auto rank_pair = [&](const double y0, const double y1, const bool second) {
                const int b0 = slice_of(y0), b1 = slice_of(y1);
                int l0 = rguide[b0], h0 = rguide[b0 + 1], l1 = rguide[b1], h1 = rguide[b1 + 1];
                for (int it = 0; l0 < h0 || l1 < h1; ++it) {
                    const int m0 = it == 0 ? h0 - 1 : (it == 1 ? l0 : (l0 + h0) >> 1);
                    const int m1 = it == 0 ? h1 - 1 : (it == 1 ? l1 : (l1 + h1) >> 1);
                    const bool g0 = skey[m0] <= y0, g1 = skey[m1] <= y1;
                    if (l0 < h0) {
                        l0 = g0 ? m0 + 1 : l0;
                        h0 = g0 ? h0 : m0;
                    }
                    if (l1 < h1) {
                        l1 = g1 ? m1 + 1 : l1;
                        h1 = g1 ? h1 : m1;
                    }
                }
                atomicAdd(isnan(y0) ? &misc[0] : &hist[l0], 1);
                if (second) atomicAdd(isnan(y1) ? &misc[0] : &hist[l1], 1);
            };
            rank_pair(x0, x1, ns > 1);
            if (ns > 2) rank_pair(x2, x3, ns > 3);
        }
        __syncthreads();
    }
    if (__syncthreads_or(left_out ? 1 : 0)) {  // a draw fell into a row's rest: the whole interval goes to the full launch
        if (tid == 0) a.ws_misc[3 * iv + 2] = 1;
        return;
    }
    int32_t *gh = sa.ghist + (goff & ((1ll << 40) - 1));
    for (int i = tid; i <= m; i += NT) {
        const int v = hist[i];
        if (v) atomicAdd(&gh[i], v);
    }
    if (tid == 0 && misc[0]) atomicAdd(&sa.gnan[iv], misc[0]);
}

// one workgroup per sliced interval: the inclusive prefix of its counts, then the efdr of every position
// (step 3 of k_fdr_null); nothing for an interval that was marked
__global__ void __launch_bounds__(256) k_fdr_slice_finish(const fdr_slice_args sa) {
    extern __shared__ double smem[];
    constexpr int NT = 256;
    const fdr_args &a = sa.a;
    int *hist = reinterpret_cast<int *>(smem);  // n2 + 2
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int64_t iv = a.iv_list ? (int64_t)a.iv_list[blockIdx.x] : a.iv_first + blockIdx.x;
    const int64_t off = a.interval_off ? a.interval_off[iv] : iv * (int64_t)a.interval_len;
    const int L = a.interval_off ? (int)(a.interval_off[iv + 1] - off) : a.interval_len;
    int np2 = 1;
    while (np2 < L) np2 <<= 1;
    if (L <= 0) return;
    if (np2 > a.n2_max || off + L > a.ws_total) {  // (as k_fdr_null's guard: the interval does not fit what the host sized)
        for (int i = tid; i < L; i += NT) a.efdr[off + i] = NAN;
        return;
    }
    if (a.ws_misc[3 * iv + 2] != 0) return;  // left to the full launch
    const int m = a.ws_misc[3 * iv];
    const int64_t goff = sa.goff ? sa.goff[iv] : (iv * (int64_t)(L + 2)) | ((int64_t)(L + 2) << 40);
    const bool room = (goff >> 40) >= L + 2;
    const int32_t *gh = sa.ghist + (goff & ((1ll << 40) - 1));
    for (int i = tid; i <= m; i += NT) hist[i] = room ? gh[i] : 0;
    __syncthreads();
    if (tid < kWave) {
        int carry = 0;
        for (int base = 0; base <= m; base += kWave) {
            const int i = base + lane;
            int v = (i <= m) ? hist[i] : 0;
            v = wave_scan_i32(v) + carry;
            if (i <= m) hist[i] = v;
            carry = __shfl(v, kWave - 1, kWave);
        }
    }
    __syncthreads();
    const int n_finite = hist[m];
    const int n_nan = sa.gnan[iv];
    const double denom = (double)L * (double)a.times;
    // The slices were cut from the HOST copy of the offsets; this length comes from the device.  If the two
    // disagree (a stale host array) the counts are not the interval's L * times: NaN says so, as in k_fdr_null.
    if (!room || (long long)n_finite + n_nan != (long long)L * a.times) {
        for (int i = tid; i < L; i += NT) a.efdr[off + i] = NAN;
        return;
    }
    for (int i = tid; i < L; i += NT) {
        const int pos = (int)a.ws_idx[off + i];
        double f = 1.0;  // NaN observed: the two-pointer walk runs to the end (utils.pyx:76)
        if (i < m) {
            int cnt = hist[i];
            if (cnt == n_finite) cnt += n_nan;  // nothing finite above: the walk passes the NaNs too
            f = (double)cnt / denom;
            if (f > 1.0) f = 1.0;
        }
        a.efdr[off + pos] = f;
    }
}

// size classes of short intervals (64, 128, 192 lanes): fewer idle lanes; 192 lanes take 129..192
// bases in one pass and 257..384 in two
template <int HSC, bool ONE, int MODE = 0>
fdr_kernel_t fdr_kernel(int nt) {
    // (512 lanes: intervals of 513 bases and more, in several rounds -- their buffers let two
    // workgroups live on a CU, and with 256 lanes each those were 8 wavefronts)
    if (!ONE && nt == 512) return k_fdr_null<512, false, HSC, false, MODE>;
    return nt == 64 ? k_fdr_null<64, false, HSC, ONE, MODE> : nt == 128 ? k_fdr_null<128, false, HSC, ONE, MODE>
           : nt == 192 ? k_fdr_null<192, false, HSC, ONE, MODE> : k_fdr_null<256, false, HSC, ONE, MODE>;
}

// ===========================================================================
// k_detect_columns: the record columns of `ftd detect` for a whole batch (cli/detect.py:136-146):
//   stats = column_stack((exp, obs, -log(pvals), -log(win_pvals), efdr)), rows = bases
// and, for an interval whose statistics the reference could not compute (its `except Exception`
// branch, here: the status word of the scan), pvals = win_pvals = efdr = 1.  One lane per base;
// the five values of a base are 40 consecutive bytes of the output.
// ===========================================================================
__global__ void __launch_bounds__(256) k_detect_columns(int64_t n_intervals, int32_t interval_len,
                                                        const int64_t *__restrict__ interval_off,
                                                        const int32_t *__restrict__ status,
                                                        const double *__restrict__ ex, const double *__restrict__ ob,
                                                        const double *__restrict__ pv, const double *__restrict__ wp,
                                                        const double *__restrict__ ef, int64_t total,
                                                        double *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        bool failed = false;
        if (status) {
            int64_t iv;
            if (interval_off) {  // the interval of base g: last offset <= g
                int64_t lo = 0, hi = n_intervals;
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (interval_off[mid] <= g) lo = mid; else hi = mid;
                }
                iv = lo;
            } else {
                iv = g / interval_len;
            }
            failed = status[iv] != 0;
        }
        const double p = failed ? 1.0 : pv[g], w = failed ? 1.0 : wp[g], f = failed ? 1.0 : ef[g];
        double *o = out + g * 5;
        o[0] = ex[g];
        o[1] = ob[g];
        o[2] = -log(p);
        o[3] = -log(w);
        o[4] = f;
    }
}

// ===========================================================================
// k_hist2d: hist[int(exp), int(obs)] += 1 (cli/learn_dm.py:276-287), pairs outside the
// histogram ignored.  The dense low corner (64 x 64 bins) is accumulated per workgroup in LDS,
// the rest goes straight to global atomics.
// ===========================================================================
__global__ void __launch_bounds__(256) k_hist2d(const double *__restrict__ ex, const double *__restrict__ ob,
                                                int64_t n, int rows, int cols,
                                                unsigned long long *__restrict__ hist) {
    constexpr int C = 64;
    __shared__ unsigned int sub[C * C];
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) sub[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double e = ex[i], o = ob[i];
        // int(x) truncates toward zero (-0.5 is bin 0); a negative index of a numpy array counts from the end
        // while it is >= -dim; beyond either end: IndexError -> pass (learn_dm.py:285-287).  Values that are not
        // finite (int() raises on them in the reference, uncaught) are skipped.
        if (!(fabs(e) < 2147483647.0) || !(fabs(o) < 2147483647.0)) continue;
        int r = (int)e, c = (int)o;
        if (r < -rows || r >= rows || c < -cols || c >= cols) continue;
        r += r < 0 ? rows : 0;
        c += c < 0 ? cols : 0;
        if (r < C && c < C) atomicAdd(&sub[r * C + c], 1u);
        else atomicAdd(&hist[(size_t)r * cols + c], 1ull);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const unsigned int v = sub[i];
        const int r = i / C, c = i % C;
        if (v && r < rows && c < cols) atomicAdd(&hist[(size_t)r * cols + c], (unsigned long long)v);
    }
}

// ===========================================================================
// k_segment: utils.segment (stats/utils.pyx:15-50) of every interval of a track, one wavefront
// per interval.  64 positions at a time are turned into two ballot masks (passing / failing the
// threshold; NaN is in neither) and the open/close state machine runs on the masks in scalar
// code, so its cost follows the number of runs, not of bases.  FILL = false counts the merged
// segments, FILL = true writes them at the offsets computed from the counts.
// ===========================================================================
struct seg_args {
    int64_t n_intervals;
    int32_t interval_len;
    const int64_t *interval_off;
    const double *x;
    double thr;
    int32_t w, decreasing;
    int32_t *counts;
    const int64_t *offsets;
    int32_t *seg_iv, *seg_start, *seg_end;
    double *seg_score;
};

template <bool FILL>
__global__ void __launch_bounds__(256) k_segment(const seg_args a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t iv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iv >= a.n_intervals) return;
    int64_t off;
    int L;
    if (a.interval_off) {
        off = a.interval_off[iv];
        L = (int)(a.interval_off[iv + 1] - off);
    } else {
        L = a.interval_len;
        off = iv * (int64_t)L;
    }
    const double d = a.decreasing ? -1.0 : 1.0, thr = d * a.thr;
    bool open = false, have = false;
    int cs = 0, ps = 0, pe = 0, n_out = 0;
    const int64_t o = FILL ? a.offsets[iv] : 0;
    auto emit = [&](int s, int e) {
        if (FILL) {
            const int lo = s < 0 ? 0 : s, hi = e > L ? L : e;
            double m = fptm::kInf;
            bool bad = lo >= hi;  // empty slice: numpy's min raises, here the score is NaN
            for (int j = lo + lane; j < hi; j += kWave) {
                const double v = a.x[off + j];
                bad |= isnan(v);
                m = fmin(m, v);
            }
            for (int sft = kWave / 2; sft > 0; sft >>= 1) m = fmin(m, __shfl_xor(m, sft, kWave));
            if (__ballot(bad)) m = NAN;
            if (lane == 0) {
                a.seg_iv[o + n_out] = (int32_t)iv;
                a.seg_start[o + n_out] = s;
                a.seg_end[o + n_out] = e;
                a.seg_score[o + n_out] = m;
            }
        }
        ++n_out;
    };
    for (int base = 0; base < L; base += kWave) {
        const int i = base + lane;
        const double dv = (i < L) ? d * a.x[off + i] : NAN;
        const unsigned long long P = __ballot(dv >= thr), F = __ballot(dv < thr);
        unsigned long long rem = ~0ull;  // positions of this block not consumed yet
        // the reference keeps "no open run" as curr_start < 0, so a passing element whose
        // curr_start = i - w + 1 would be negative does not open one: only positions >= w - 1 can
        const int first_ok = a.w - 1 - base;
        const unsigned long long can_open = first_ok <= 0 ? ~0ull : (first_ok >= 64 ? 0ull : (~0ull << first_ok));
        for (;;) {
            const unsigned long long m = (open ? F : (P & can_open)) & rem;
            if (!m) break;
            const int b = __ffsll((long long)m) - 1;
            rem = b == 63 ? 0ull : (~0ull << (b + 1));
            if (!open) {
                cs = base + b - a.w + 1;
                open = true;
            } else {
                const int e = base + b - 1 + a.w;
                open = false;
                if (have && cs <= pe) {
                    pe = e;  // overlaps the previous segment: extend it
                } else {
                    if (have) emit(ps, pe);
                    ps = cs;
                    pe = e;
                    have = true;
                }
            }
        }
    }
    if (have) emit(ps, pe);
    if (!FILL && lane == 0) a.counts[iv] = n_out;
}

template __global__ void k_segment<false>(const seg_args);
template __global__ void k_segment<true>(const seg_args);

// ===========================================================================
// synthetic workload + checksum
// ===========================================================================
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) k_synth_counts(uint64_t key, int64_t pos0, int64_t n,
                                                      double *__restrict__ out) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = splitmix64(key + (uint64_t)(pos0 + i));
        out[i] = (double)((h >> 33) % 20u);
    }
}

// Hotspot bursts on top of the uniform counts (heavy-tailed workload: real DNase data has
// hotspots with counts in the hundreds): interval iv = position / padded_len carries one with
// probability per_mille / 1000 -- a triangular bump of 80..199 positions and peak 100..499 per
// strand, plus 0..15 of noise -- decided by a hash of (seed, iv), so every strand, every rank and
// the CPU checker see the same hotspots.
__global__ void __launch_bounds__(256) k_synth_hotspots(uint64_t key_iv, uint64_t key_noise, int64_t pos0, int64_t n,
                                                        int32_t padded_len, int32_t per_mille,
                                                        double *__restrict__ counts) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t p = pos0 + i;
        const int64_t iv = p / padded_len;
        const int u = (int)(p - iv * padded_len);
        const uint64_t hv = splitmix64(key_iv + (uint64_t)iv);
        if ((int)(hv % 1000u) >= per_mille) continue;
        const int width = 80 + (int)((hv >> 40) % 120u), half = width / 2;
        const int span = padded_len - 2 * half > 1 ? padded_len - 2 * half : 1;
        const int centre = half + (int)((hv >> 20) % (uint64_t)span);
        const int dist = u > centre ? u - centre : centre - u;
        if (dist >= half) continue;
        const int peak = 100 + (int)((hv >> 10) % 400u);
        const uint64_t hn = splitmix64(key_noise + (uint64_t)p);
        counts[i] += (double)((peak * (half - dist)) / half + (int)((hn >> 7) & 15u));
    }
}

__global__ void __launch_bounds__(256) k_synth_bases(uint64_t key, int64_t pos0, int64_t n,
                                                     uint8_t *__restrict__ out) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = splitmix64(key + (uint64_t)(pos0 + i));
        const uint32_t acgt = 0x54474341u;  // "ACGT" little endian
        out[i] = (uint8_t)(acgt >> (8 * ((h >> 13) & 3u)));
    }
}

__global__ void __launch_bounds__(256) k_checksum(const double *__restrict__ x, int64_t n,
                                                  unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        acc += (unsigned long long)__double_as_longlong(x[i]);
    for (int d = kWave / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(out, acc);
}

// ===========================================================================
// launchers (called from fpt_capi.cpp through fpt_kernels.hpp)
// ===========================================================================
namespace fptk {

static inline int grid_for(int64_t n, int block, int cap) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

void launch_kmer_probs(hipStream_t st, const uint8_t *seq, int64_t n_out, const double *table,
                       double *fwd, double *rev) {
    if (n_out <= 0) return;
    hipLaunchKernelGGL(k_kmer_probs, dim3(grid_for(n_out, 256, 2048)), dim3(256), 0, st, seq, n_out,
                       table, fwd, rev);
}

void launch_predict_rows(hipStream_t st, const double *obs, const double *probs, int64_t n_rows,
                         int l, int hw, int shw, int k_trim, double *exp_out, double *win_out) {
    if (n_rows <= 0 || l <= 0) return;
    const int tile_len = 1024;
    int tiles = (l + tile_len - 1) / tile_len;
    // scratch windows for the re-evaluation in the reference's order (k_predict_rows): 16, fewer for
    // very wide smoothing windows
    const int w = 2 * shw + 1;
    int n_slots = 16;
    while (n_slots > 1 && (size_t)(tile_len + 2 * shw + (size_t)n_slots * w + 2) * sizeof(double) > 96 * 1024) n_slots >>= 1;
    size_t lds = (size_t)(tile_len + 2 * shw + (size_t)n_slots * w + 2) * sizeof(double);
    (void)hipFuncSetAttribute((const void *)k_predict_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // grid.y is limited to 65535: loop over row chunks
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
        int64_t nr = n_rows - r0 < 65535 ? n_rows - r0 : 65535;
        hipLaunchKernelGGL(k_predict_rows, dim3(tiles, (unsigned)nr), dim3(256), lds, st,
                           obs + r0 * l, probs + r0 * l, l, hw, shw, k_trim, tile_len, n_slots,
                           exp_out + r0 * l, win_out + r0 * l);
    }
}

void launch_nb_values(hipStream_t st, int what, const double *model, const double *ex,
                      const double *ob, int64_t n, double *out, int *flags) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_nb_values, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, what, model,
                       ex, ob, n, out, flags);
}

void launch_nb_scalar(hipStream_t st, int what, const int32_t *k, const double *p, const double *r,
                      int64_t n, double *out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_nb_scalar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, what, k, p,
                       r, n, out);
}

void launch_special(hipStream_t st, int fn, const double *a, const double *b, const double *x,
                    int64_t n, double *out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_special, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, fn, a, b, x,
                       n, out);
}

void launch_window_rows(hipStream_t st, int op, const double *x, const double *w, int64_t n_rows,
                        int n, int hw, double *out) {
    if (n_rows <= 0 || n <= 0) return;
    const int tile_len = 1024;
    int tiles = (n + tile_len - 1) / tile_len;
    size_t lds = (size_t)(tile_len + 2 * hw) * sizeof(double) * (op == 4 ? 2 : 1);
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
        int64_t nr = n_rows - r0 < 65535 ? n_rows - r0 : 65535;
        hipLaunchKernelGGL(k_window_rows, dim3(tiles, (unsigned)nr), dim3(256), lds, st, op,
                           x + r0 * n, w ? w + r0 * n : nullptr, n, hw, tile_len, out + r0 * n);
    }
}

size_t scan_lds_bytes(int nc_max, bool tblg, bool memo_only) {
    if (memo_only) return (size_t)(8 * (size_t)nc_max) * sizeof(double);
    return (size_t)((tblg ? 0 : kTable + 2) + 26 + 8 * (size_t)nc_max) * sizeof(double) + (size_t)nc_max + 16;
}

typedef void (*scan_kernel_t)(const scan_args);

template <bool TBLG, bool MO, bool REDO>
static scan_kernel_t scan_kernel_t_(int nt, bool dflt) {
    switch (nt) {
    case 256: return dflt ? k_scan_fused<256, 5, 50, TBLG, MO, REDO> : k_scan_fused<256, 0, 0, TBLG, MO, REDO>;
    case 512: return dflt ? k_scan_fused<512, 5, 50, TBLG, MO, REDO> : k_scan_fused<512, 0, 0, TBLG, MO, REDO>;
    default: return dflt ? k_scan_fused<1024, 5, 50, TBLG, MO, REDO> : k_scan_fused<1024, 0, 0, TBLG, MO, REDO>;
    }
}

// second_pass: the full instance over the tiles flagged by the memo-only pass (table through L2 only)
static scan_kernel_t scan_kernel(int nt, int hw, int shw, bool tblg, bool memo_only, bool second_pass) {
    const bool dflt = (hw == 5 && shw == 50);
    if (memo_only) return scan_kernel_t_<true, true, false>(nt, dflt);
    if (second_pass) return scan_kernel_t_<true, false, true>(nt, dflt);
    return tblg ? scan_kernel_t_<true, false, false>(nt, dflt) : scan_kernel_t_<false, false, false>(nt, dflt);
}

hipError_t scan_set_lds(int nt, int hw, int shw, bool tblg, bool memo_only, bool second_pass, size_t lds) {
    return hipFuncSetAttribute((const void *)scan_kernel(nt, hw, shw, tblg, memo_only, second_pass),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

void launch_scan(hipStream_t st, int nt, int grid, size_t lds, const scan_launch &sl, bool memo_only) {
    scan_args a;
    a.n_intervals = sl.n_intervals;
    a.interval_len = sl.interval_len;
    a.interval_off = sl.interval_off;
    a.tile_iv = sl.tile_iv;
    a.tile_t0 = sl.tile_t0;
    a.tile_tl = sl.tile_tl;
    a.tile_first = sl.tile_first;
    a.tiles_per_interval = sl.tiles_per_interval;
    a.tile_len = sl.tile_len;
    a.hw = sl.hw;
    a.shw = sl.shw;
    a.k_trim = sl.k_trim;
    a.n_scales = sl.n_scales;
    a.max_scale = 0;
    a.min_scale = 1 << 30;
    for (int i = 0; i < sl.n_scales; ++i) a.min_scale = sl.scales[i] < a.min_scale ? sl.scales[i] : a.min_scale;
    for (int i = 0; i < FPT_MAX_SCALES; ++i) {
        a.scales[i] = i < sl.n_scales ? sl.scales[i] : 0;
        // 1/sqrt(K): the reference divides by sqrt(K) (windowing.h:64); multiplying by the
        // reciprocal moves z by an ulp, far inside the 1e-6 contract on the window p-value
        a.scale_sqrt[i] = i < sl.n_scales ? 1.0 / sqrt((double)(2 * sl.scales[i] + 1)) : 1.0;
        if (i < sl.n_scales && sl.scales[i] > a.max_scale) a.max_scale = sl.scales[i];
    }
    a.nc_max = sl.nc_max;
    a.total_bases = sl.total_bases;
    a.counts_plus = sl.counts_plus;
    a.counts_minus = sl.counts_minus;
    a.seq = sl.seq;
    a.table = sl.table;
    a.model = sl.model;
    a.exp_out = sl.exp_out;
    a.obs_out = sl.obs_out;
    a.pval_out = sl.pval_out;
    a.winp_out = sl.winp_out;
    a.status_out = sl.status_out;
    a.memo = (const double2 *)sl.memo;
    a.memo_exp = sl.memo_exp;
    a.memo_obs = sl.memo_obs;
    a.ablate = sl.ablate;
    a.counts_only = sl.counts_only;
    a.redo = sl.redo;
    a.dm_ids = sl.dm_ids;
    a.memo2 = (const double2 *)sl.memo2;
    a.memo2_max = sl.memo2_max;
    a.memo2_have = sl.memo2_have;
    a.memo2_rows = sl.memo2_rows;
    a.memo2_stride = sl.memo2_stride;
    a.fast_trim = (sl.k_trim == 1 && sl.shw >= 32 && sl.nc_max <= 3 * nt) ? 1 : 0;
    const bool second_pass = sl.redo && !memo_only;
    a.redo_list = nullptr;
    a.redo_cursor = nullptr;
    void (*kern)(const scan_args) = scan_kernel(nt, sl.hw, sl.shw, sl.table_global != 0, memo_only, second_pass);
    if (second_pass) {  // the flagged tiles as a list, then as many workgroups as are resident at once
        a.redo_list = sl.redo_list + sl.tile_first;
        a.redo_cursor = sl.redo_cursor;
        if (sl.redo_cursor_clear) (void)hipMemsetAsync(sl.redo_cursor, 0, 2 * sizeof(int32_t), st);
        hipLaunchKernelGGL(k_redo_compact, dim3((grid + 255) / 256), dim3(256), 0, st, sl.redo + sl.tile_first,
                           (int64_t)grid, sl.redo_list + sl.tile_first, sl.redo_cursor);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, nt, lds) != hipSuccess || per_cu < 1)
            per_cu = 1;
        const int64_t resident = (int64_t)per_cu * (sl.n_cu > 0 ? sl.n_cu : 256);
        if (grid > resident) grid = (int)resident;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, st, a);
}

void launch_nb_memo(hipStream_t st, const double *models, int n_models, int memo_exp, int memo_obs,
                    void *memo, int32_t *clear, int64_t n_clear, int32_t *state, int32_t *have, int have_state,
                    int rows, int stride) {
    int n = memo_exp * memo_obs;
    hipLaunchKernelGGL(k_nb_memo, dim3((n + 255) / 256, n_models), dim3(256), 0, st, models, memo_exp,
                       memo_obs, (double2 *)memo, clear, n_clear, state, have, have_state, rows, stride);
}

void launch_nb_memo2(hipStream_t st, const double *models, int n_models, const int32_t *miss_max, const int32_t *have,
                     int memo_exp, int memo_obs, int rows, int stride, void *memo2) {
    hipLaunchKernelGGL(k_nb_memo2, dim3(1024, n_models), dim3(256), 0, st, models, miss_max, have, memo_exp, memo_obs,
                       rows, stride, (double2 *)memo2);
}

// ---- the tile table of a ragged batch, made on the device (fpt_scan_dev).  One lane per interval,
// kPlanBlock intervals per workgroup: an interval of up to 1,024 bases is one tile, a longer one is
// cut every split_len bases (each piece read with a halo of H); a tile goes to the workgroup-size
// class that holds it.  The table is class-major, intervals in their order inside a class: the host
// has counted the tiles of every class (it needs them for the grid sizes) and, on the way, where each
// workgroup's intervals start inside every class (`block_base`, kLeanClasses ints per workgroup); the
// places inside a workgroup come from seven small prefix sums.  (The table used to be made on the
// host -- 27 growing vectors, 19 MB of pageable copies and a wait: 4.4 ms per call on the
// whole-genome shape, whose kernels take 1.8, whenever a call's offsets differed from the last one's.
// One atomic per class and wavefront instead of the prefix sums: 0.44 ms of contention.)
struct plan_args {
    const int64_t *off;
    int64_t n_intervals, n_tiles;
    int32_t H, split_len;
    const int32_t *block_base;
    int32_t *tile_iv, *tile_t0, *tile_tl;
    lean_tile_rec *recs;
    int32_t lmax[kLeanClasses];
    int32_t first_split;
};
__device__ __forceinline__ int plan_class_of(const plan_args &a, int n) {  // the first class that holds n bases
    int k = 0;
#pragma unroll
    for (int i = 0; i < kLeanClasses - 1; ++i) k += a.lmax[i] < n ? 1 : 0;
    return k;
}
__global__ void __launch_bounds__(kPlanBlock) k_plan_tiles(const plan_args a) {
    __shared__ int wave_total[kPlanBlock / 64][kLeanClasses];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * kPlanBlock + tid;
    int L = 0;
    int64_t o = 0;
    if (i < a.n_intervals) {
        o = a.off[i];
        L = (int)(a.off[i + 1] - o);
    }
    // this interval's tiles: n_full pieces of split_len bases (the top class: split_len + H > 768
    // positions with their halo) and a last -- or only -- one
    constexpr int kTop = kLeanClasses - 1;
    int n_full = 0, last_t0 = 0, last_cls = -1;
    if (L > 1024) {
        n_full = (L - 1) / a.split_len;
        last_t0 = n_full * a.split_len;
        last_cls = max(plan_class_of(a, L - last_t0 + a.H), a.first_split);  // (a piece is no whole interval)
    } else if (L > 0) {
        last_cls = plan_class_of(a, L);
    }
    int mine[kLeanClasses];  // where this lane's tiles of a class start
#pragma unroll
    for (int c = 0; c < kLeanClasses; ++c) {
        const int v = (c == kTop ? n_full : 0) + (last_cls == c ? 1 : 0);
        const int incl = wave_scan_i32(v);
        if (lane == 63) wave_total[wave][c] = incl;
        mine[c] = incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kLeanClasses; ++c) {
        int before = a.block_base[(int64_t)blockIdx.x * kLeanClasses + c];
        for (int w = 0; w < wave; ++w) before += wave_total[w][c];
        mine[c] += before;
    }
    auto put = [&](int slot, int t0, int tl) {
        a.tile_iv[slot] = (int32_t)i;
        a.tile_t0[slot] = t0;
        a.tile_tl[slot] = tl;
        a.recs[slot] = lean_tile_rec{o, (int32_t)i, t0, tl, L, {0, 0}};
    };
    for (int p = 0; p < n_full; ++p) put(mine[kTop] + p, p * a.split_len, a.split_len);
    if (last_cls >= 0) {
        int slot = mine[0];
#pragma unroll
        for (int c = 1; c < kLeanClasses; ++c) slot = last_cls == c ? mine[c] : slot;
        put(slot + (last_cls == kTop ? n_full : 0), last_t0, L - last_t0);
    }
}

lean_class_set make_lean_classes() {
    lean_class_set c{};
    for (int i = 0; i < kLeanClasses; ++i) c.lmax[i] = c.nt[i] = kLeanNT[i];
    c.first_split = 0;
    return c;
}

void launch_plan_tiles(hipStream_t st, const int64_t *off, int64_t n_intervals, int64_t n_tiles, int H, int split_len,
                       const lean_class_set &cls, const int32_t *block_base, int32_t *flat, void *recs) {
    plan_args a;
    for (int i = 0; i < kLeanClasses; ++i) a.lmax[i] = cls.lmax[i];
    a.first_split = cls.first_split;
    a.off = off;
    a.n_intervals = n_intervals;
    a.n_tiles = n_tiles;
    a.H = H;
    a.split_len = split_len;
    a.block_base = block_base;
    a.tile_iv = flat;
    a.tile_t0 = flat + n_tiles;
    a.tile_tl = flat + 2 * n_tiles;
    a.recs = (lean_tile_rec *)recs;
    hipLaunchKernelGGL(k_plan_tiles, dim3((unsigned)((n_intervals + kPlanBlock - 1) / kPlanBlock)), dim3(kPlanBlock), 0, st, a);
}

// alias tables, z tables and widths of n_models x memo_exp rows: `tables` holds nb_alias_bytes() -- the
// entries, then the z (at nb_alias_z_offset), then a byte per row
static size_t alias_rows(int n_models, int memo_exp) { return (size_t)n_models * memo_exp; }
size_t nb_alias_z_offset(int n_models, int memo_exp, int memo_obs) {
    return ((alias_rows(n_models, memo_exp) << alias_lg_of(memo_obs)) * 4 + 255) & ~(size_t)255;
}
static size_t nb_alias_lg_offset(int n_models, int memo_exp, int memo_obs) {
    return nb_alias_z_offset(n_models, memo_exp, memo_obs) + (alias_rows(n_models, memo_exp) << alias_lg_of(memo_obs)) * 8;
}
size_t nb_alias_bytes(int n_models, int memo_exp, int memo_obs) {
    return nb_alias_lg_offset(n_models, memo_exp, memo_obs) + alias_rows(n_models, memo_exp);
}
void launch_nb_alias(hipStream_t st, const void *memo, int n_models, int memo_exp, int memo_obs, void *tables) {
    const int lg = alias_lg_of(memo_obs);
    const size_t n = (size_t)1 << lg;
    char *t = (char *)tables;
    hipLaunchKernelGGL(k_nb_alias, dim3(memo_exp, n_models), dim3(64), n * 8 + 3 * n * 2, st, (const double2 *)memo,
                       memo_exp, memo_obs, lg, (uint32_t *)t, (double *)(t + nb_alias_z_offset(n_models, memo_exp, memo_obs)),
                       (uint8_t *)(t + nb_alias_lg_offset(n_models, memo_exp, memo_obs)));
}

size_t fdr_slice_lds_bytes(int n2, int lanes) {
    const size_t guide = ((size_t)fdr_guide_slices(n2) + 2) * sizeof(uint16_t);
    return (size_t)(24 + (size_t)n2 + 4 * (size_t)lanes) * sizeof(double) + (size_t)((size_t)n2 + 2 + 8) * sizeof(int) +
           ((guide + 7) & ~(size_t)7);
}
// (uniform: every interval of the batch has L bases and the lanes are chosen for it; ragged batches: 192)
int fdr_slice_positions_of(int L, bool uniform) { return (uniform ? fdr_slice_lanes(L) : kFdrSliceLanes) - 6; }
int fdr_slices_of(int L, bool uniform) { return (L + fdr_slice_positions_of(L, uniform) - 1) / fdr_slice_positions_of(L, uniform); }

size_t fdr_lds_bytes(int n2, bool dbuf, bool global_buffers) {
    const size_t guide = ((size_t)fdr_guide_slices(n2) + 2) * (global_buffers ? sizeof(int) : sizeof(uint16_t));
    return (size_t)(24 + (dbuf ? 9 : 5) * (size_t)n2) * sizeof(double) + (size_t)(3 * (size_t)n2 + 2 + 8) * sizeof(int) +
           ((guide + 7) & ~(size_t)7);
}
// the set-up launch (MODE 1): par, skey, two n2 of zb (the z of the observed counts, the keys), then sidx, nf,
// hist and misc -- a third of the draw launch's buffers
size_t fdr_setup_lds_bytes(int n2) {
    return (size_t)(24 + 3 * (size_t)n2) * sizeof(double) + (size_t)(3 * (size_t)n2 + 2 + 8) * sizeof(int);
}

hipError_t launch_fdr(hipStream_t st, const fdr_launch &fl) {
    fdr_args a;
    a.n_intervals = fl.n_intervals;
    a.interval_len = fl.interval_len;
    a.interval_off = fl.interval_off;
    a.base_index0 = fl.base_index0;
    a.hw = fl.hw;
    a.times = fl.times;
    a.seed = fl.seed;
    a.model = fl.model;
    a.memo = (const double2 *)fl.memo;
    a.alias = (const uint32_t *)fl.alias;
    a.zt = (const double *)((const char *)fl.alias + nb_alias_z_offset(fl.n_models, fl.memo_exp, fl.memo_obs));
    a.row_lg = (const uint8_t *)fl.alias + nb_alias_lg_offset(fl.n_models, fl.memo_exp, fl.memo_obs);
    a.alias_lg = alias_lg_of(fl.memo_obs);
    a.memo_exp = fl.memo_exp;
    a.memo_obs = fl.memo_obs;
    a.exp = fl.exp;
    a.winp = fl.winp;
    a.obs = fl.obs;
    a.efdr = fl.efdr;
    a.null_uniform = fl.null_uniform;
    a.null_out = fl.null_out;
    a.dm_ids = fl.dm_ids;
    a.ablate = fl.ablate;
    a.n2_max = fl.n2_max;
    a.sqrt_k = sqrt((double)(2 * fl.hw + 1));
    a.inv_sqrt_k = 1.0 / a.sqrt_k;
    a.iv_list = fl.iv_list;
    a.dbuf = 0;
    a.gws = (char *)fl.gws;
    a.gws_stride = fl.gws_stride;
    const int64_t n_blocks = fl.iv_list ? fl.n_list : fl.n_intervals;
    if (fl.gws) {  // long intervals: buffers in global memory, as many workgroups at a time as fit
        const int64_t per = fl.gws_blocks < 1 ? 1 : fl.gws_blocks;
        for (int64_t done = 0; done < n_blocks; done += per) {
            fdr_args b = a;
            if (b.iv_list) b.iv_list += done; else b.iv_first = done;
            const int64_t n = n_blocks - done < per ? n_blocks - done : per;
            if (fl.hw == 3) hipLaunchKernelGGL((k_fdr_null<256, true, 3, false>), dim3((unsigned)n), dim3(256), 0, st, b);
            else hipLaunchKernelGGL((k_fdr_null<256, true, 0, false>), dim3((unsigned)n), dim3(256), 0, st, b);
        }
        return hipSuccess;
    }
    // the second pair of z buffers only where four workgroups still fit a CU's LDS
    a.dbuf = fdr_lds_bytes(fl.n2_max, true) <= 40 * 1024 ? 1 : 0;
#ifdef FPT_ABLATE
    if (const char *e = getenv("FPT_FDR_DBUF")) a.dbuf = atoi(e) && a.dbuf;
#endif
    size_t lds = fdr_lds_bytes(fl.n2_max, a.dbuf != 0);
    // one lane per base and null track: intervals of up to 64 / 128 bases get workgroups of that size
    const int nt = fl.nt ? fl.nt : (fl.n2_max <= 64 ? 64 : (fl.n2_max <= 128 ? 128 : 256));
    // (the longest interval of the launch is known to be <= n2_max only; ONE when that says enough
    // or the caller does: max_len)
    const bool one = nt <= 256 && (fl.max_len > 0 ? fl.max_len : fl.n2_max) <= nt;  // (512 lanes: the several-rounds form only)
    // two launches -- the per-interval set-up, then the draws -- where the caller gave a workspace for the
    // hand-over (the `detect` window width; the other widths and the long intervals keep the single launch)
    const bool split = fl.ws_key && fl.ws_idx && fl.ws_misc && fl.hw == 3;
    a.ws_key = fl.ws_key;
    a.ws_idx = fl.ws_idx;
    a.ws_misc = fl.ws_misc;
    a.ws_total = fl.ws_total;
    // split: set-up (1), light draws (3), full draws of what the light ones left (2); without `light`: 1, 2
    const bool light = fl.light;
    const int modes_split[3] = {1, light ? 3 : 2, 2}, n_modes = split ? (light ? 3 : 2) : 1;
    const bool sliced = split && light && fl.n_slices > 0 && fl.ghist && (fl.interval_off ? fl.slice_iv != nullptr : fl.interval_len > 0);
    for (int mi = 0; mi < n_modes; ++mi) {
        const int mode = split ? modes_split[mi] : 0;
        if (mode == 3 && sliced) {  // the light draws of this class as slices, then the counts -> efdr per interval
            fdr_slice_args sa;
            sa.a = a;
            sa.slice_iv = fl.slice_iv;
            sa.slice_start = fl.slice_start;
            sa.goff = fl.goff;
            sa.ghist = fl.ghist;
            sa.gnan = fl.gnan;
            sa.per_interval = fl.interval_off ? 0 : fdr_slices_of(fl.interval_len, true);
            sa.slice_first = 0;
            const int lanes = fl.interval_off ? kFdrSliceLanes : fdr_slice_lanes(fl.interval_len);
            void (*kslice)(const fdr_slice_args) = lanes == 128 ? k_fdr_slice<128> : (lanes == 256 ? k_fdr_slice<256> : k_fdr_slice<192>);
            const size_t lds_s = fdr_slice_lds_bytes(fl.n2_max, lanes), lds_f = ((size_t)fl.n2_max + 2) * sizeof(int);
            hipError_t e = hipFuncSetAttribute((const void *)kslice, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
            if (e != hipSuccess) return e;
            for (int64_t done = 0; done < fl.n_slices; done += 0x7fffff00) {
                fdr_slice_args b = sa;
                if (b.slice_iv) {
                    b.slice_iv += done;
                    b.slice_start += done;
                }
                b.slice_first = done;
                const int64_t n = fl.n_slices - done < 0x7fffff00 ? fl.n_slices - done : 0x7fffff00;
                hipLaunchKernelGGL(kslice, dim3((unsigned)n), dim3(lanes), lds_s, st, b);
            }
            for (int64_t done = 0; done < n_blocks; done += 0x7fffff00) {
                fdr_slice_args b = sa;
                if (b.a.iv_list) b.a.iv_list += done; else b.a.iv_first = done;
                const int64_t n = n_blocks - done < 0x7fffff00 ? n_blocks - done : 0x7fffff00;
                hipLaunchKernelGGL(k_fdr_slice_finish, dim3((unsigned)n), dim3(256), lds_f, st, b);
            }
            continue;
        }
        fdr_kernel_t kern = fl.hw == 3 ? (mode == 1 ? (one ? fdr_kernel<3, true, 1>(nt) : fdr_kernel<3, false, 1>(nt))
                                          : mode == 2 ? (one ? fdr_kernel<3, true, 2>(nt) : fdr_kernel<3, false, 2>(nt))
                                          : mode == 3 ? (one ? fdr_kernel<3, true, 3>(nt) : fdr_kernel<3, false, 3>(nt))
                                                      : (one ? fdr_kernel<3, true>(nt) : fdr_kernel<3, false>(nt)))
                                       : (one ? fdr_kernel<0, true>(nt) : fdr_kernel<0, false>(nt));
        fdr_args am = a;
        size_t lds_m = lds;
        if (mode == 1) {
            am.dbuf = 0;
            lds_m = fdr_setup_lds_bytes(fl.n2_max);
        }
        am.redo_only = (mode == 2 && light) ? 1 : 0;
        if (mode == 3) {
            // (eight wavefronts per SIMD cover a second barrier per pass; the second set of z buffers would
            // cost a third of them: 8.3 against 8.5 ms per 100-draw call)
            am.dbuf = am.dbuf && fl.light_dbuf;
            lds_m = fdr_lds_bytes(fl.n2_max, am.dbuf != 0);
        }
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
        if (e != hipSuccess) return e;
        for (int64_t done = 0; done < n_blocks; done += 0x7fffff00) {
            fdr_args b = am;
            if (b.iv_list) b.iv_list += done; else b.iv_first = done;
            const int64_t n = n_blocks - done < 0x7fffff00 ? n_blocks - done : 0x7fffff00;
            hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(nt), lds_m, st, b);
        }
    }
#ifdef FPT_FDR_MARKS
    if (getenv("FPT_FDR_PHASES")) {
        unsigned long long h[16], z[16] = {};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fdr_phase), sizeof h);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fdr_phase), z, sizeof z);
        fprintf(stderr, "fdr phases nt=%d n=%lld (100 MHz ticks per workgroup): obs-z %.1f keys %.1f sort %.1f thresholds %.1f guide %.1f passes %.1f out %.1f\n",
                nt, (long long)n_blocks, h[0] / (double)n_blocks, h[1] / (double)n_blocks, h[2] / (double)n_blocks,
                h[3] / (double)n_blocks, h[4] / (double)n_blocks, (h[5] + h[7] + h[8] + h[9]) / (double)n_blocks, h[6] / (double)n_blocks);
        fprintf(stderr, "    of the passes: draws %.1f barrier %.1f windows+ranks %.1f\n", h[7] / (double)n_blocks,
                h[8] / (double)n_blocks, h[9] / (double)n_blocks);
    }
#endif
    return hipSuccess;
}

void launch_segment(hipStream_t st, const segment_launch &sl, bool fill) {
    seg_args a;
    a.n_intervals = sl.n_intervals;
    a.interval_len = sl.interval_len;
    a.interval_off = sl.interval_off;
    a.x = sl.track;
    a.thr = sl.threshold;
    a.w = sl.w;
    a.decreasing = sl.decreasing;
    a.counts = sl.counts;
    a.offsets = sl.offsets;
    a.seg_iv = sl.seg_iv;
    a.seg_start = sl.seg_start;
    a.seg_end = sl.seg_end;
    a.seg_score = sl.seg_score;
    const unsigned grid = (unsigned)((sl.n_intervals + 3) / 4);
    if (fill) hipLaunchKernelGGL(k_segment<true>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_segment<false>, dim3(grid), dim3(256), 0, st, a);
}

void launch_detect_columns(hipStream_t st, int64_t n_intervals, int32_t interval_len, const int64_t *interval_off,
                           const int32_t *status, const double *ex, const double *ob, const double *pv,
                           const double *wp, const double *ef, int64_t total, double *out) {
    if (total <= 0) return;
    hipLaunchKernelGGL(k_detect_columns, dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, n_intervals, interval_len,
                       interval_off, status, ex, ob, pv, wp, ef, total, out);
}

void launch_hist2d(hipStream_t st, const double *ex, const double *ob, int64_t n, int rows, int cols,
                   unsigned long long *hist) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_hist2d, dim3(grid_for(n, 256 * 16, 2048)), dim3(256), 0, st, ex, ob, n, rows, cols, hist);
}

void launch_synth(hipStream_t st, uint64_t seed, int64_t pos0_counts, int64_t n_counts,
                  double *counts_plus, double *counts_minus, int64_t pos0_seq, int64_t n_seq,
                  uint8_t *seq) {
    // host-side splitmix64 of (seed + stream) gives the per-stream key
    auto mix = [](uint64_t x) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    if (counts_plus && n_counts > 0)
        hipLaunchKernelGGL(k_synth_counts, dim3(grid_for(n_counts, 256, 4096)), dim3(256), 0, st,
                           mix(seed + 0), pos0_counts, n_counts, counts_plus);
    if (counts_minus && n_counts > 0)
        hipLaunchKernelGGL(k_synth_counts, dim3(grid_for(n_counts, 256, 4096)), dim3(256), 0, st,
                           mix(seed + 1), pos0_counts, n_counts, counts_minus);
    if (seq && n_seq > 0)
        hipLaunchKernelGGL(k_synth_bases, dim3(grid_for(n_seq, 256, 4096)), dim3(256), 0, st,
                           mix(seed + 2), pos0_seq, n_seq, seq);
}

void launch_synth_hotspots(hipStream_t st, uint64_t seed, int64_t pos0, int64_t n, int padded_len, int per_mille,
                           double *counts_plus, double *counts_minus) {
    auto mix = [](uint64_t x) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    if (n <= 0) return;
    if (counts_plus)
        hipLaunchKernelGGL(k_synth_hotspots, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, mix(seed + 7),
                           mix(seed + 8), pos0, n, padded_len, per_mille, counts_plus);
    if (counts_minus)
        hipLaunchKernelGGL(k_synth_hotspots, dim3(grid_for(n, 256, 4096)), dim3(256), 0, st, mix(seed + 7),
                           mix(seed + 9), pos0, n, padded_len, per_mille, counts_minus);
}

void launch_checksum(hipStream_t st, const double *x, int64_t n, unsigned long long *out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_checksum, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, x, n, out);
}

}  // namespace fptk
