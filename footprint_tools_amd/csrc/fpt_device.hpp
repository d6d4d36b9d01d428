// fpt_device.hpp -- device-side building blocks shared by the standalone kernels
// and the fused scan kernel (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fpt_math.hpp"

namespace fptd {

constexpr int kWave = 64;
constexpr int kTable = 4096;  // 6-mer table entries; slot kTable holds the default value

// ---- 6-mer index (reference: modeling/bias.py:101-111, modeling/predict.pyx:47-61,150-153)

// 2-bit code of an ASCII base, case-insensitive like str.upper() (predict.pyx:140); 4 = other.
__device__ __forceinline__ int base_code(uint8_t ch) {
    int u = ch & 0xDF;
    return u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
}

// codes[0..6] are the codes of seq[j..j+6].  fwd = index of seq[j..j+5]; rev = index of
// revcomp(seq[j+1..j+6]); an index of kTable means "unknown k-mer" (table slot with the default).
__device__ __forceinline__ void kmer_indices(const uint8_t *codes, int &fwd, int &rev) {
    int f = 0, r = 0, badf = 0, badr = 0;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        int c = codes[m];
        int d = codes[6 - m];
        badf |= c >> 2;
        badr |= d >> 2;
        f = f * 4 + (c & 3);
        r = r * 4 + (3 - (d & 3));
    }
    fwd = badf ? kTable : f;
    rev = badr ? kTable : r;
}

// ---- trimmed mean of one smoothing window (reference: modeling/smoothing.h:59-104)
//
// The reference selects OS1 = sorted[k], OS2 = sorted[n-k-1] with quickselect and adds up
// Beliakov-weighted elements; the value depends only on the multiset, so it is computed here
// from order statistics and class sums.  `x` points at the n window values (LDS or global).

struct trim_result {
    double value;
};

// k == 1 (w=101, clip=0.01: the `detect` default): two smallest / two largest in one pass.
// Returns the trimmed SUM (the caller divides by n - 2).
__device__ __forceinline__ double trimmed_sum_k1(const double *x, int n) {
#pragma clang fp contract(off)
    double lo1 = fptm::kInf, lo2 = fptm::kInf, hi1 = -fptm::kInf, hi2 = -fptm::kInf, s = 0.0;
    for (int i = 0; i < n; ++i) {
        double v = x[i];
        s += v;
        double a = fmin(lo1, v);
        lo2 = fmin(lo2, fmax(lo1, v));
        lo1 = a;
        double b = fmax(hi1, v);
        hi2 = fmax(hi2, fmin(hi1, v));
        hi1 = b;
    }
    if (lo2 < hi2) return (s - lo1) - hi1;  // OS1 < OS2: sum of sorted[1 .. n-2]
    // OS1 == OS2 == c: only the OS1 weight is applied (smoothing.h:61-69), giving
    // (b + bm - k) * c with b = #{== c}, bm = #{< c}
    double c = lo2;
    double cnt = (double)(n - 1) - ((hi1 > c) ? 1.0 : 0.0);
    return cnt * c;
}

__device__ __forceinline__ double trimmed_mean_k1(const double *x, int n) {
    return trimmed_sum_k1(x, n) / (double)(n - 2);
}

// general k >= 0: order statistics by walking distinct values from each end, then one
// class-sum pass.  O(n * (k+1)) reads; only used for non-default clip values.
__device__ __noinline__ double trimmed_mean_general(const double *x, int n, int k) {
#pragma clang fp contract(off)
    double os1 = 0.0, os2 = 0.0;
    {
        double cur = -fptm::kInf;
        int cum = 0;
        bool first = true;
        for (;;) {
            double m = fptm::kInf;
            int cnt = 0;
            for (int i = 0; i < n; ++i) {
                double v = x[i];
                if (first || v > cur) {
                    if (v < m) { m = v; cnt = 1; }
                    else if (v == m) ++cnt;
                }
            }
            first = false;
            if (cnt == 0 || cum + cnt > k) { os1 = m; break; }
            cum += cnt;
            cur = m;
        }
    }
    {
        double cur = fptm::kInf;
        int cum = 0;
        bool first = true;
        for (;;) {
            double m = -fptm::kInf;
            int cnt = 0;
            for (int i = 0; i < n; ++i) {
                double v = x[i];
                if (first || v < cur) {
                    if (v > m) { m = v; cnt = 1; }
                    else if (v == m) ++cnt;
                }
            }
            first = false;
            if (cnt == 0 || cum + cnt > k) { os2 = m; break; }
            cum += cnt;
            cur = m;
        }
    }
    double b = 0, bm = 0, d = 0, dm = 0, mid = 0;
    for (int i = 0; i < n; ++i) {
        double v = x[i];
        if (v < os1) bm += 1; else if (v == os1) b += 1;
        if (v < os2) dm += 1; else if (v == os2) d += 1;
        if (v < os2 && v > os1) mid += v;
    }
    double w1 = (b + bm - (double)k) / b;
    double w2 = ((double)(n - k) - dm) / d;
    double t = mid + b * (w1 * os1);
    if (os1 < os2) t += d * (w2 * os2);
    return t / (double)(n - 2 * k);
}

__device__ __forceinline__ double trimmed_mean(const double *x, int n, int k) {
    if (k == 1) return trimmed_mean_k1(x, n);
    return trimmed_mean_general(x, n, k);
}

// ---- the same window in the reference's own order of operations (modeling/smoothing.h:11-99)
//
// The multiset forms above give the exact trimmed sum T.  The reference's value is T plus the
// rounding noise of its evaluation: Beliakov weights such as 2/3 are inexact, and the elements
// are added up in whatever order the two in-place selections left them in.  That noise (at most
// ~1.2e-14 relative for non-negative data) matters in exactly one place: when P/Q * W' lies so
// close to a half-integer that it decides round() (predict.h:62).  Windows flagged as such are
// evaluated once more by the two functions below, operation for operation like the reference.

// smoothing.h:11-53 -- Numerical-Recipes selection of the k-th smallest of v[0..n): median of
// v[lo], v[lo+1], v[hi] as the pivot, two cursors walking towards each other.  The permutation it
// leaves in v is part of the result.  The cursor guards (i < hi, j > lo) never fire on ordered
// data -- v[lo] <= pivot <= v[hi] stop the cursors there at the latest -- and keep the walk inside
// the window when a NaN breaks the ordering (the reference then reads out of bounds).
__device__ __forceinline__ void exch(double *v, int p, int q) {
    const double t = v[p];
    v[p] = v[q];
    v[q] = t;
}
__device__ __forceinline__ double select_in_place(double *v, int n, int k) {
    int lo = 0, hi = n - 1;
    while (hi > lo + 1) {
        exch(v, (lo + hi) >> 1, lo + 1);
        if (v[lo] > v[hi]) exch(v, lo, hi);
        if (v[lo + 1] > v[hi]) exch(v, lo + 1, hi);
        if (v[lo] > v[lo + 1]) exch(v, lo, lo + 1);
        int i = lo + 1, j = hi;
        const double pivot = v[lo + 1];
        for (;;) {
            do ++i; while (i < hi && v[i] < pivot);
            do --j; while (j > lo && v[j] > pivot);
            if (j < i) break;
            exch(v, i, j);
        }
        v[lo + 1] = v[j];
        v[j] = pivot;
        if (j >= k) hi = j - 1;
        if (j <= k) lo = i;
    }
    if (hi == lo + 1 && v[hi] < v[lo]) exch(v, lo, hi);
    return v[k];
}

// smoothing.h:59-99 on a private, writable copy of the window: both selections, the class counts,
// the two weights as quotients, and the weighted elements added left to right in the permuted
// order.  Returns the trimmed SUM.
__device__ __noinline__ double trimmed_sum_reference_order(double *v, int n, int k) {
#pragma clang fp contract(off)
    const double os1 = select_in_place(v, n, k);
    const double os2 = select_in_place(v, n, n - k - 1);
    double b = 0.0, bm = 0.0, d = 0.0, dm = 0.0;
    for (int i = 0; i < n; ++i) {
        const double r = v[i];
        if (r < os1) bm += 1.0; else if (r == os1) b += 1.0;
        if (r < os2) dm += 1.0; else if (r == os2) d += 1.0;
    }
    const double w1 = (b + bm - (double)k) / b;
    const double w2 = ((double)(n - k) - dm) / d;
    double t = 0.0;
    for (int i = 0; i < n; ++i) {
        const double x = v[i];
        double term;  // smoothing.h:59-70, in its order of tests (it matters when os1 == os2)
        if (x < os2 && x > os1) term = x;
        else if (x < os1) term = 0.0;
        else if (x > os2) term = 0.0;
        else if (x == os1) term = w1 * x;
        else term = w2 * x;
        t += term;
    }
    return t;
}

// Could the reference's rounding noise move round(x) away from round() of the exactly computed
// product?  x = P/Q * T/(n-2k) from the exact trimmed sum T; tol_x bounds |x_reference - x|.
__device__ __forceinline__ bool near_rounding_tie(double x, double tol_x) {
    const double ax = fabs(x);
    return fabs((ax - floor(ax)) - 0.5) <= tol_x;
}

// Workgroup-cooperative re-evaluation: lanes with `need` copy their window (n values at src) into
// one of n_slots scratch windows in LDS (handed out through an LDS counter, as many rounds as it
// takes) and run the reference's order of operations on it.  Must be called by every lane of the
// workgroup; scratch (n_slots * n doubles) may still be read by other lanes on entry -- the first
// vote is a barrier.  Returns the trimmed sum for lanes with `need`, 0 otherwise.
__device__ __forceinline__ double trimmed_sum_rounds(bool need, const double *src, int n, int k, double *scratch,
                                                     int n_slots, int *counter, int tid) {
    double res = 0.0;
    bool pending = need;
    while (__syncthreads_or(pending ? 1 : 0)) {
        if (tid == 0) *counter = 0;
        __syncthreads();
        if (pending) {
            const int slot = atomicAdd(counter, 1);
            if (slot < n_slots) {
                double *buf = scratch + (size_t)slot * n;
                for (int i = 0; i < n; ++i) buf[i] = src[i];
                res = trimmed_sum_reference_order(buf, n, k);
                pending = false;
            }
        }
    }
    return res;
}

// Correctly rounded t / d for a divisor whose reciprocal rd = RN(1/d) is computed once
// (Markstein: q = RN(t*rd) is within an ulp, the remainder fma is exact, the corrected quotient
// rounds like the true division).  Replaces the ~25-instruction IEEE divide where the divisor
// is loop invariant (the w - 2k of the trimmed mean).
__device__ __forceinline__ double div_invariant(double t, double d, double rd) {
#pragma clang fp contract(off)
    const double q = t * rd;
    const double r = fma(-q, d, t);
    const double q2 = fma(r, rd, q);
    return isfinite(q) ? q2 : q;  // d > 1, so q is non-finite only when t is (inf / NaN pass through)
}

// ---- wave64 inclusive scans on the DPP cross-lane path of the vector ALU (no LDS traffic).
// gfx9-family pattern: row_shr:1,2,4,8 scan each row of 16 lanes, row_bcast:15 carries rows
// 0->1 and 2->3, row_bcast:31 carries the lower half into rows 2,3.  Lanes whose DPP source is
// masked or out of the row receive `ident`, the identity of the operator.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_i32(int ident, int x) {
    return __builtin_amdgcn_update_dpp(ident, x, CTRL, ROW_MASK, BANK_MASK, false);
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double ident, double x) {
    int lo = dpp_i32<CTRL, ROW_MASK, BANK_MASK>(__double2loint(ident), __double2loint(x));
    int hi = dpp_i32<CTRL, ROW_MASK, BANK_MASK>(__double2hiint(ident), __double2hiint(x));
    return __hiloint2double(hi, lo);
}

struct op_add {
    __device__ __forceinline__ double operator()(double a, double b) const { return a + b; }
    __device__ __forceinline__ int operator()(int a, int b) const { return a + b; }
};
struct op_min {
    __device__ __forceinline__ double operator()(double a, double b) const { return fmin(a, b); }
};
struct op_max {
    __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); }
};

#define FPT_DPP_SCAN_STEPS(MOV)                                      \
    x = op(MOV<0x111, 0xf, 0xf>(ident, x), x); /* row_shr:1 */       \
    x = op(MOV<0x112, 0xf, 0xf>(ident, x), x); /* row_shr:2 */       \
    x = op(MOV<0x114, 0xf, 0xf>(ident, x), x); /* row_shr:4 */       \
    x = op(MOV<0x118, 0xf, 0xf>(ident, x), x); /* row_shr:8 */       \
    x = op(MOV<0x142, 0xa, 0xf>(ident, x), x); /* row_bcast:15 */    \
    x = op(MOV<0x143, 0xc, 0xf>(ident, x), x); /* row_bcast:31 */

template <typename Op>
__device__ __forceinline__ double wave_scan_f64(double x, double ident, Op op) {
    FPT_DPP_SCAN_STEPS(dpp_f64)
    return x;
}

__device__ __forceinline__ int wave_scan_i32(int x) {
    const int ident = 0;
    op_add op;
    FPT_DPP_SCAN_STEPS(dpp_i32)
    return x;
}

struct op_min_i {
    __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; }
};
struct op_max_i {
    __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; }
};

template <typename Op>
__device__ __forceinline__ int wave_scan_i32_op(int x, int ident, Op op) {
    FPT_DPP_SCAN_STEPS(dpp_i32)
    return x;
}

// Extrema scans of NON-NEGATIVE int32 values: an unsigned max whose identity is 0 folds the DPP
// move into the v_max_u32 itself (zero fill for lanes without a source), one instruction per
// step instead of three; the minimum is the same scan on kNonnegTop - x.
constexpr int kNonnegTop = 0x3fffffff;
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ unsigned dpp_u32_zero(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, BANK_MASK, true);
}
__device__ __forceinline__ unsigned wave_scan_umax(unsigned x) {
#define FPT_UMAX_STEP(C, R) { const unsigned y_ = dpp_u32_zero<C, R, 0xf>(x); x = y_ > x ? y_ : x; }
    FPT_UMAX_STEP(0x111, 0xf) FPT_UMAX_STEP(0x112, 0xf) FPT_UMAX_STEP(0x114, 0xf)
    FPT_UMAX_STEP(0x118, 0xf) FPT_UMAX_STEP(0x142, 0xa) FPT_UMAX_STEP(0x143, 0xc)
#undef FPT_UMAX_STEP
    return x;
}
// x in [0, kNonnegTop], or `absent` for positions beyond the data (identity of the scan)
__device__ __forceinline__ int scan_max_nonneg(int x, bool present) {
    return (int)wave_scan_umax(present ? (unsigned)x : 0u);
}
__device__ __forceinline__ int scan_min_nonneg(int x, bool present) {
    return kNonnegTop - (int)wave_scan_umax(present ? (unsigned)(kNonnegTop - x) : 0u);
}

// inclusive prefix sum within each row of 16 lanes (row_shr:1,2,4,8 with zero fill): the first
// level of a two-level workgroup-wide prefix sum -- 12 instructions instead of the 20+ of the
// full-wavefront scan, whose row_bcast steps need a masked move per half
template <int CTRL>
__device__ __forceinline__ double dpp_f64_zero(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_scan_f64(double x) {
    x += dpp_f64_zero<0x111>(x);
    x += dpp_f64_zero<0x112>(x);
    x += dpp_f64_zero<0x114>(x);
    x += dpp_f64_zero<0x118>(x);
    return x;
}

// typed front-ends: the smoothing phase runs its scans either on doubles or, when every window
// sum of the tile is a small integer (cut counts are), on int32, where a DPP scan step is one
// v_add/v_min/v_max with a DPP operand instead of 5-7 instructions for a 64-bit value
__device__ __forceinline__ double scan_add(double x) { return wave_scan_f64(x, 0.0, op_add()); }
__device__ __forceinline__ int scan_add(int x) { return wave_scan_i32(x); }
__device__ __forceinline__ double scan_min(double x) { return wave_scan_f64(x, fptm::kInf, op_min()); }
__device__ __forceinline__ int scan_min(int x) { return wave_scan_i32_op(x, 0x7fffffff, op_min_i()); }
__device__ __forceinline__ double scan_max(double x) { return wave_scan_f64(x, -fptm::kInf, op_max()); }
__device__ __forceinline__ int scan_max(int x) { return wave_scan_i32_op(x, (int)0x80000000, op_max_i()); }

template <typename T> struct scan_lim;
template <> struct scan_lim<double> {
    static __device__ __forceinline__ double hi() { return fptm::kInf; }
    static __device__ __forceinline__ double lo() { return -fptm::kInf; }
    static __device__ __forceinline__ double pick_min(double a, double b) { return fmin(a, b); }
    static __device__ __forceinline__ double pick_max(double a, double b) { return fmax(a, b); }
};
template <> struct scan_lim<int> {
    static __device__ __forceinline__ int hi() { return 0x7fffffff; }
    static __device__ __forceinline__ int lo() { return (int)0x80000000; }
    static __device__ __forceinline__ int pick_min(int a, int b) { return a < b ? a : b; }
    static __device__ __forceinline__ int pick_max(int a, int b) { return a > b ? a : b; }
};

// inclusive prefix sum of a (double, int) pair across the 64 lanes
__device__ __forceinline__ void wave_scan(double &v, int &c, int /*lane*/) {
    v = wave_scan_f64(v, 0.0, op_add());
    c = wave_scan_i32(c);
}

// sum over [lo, hi] (lo <= hi) from per-tile inclusive prefix sums (tile = 64 positions):
// suffix of the first tile + whole middle tiles + prefix of the last tile
template <typename T>
__device__ __forceinline__ T tile_range_sum(const T *ps, int lo, int hi) {
    const int qlo = lo >> 6, qhi = hi >> 6;
    T s = ps[hi];
    if (qlo == qhi) {
        if (lo & 63) s -= ps[lo - 1];
        return s;
    }
    for (int q = qlo + 1; q < qhi; ++q) s += ps[(q << 6) + 63];
    T head = ps[(qlo << 6) + 63];
    if (lo & 63) head -= ps[lo - 1];
    return head + s;
}

// min / max over [lo, hi] spanning at least two tiles, from per-tile prefix (p) and suffix (s) scans
template <typename T>
__device__ __forceinline__ T tile_range_min(const T *p, const T *s, int lo, int hi) {
    T m = scan_lim<T>::pick_min(s[lo], p[hi]);
    for (int q = (lo >> 6) + 1; q < (hi >> 6); ++q) m = scan_lim<T>::pick_min(m, p[(q << 6) + 63]);
    return m;
}

template <typename T>
__device__ __forceinline__ T tile_range_max(const T *p, const T *s, int lo, int hi) {
    T m = scan_lim<T>::pick_max(s[lo], p[hi]);
    for (int q = (lo >> 6) + 1; q < (hi >> 6); ++q) m = scan_lim<T>::pick_max(m, p[(q << 6) + 63]);
    return m;
}

// Branch-free forms for windows that span at most three tiles (hi - lo <= 128) and start in a
// different tile than they end in or not: every operand is read (from a clamped, always valid
// index) and selected, so a wavefront whose lanes straddle different tile counts does not
// serialise the cases.  Same association order as the loops above.
template <typename T>
__device__ __forceinline__ T tile_range_sum3(const T *ps, int lo, int hi) {
    const int qlo = lo >> 6, nq = (hi >> 6) - qlo;
    const int last = hi | 63;  // end of the tile hi is in: inside the padded array
    const int i1 = (qlo << 6) + 63, i2 = i1 + 64;
    const T s_hi = ps[hi];
    const T below = ps[(lo & 63) ? lo - 1 : lo];
    const T e1 = ps[i1 < last ? i1 : last], e2 = ps[i2 < last ? i2 : last];
    const T sub = (lo & 63) ? below : (T)0;
    const T s = nq == 2 ? s_hi + e2 : s_hi;
    return nq == 0 ? s_hi - sub : (e1 - sub) + s;
}

template <typename T>
__device__ __forceinline__ T tile_range_min3(const T *p, const T *s, int lo, int hi) {
    const int qlo = lo >> 6, nq = (hi >> 6) - qlo;  // 1 or 2
    const T m = scan_lim<T>::pick_min(s[lo], p[hi]);
    const T mid = p[(qlo << 6) + 127 < (hi | 63) ? (qlo << 6) + 127 : (hi | 63)];
    return nq == 2 ? scan_lim<T>::pick_min(m, mid) : m;
}

template <typename T>
__device__ __forceinline__ T tile_range_max3(const T *p, const T *s, int lo, int hi) {
    const int qlo = lo >> 6, nq = (hi >> 6) - qlo;
    const T m = scan_lim<T>::pick_max(s[lo], p[hi]);
    const T mid = p[(qlo << 6) + 127 < (hi | 63) ? (qlo << 6) + 127 : (hi | 63)];
    return nq == 2 ? scan_lim<T>::pick_max(m, mid) : m;
}

}  // namespace fptd

// ---------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al., SC'11): the null draws of the
// empirical-FDR pass are a pure function of (seed, base index, sample index).
// ---------------------------------------------------------------------------
namespace fptd {

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                       uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Four uniforms in (0, 1) from ONE Philox block keyed by (seed, base, quad): word w of the block is
// the draw of sample 4 * quad + w, u = (word + 1/2) * 2^-32.  A null draw is an inverse-CDF bracket
// in a tabulated cdf row (the smallest k with cdf(k) >= u), so 32 bits of resolution only merge
// brackets narrower than 2.3e-10 -- far below what 100 draws per base can tell apart -- and the
// half-step keeps u off 0 and 1.  One block per four draws instead of two halves the generator's
// share of the kernel (measured: DESIGN.md, empirical-FDR kernel).
__host__ __device__ __forceinline__ void philox_uniform4(uint64_t seed, uint64_t base, uint32_t quad, double u[4]) {
    uint32_t o[4];
    philox4x32_10((uint32_t)base, (uint32_t)(base >> 32), quad, 0x66707464u /* "fptd" */,
                  (uint32_t)seed, (uint32_t)(seed >> 32), o);
#pragma unroll
    for (int w = 0; w < 4; ++w) u[w] = ((double)o[w] + 0.5) * (1.0 / 4294967296.0);
}

// the uniform of one (seed, base, sample)
__host__ __device__ __forceinline__ double philox_uniform(uint64_t seed, uint64_t base, uint32_t sample) {
    double u[4];
    philox_uniform4(seed, base, sample >> 2, u);
    return u[sample & 3u];
}

}  // namespace fptd
