// fpt_lean_common.hpp -- the pieces of the first-pass kernel of memo mode (fpt_scan_lean.hip: one workgroup per
// tile) that do not depend on its LDS layout: the kernel arguments, fp64 arithmetic with scalar-register
// operands, the one-formula normal cdf, the Stouffer-window phase and the track stores.  gfx950 only.
#pragma once
#include "fpt_kernels.hpp"

#include <cstddef>
#include <cstdlib>

#include "fpt_device.hpp"

namespace fptlean {
using namespace fptd;

typedef unsigned int u32;

constexpr int kHW = 5, kSHW = 50, kPad = kHW + kSHW, kW = 2 * kSHW + 1;
constexpr u32 kCountMax = 6553;  // 10 of them fit 16 bits

// fp64 constants that the kernel wants in scalar registers: they travel in the kernel-argument
// segment, which the compiler cannot fold back into literals
struct lean_coef {
    double g[FPT_NDTR_G_N + 1];  // FPT_NDTR_G_LIST divided by its first entry (monic form)
    double e[FPT_NDTR_E_N + 1];  // FPT_NDTR_E_LIST times that entry
    double neg_r0, neg_half_log2e;
    double c99, band, limit;
    double inv_g0;  // 1 / FPT_NDTR_G_LIST[0]: the table of g is staged divided by what e's coefficients carry
};

struct lean_args {
    int32_t interval_len;        // uniform mode when interval_off == nullptr
    const int64_t *interval_off; // ragged: output offsets
    const fptk::lean_tile_rec *tile_recs;  // ragged: one 32-byte record per tile (ONE scalar load: a short
                                           // workgroup lives ~7 us, and every dependent load is ~5 % of it)
    int64_t tile_first;
    int32_t tiles_per_interval, tile_len;
    int32_t n_scales;
    int32_t scales[FPT_MAX_SCALES];
    double scale_rsqrt[FPT_MAX_SCALES];
    int32_t max_scale;
    int64_t total_bases;
    const double *counts_plus, *counts_minus;
    const uint8_t *seq;
    const double2 *table2;  // 4096 x (forward, reverse-complement) propensity, bit-plane index
    double *exp_out, *obs_out, *pval_out, *winp_out;
    const double2 *memo;
    int32_t memo_exp, memo_obs;
    int32_t *redo;
    int32_t *miss_max;  // [0] largest exp, [1] largest obs among the pairs that missed the table
    int32_t miss_rows, miss_stride;  // ... as far as the second-level table could hold them
    // the second-level table the context keeps (fpt_capi.cpp): filled for exp <= memo2_have[0] and
    // obs <= memo2_have[1] by earlier calls; a pair that misses the first level is looked up there
    const double2 *memo2;
    const int32_t *memo2_have;
    const int32_t *dm_ids;
    int32_t stop;  // timing-only diagnostics, honoured only in -DFPT_ABLATE builds (FPT_ABLATE env)
    int32_t tab;   // phase E's g from the LDS table (FPT_LEAN_TAB=0, read once: from the Horner chain -- A/B runs)
    int32_t prio;  // raised wave priority until a tile's inputs are staged (FPT_LEAN_PRIO=0, read once, switches it off: A/B runs)
    int64_t *trace;  // -DFPT_ABLATE builds: per-workgroup timestamps (FPT_LEAN_TRACE)
    lean_coef c;
};
#ifdef FPT_ABLATE
#define LEAN_STOP(n) (a.stop == (n))
// word 0: HW_ID | XCC_ID << 32; words 1..6: 100 MHz clock of the first wavefront at the start, after
// the wait for the loads, after barriers 1 and 2, after phase D, at the end; word 7: end of the last
#define LEAN_TRACE(n)                                                                              \
    if (a.trace && (threadIdx.x == 0 || ((n) == 6 && threadIdx.x == blockDim.x - 64))) {           \
        int64_t *tr = a.trace + (int64_t)blockIdx.x * 8;                                            \
        if ((n) == 1)                                                                              \
            tr[0] = (int64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                           \
                    ((int64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);                   \
        tr[threadIdx.x == 0 ? (n) : 7] = (int64_t)wall_clock64();                                  \
    }
#else
#define LEAN_STOP(n) false
#define LEAN_TRACE(n)
#endif

// ---- fp64 arithmetic with one operand in scalar registers (VOP3, one instruction each)
__device__ __forceinline__ double mul_vs(double a, double s) {
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s));
    return r;
}
__device__ __forceinline__ double add_vs(double a, double s) {
    double r;
    asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s));
    return r;
}
__device__ __forceinline__ double fma_svv(double s, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(s), "v"(b), "v"(c));
    return r;
}

typedef const __attribute__((address_space(4))) lean_coef kcoef;        // in the kernel-argument segment
typedef const __attribute__((address_space(4))) double kdouble;

#define FPT_HSTEP(n) "v_fma_f64 %0, %0, %1, %" #n "\n\t"
// Horner chains with the coefficients in scalar registers.  g is evaluated in monic form
// (x^14 + c[1] x^13 + ... + c[14], the leading coefficient folded into e's on the host), which
// opens the chain with one v_add instead of a v_mov + v_fma; e opens with a move.
__device__ __forceinline__ double horner_g_s(double x, kdouble *c) {
    static_assert(FPT_NDTR_G_N == 14, "asm operand list is written for degree 14");
    double acc;
    asm("v_add_f64 %0, %1, %2\n\t" FPT_HSTEP(3) FPT_HSTEP(4) FPT_HSTEP(5) FPT_HSTEP(6) FPT_HSTEP(7) FPT_HSTEP(8)
            FPT_HSTEP(9) FPT_HSTEP(10) FPT_HSTEP(11) FPT_HSTEP(12) FPT_HSTEP(13) FPT_HSTEP(14) FPT_HSTEP(15)
        : "=&v"(acc)
        : "v"(x), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]), "s"(c[6]), "s"(c[7]), "s"(c[8]), "s"(c[9]),
          "s"(c[10]), "s"(c[11]), "s"(c[12]), "s"(c[13]), "s"(c[14]));
    return acc;
}
__device__ __forceinline__ double horner_e_s(double x, kdouble *c) {
    static_assert(FPT_NDTR_E_N == 8, "asm operand list is written for degree 8");
    double acc;
    asm("v_mov_b64 %0, %2\n\t" FPT_HSTEP(3) FPT_HSTEP(4) FPT_HSTEP(5) FPT_HSTEP(6) FPT_HSTEP(7) FPT_HSTEP(8)
            FPT_HSTEP(9) FPT_HSTEP(10)
        : "=&v"(acc)
        : "v"(x), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]), "s"(c[6]), "s"(c[7]), "s"(c[8]));
    return acc;
}
#undef FPT_HSTEP

// fptm::ndtr_fast with the constants in scalar registers (same operations, same coefficients).
// The 14 + 9 coefficient pairs do not fit the scalar register file next to the kernel's own
// state, and left alone the compiler loads them all before the loop over the scales and spills
// them through v_writelane / v_readlane (32 vector instructions per evaluation).  Passing the
// pointer through an empty asm that depends on the argument keeps the loads (three
// s_load_dwordx16) inside the evaluation; both sets fit at degrees 14 / 8, so one wait covers them
// (measured against a second hand-over between the chains: 25.1 vs 25.6 ms on config 3).
// Valid for a > -limit (26).  Beyond +26 the result is exactly 1.0, as in the reference: y underflows to
// zero through ldexp whatever the polynomials extrapolate to (they stay bounded: 1/(t+5) only moves
// from 0.032 to 0 -- which is also how the sentinel edge slot yields 1.0), so only arguments below
// -26, where the reference's own value runs into the subnormal range, have to leave this kernel.
// TAB: g from the 256-interval table of fptm::ndtr_fast_tab staged in LDS (`gt`: c3, c2, c1, c0 per
// interval TIMES the leading coefficient the e chain carries -- see fill_lean_gtab) instead of the
// 14-step chain: 15 fp64 instructions become 6 fp64 + 2 integer ones and two 16-byte LDS reads.
template <bool TAB = false>
__device__ __forceinline__ double ndtr_fast_s(double a, kcoef *c, const double *gt = nullptr) {
    asm volatile("" : "+s"(c) : "v"(a));
    const double t = fabs(a);
    const double d = t + 5.0;
    double r = __builtin_amdgcn_rcp(d);  // 2^-24 (measured 4.6e-8); one Newton step: 2.2e-15
    r = fma(fma(-d, r, 1.0), r, r);
    double g;
    if (TAB) {
        const double kf = fma(r, FPT_NDTR_GTAB_SCALE, -(FPT_NDTR_GTAB_XLO * FPT_NDTR_GTAB_SCALE));
        const int k = min(max((int)kf, 0), FPT_NDTR_GTAB_N - 1);
        const double w = __builtin_amdgcn_fract(kf);
        // (two arrays of 16-byte halves, not one of 32-byte entries: slots k and k' then share LDS banks when
        // k = k' mod 8 instead of mod 4 -- a wavefront's lanes read a dozen different slots)
        const double2 c32 = *reinterpret_cast<const double2 *>(gt + 2 * k);
        const double2 c10 = *reinterpret_cast<const double2 *>(gt + 2 * FPT_NDTR_GTAB_N + 2 * k);
        g = fma(fma(fma(c32.x, w, c32.y), w, c10.x), w, c10.y);
    } else {
        g = horner_g_s(add_vs(r, c->neg_r0), c->g);  // g / its leading coefficient
    }
    const double q = mul_vs(t * t, c->neg_half_log2e);           // exp(-t^2/2) = 2^q = 2^n 2^(q - n)
    const double n = rint(q);
    const double e = horner_e_s(q - n, c->e);                   // e * that coefficient
    const double y = ldexp(e * g, (int)n);
    return a > 0.0 ? 1.0 - y : y;
}

// A uniform read of launch-constant data (tile table, interval offsets, model ids) as a SCALAR
// load: through a generic pointer the compiler issues a vector load plus v_readfirstlane, and the
// in-order memory counter then makes the wavefront wait for every store still in flight (the
// tracks of the previous phase) before the next tile's geometry is known.
template <typename T>
__device__ __forceinline__ T uniform_load(const T *p, int64_t i) {
    return ((const __attribute__((address_space(4))) T *)p)[i];
}

// the three per-base tracks of a lane between phase D and their stores
struct lean_tracks {
    double ex, pv;
    u32 k;
};

// which lanes own an output base of the tile, and which one
struct lean_owner {
    int t, L;
    int64_t out_off;
    bool mine;
};
// base[byte_off / 8] = v with a uniform base and an unsigned 32-bit BYTE offset per lane: the form the
// store instruction takes directly (base in scalar registers), without 64-bit vector address adds
__device__ __forceinline__ void store_at(double *base, u32 byte_off, double v) {
    *reinterpret_cast<double *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// (Loads and stores through buffer descriptors -- the hardware's range check in place of a divergent
// `if` around a ragged row -- were measured: config 3 24.9-25.0 ms against 23.7, config 2 0.86 against
// 0.84.  The scalar instructions they save are hidden behind the vector work of the other seven
// wavefronts of the SIMD; the MUBUF path is slower than the global one for these 8-byte lanes.)
template <typename Args>
__device__ __forceinline__ void lean_store_tracks(const Args &a, const lean_owner &o, const lean_tracks &tr) {
    // (intervals of 2^29 bases and more never reach these kernels: fpt_scan_dev)
    if (o.mine) {
        const u32 t8 = (u32)o.t * 8u;
        if (a.exp_out) store_at(a.exp_out + o.out_off, t8, tr.ex);
        if (a.obs_out) store_at(a.obs_out + o.out_off, t8, (double)tr.k);
        if (a.pval_out) store_at(a.pval_out + o.out_off, t8, tr.pv);
    }
}

// ---- E with several scales, in pieces so that two tiles can share the barriers:
// Z[16 + i] = prefix sum of z up to base i (row-of-16 prefix + C[1 + row], the sum of the rows
// before); Z[15] = 0 stands for "before the first base", and the slot kEdge holds a prefix of -1e4,
// which makes the window p-value of a base near the interval's edge come out as exactly 1.0
// (windowing.pyx:51) without a select: ndtr(+1e4 / sqrt(K)) = 1.
template <int NT>
__device__ __forceinline__ double lean_z_rows(double z, int tid, double *rowtot) {  // then a barrier
    const double zr = row_scan_f64(z);  // lanes beyond nt hold 0
    if ((tid & 15) == 15) rowtot[tid >> 4] = zr;
    return zr;
}
template <int NT>
__device__ __forceinline__ void lean_z_carries(int tid, const double *rowtot, double *C) {  // first wavefront; then a barrier
    constexpr int NROW = NT / 16;
    if (tid < kWave) {
        const double tv = tid < NROW ? rowtot[tid] : 0.0;
        const double inc = wave_scan_f64(tv, 0.0, op_add());
        if (tid < NROW) C[1 + tid] = inc - tv;
    }
}
// the carries applied once: a scale then costs two reads and one subtraction (then a barrier)
template <int NT>
__device__ __forceinline__ void lean_z_finish(double zr, int tid, const double *C, double *Z) {
    Z[16 + tid] = zr + C[1 + (tid >> 4)];
}
template <int NT, typename Args, bool TAB = false>
__device__ __forceinline__ bool lean_windows(const Args &a, kcoef *kc, const lean_owner &o, int tid, const double *Z,
                                             const double *gt = nullptr) {
    constexpr int kEdge = NT + 32 + 15;  // beyond every lane's slot
    // (uniform; typed as GLOBAL memory: behind the empty asm below a generic pointer made the store a
    // flat_store with a 64-bit vector add for its address -- and flat instructions count in lgkmcnt, which the
    // next scale's LDS reads wait for; this way it is global_store with the base in scalar registers)
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble *row = (gdouble *)(a.winp_out + o.out_off);
    const u32 t8 = (u32)o.t * 8u;
    // a window of half-width hs fits iff hs <= the distance to the nearer end (-1: not this lane's base)
    const int room = o.mine ? min(o.t, o.L - 1 - o.t) : -1;
    // The lane's two prefix slots as 32-bit LDS addresses, made ONCE (the empty asm keeps the
    // compiler from re-deriving them from the array base inside the loop: machine LICM is off for
    // this library), so that a scale costs an add, a subtract, a compare and two selects.
    typedef const __attribute__((address_space(3))) double lds_double;
    u32 ahi0 = (u32)(size_t)(lds_double *)(Z + 16 + tid), alo0 = (u32)(size_t)(lds_double *)(Z + 15 + tid);
    u32 aedge = (u32)(size_t)(lds_double *)(Z + kEdge), afirst = (u32)(size_t)(lds_double *)(Z + 15);
    asm volatile("" : "+v"(ahi0), "+v"(alo0), "+v"(aedge), "+v"(afirst));  // (vector registers: a select cannot read a scalar next to vcc)
    const double neg_limit = -kc->limit;
    bool low = false;  // an argument below -26: the tile needs the restated ndtr.c (see ndtr_fast_s)
    const int n_scales = a.n_scales;
    const int64_t stride = a.total_bases;
    for (int s = 0; s < n_scales; ++s) {
        const int hs = a.scales[s];
        const u32 hs8 = (u32)hs * 8u;
        const bool inside = hs <= room;
        const u32 ah = inside ? ahi0 + hs8 : aedge, al = inside ? alo0 - hs8 : afirst;
        const double sv = *(lds_double *)(size_t)ah - *(lds_double *)(size_t)al;
        const double arg = -(sv * a.scale_rsqrt[s]);  // (the edge lanes' argument is +1e4 / sqrt(K))
        low |= !(arg > neg_limit);
        const double pw = LEAN_STOP(4) ? arg : ndtr_fast_s<TAB>(arg, kc, gt);
        asm volatile("" : "+s"(row));  // keeps the scale's base in scalar registers (no per-lane pointer carried through the loop)
        // (written out: the compiler forms the address with a 64-bit vector add and stores through it)
        if (o.mine) asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(t8), "v"(pw), "s"(row) : "memory");
        row += stride;
    }
    return low;
}
// one narrow scale (the reference's only one is 3): Z holds the raw z, summed left to right
template <int NT, typename Args, bool TAB = false>
__device__ __forceinline__ bool lean_window_narrow(const Args &a, kcoef *kc, const lean_owner &o, int tid, const double *Z,
                                                   const double *gt = nullptr) {
    const int hs = a.scales[0];
    const bool inside = o.mine && o.t >= hs && o.t < o.L - hs;
    double sv = 0.0;
    if (inside)
        for (int j = 16 + tid - hs; j <= 16 + tid + hs; ++j) sv += Z[j];
    const double arg = inside ? -(sv * a.scale_rsqrt[0]) : 1e3;  // edges are 1.0 (windowing.pyx:51)
    const double pw = LEAN_STOP(4) ? arg : ndtr_fast_s<TAB>(arg, kc, gt);
    if (o.mine) store_at(a.winp_out + o.out_off, (u32)o.t * 8u, pw);
    return inside && !(arg > -kc->limit);
}

inline int lean_tab() {
    static const int v = getenv("FPT_LEAN_TAB") ? atoi(getenv("FPT_LEAN_TAB")) : 1;
    return v;
}
inline int lean_prio() {
    static const int v = getenv("FPT_LEAN_PRIO") ? atoi(getenv("FPT_LEAN_PRIO")) : 1;
    return v;
}
// the kernel arguments of a launch (both kernels take the same block)
inline void fill_lean_args(const fptk::scan_launch &sl, lean_args &a) {
    a.interval_len = sl.interval_len;
    a.interval_off = sl.interval_off;
    a.tile_recs = (const fptk::lean_tile_rec *)sl.tile_recs;
    a.tile_first = sl.tile_first;
    a.tiles_per_interval = sl.tiles_per_interval;
    a.tile_len = sl.tile_len;
    a.n_scales = sl.n_scales;
    a.max_scale = 0;
    for (int i = 0; i < FPT_MAX_SCALES; ++i) {
        a.scales[i] = i < sl.n_scales ? sl.scales[i] : 0;
        // 1/sqrt(K): the reference divides by sqrt(K) (windowing.h:64); multiplying by the
        // reciprocal moves z by an ulp, far inside the 1e-6 contract on the window p-value
        a.scale_rsqrt[i] = i < sl.n_scales ? 1.0 / sqrt((double)(2 * sl.scales[i] + 1)) : 1.0;
        if (i < sl.n_scales && sl.scales[i] > a.max_scale) a.max_scale = sl.scales[i];
    }
    a.total_bases = sl.total_bases;
    a.counts_plus = sl.counts_plus;
    a.counts_minus = sl.counts_minus;
    a.seq = sl.seq;
    a.table2 = (const double2 *)sl.table2;
    a.exp_out = sl.exp_out;
    a.obs_out = sl.obs_out;
    a.pval_out = sl.pval_out;
    a.winp_out = sl.winp_out;
    a.memo = (const double2 *)sl.memo;
    a.memo_exp = sl.memo_exp;
    a.memo_obs = sl.memo_obs;
    a.redo = sl.redo;
    a.miss_max = sl.memo2 ? sl.memo2_max : nullptr;
    a.miss_rows = sl.memo2_rows;
    a.miss_stride = sl.memo2_stride;
    a.memo2 = (const double2 *)sl.memo2;
    a.memo2_have = sl.memo2_have;
    a.dm_ids = sl.dm_ids;
    a.stop = sl.ablate;
    a.prio = lean_prio();
    a.tab = lean_tab();
    a.trace = nullptr;
    const double g[FPT_NDTR_G_N + 1] = {FPT_NDTR_G_LIST}, e[FPT_NDTR_E_N + 1] = {FPT_NDTR_E_LIST};
    for (int i = 0; i <= FPT_NDTR_G_N; ++i) a.c.g[i] = g[i] / g[0];
    for (int i = 0; i <= FPT_NDTR_E_N; ++i) a.c.e[i] = e[i] * g[0];
    a.c.neg_r0 = -fptm::kNdtrR0;
    a.c.neg_half_log2e = fptm::kNdtrNegHalfLog2e;
    a.c.c99 = (double)(kW - 2);
    a.c.band = 1e-13;
    a.c.limit = fptm::kNdtrFastLimit;
    a.c.inv_g0 = 1.0 / g[0];
}

}  // namespace fptlean
