// fpt_scan_wave.hip -- the first pass of the fused scan in memo mode for SHORT intervals: one
// wavefront per interval, no barrier anywhere (gfx950 / CDNA4, wave64).
//
// Same path and same case as k_scan_lean (cli/detect.py:120-130 per interval: 6-mer lookup,
// predict.h:41-63 with the smoothing of smoothing.h:107-133, the (exp, obs) table, the Stouffer
// windows of windowing.h:53-84; small non-negative integer counts, A/C/G/T, hw 5 / shw 50 / clip
// 0.01) -- anything else flags the interval in `redo` for the general kernel.  What is different:
//
// A whole-genome hotspot set is made of short intervals (mean ~160 bases, half of them below 140).
// As workgroups of two or three wavefronts they spend their ~7 us of life waiting: five barriers
// chain the latencies of the input loads, the bias-table gather and the (exp, obs) gather, and
// 59 % of the vector-issue slots are used (profiles/r03_cfg4_*).  Here an interval of up to
// RP * 64 - 117 bases -- RP * 64 padded positions, RP = 4: 139 bases -- belongs to ONE wavefront:
//   * position v = 64 * row + lane for the RP "position rows" (coalesced loads, four to six per
//     array and lane, all in flight at once), base t at slot u = t + 6 of the RB = RP - 1 "base rows";
//   * the sequence bit planes never leave the scalar registers (RP ballots per plane), the 6-mer
//     index is a funnel shift of two of them;
//   * every scan result stays in vector registers: prefix sums of the window sums over the whole
//     interval (row scans on DPP + a scalar carry), prefix and suffix extrema per 64-position
//     tile.  A base's smoothing window [u, u + 100] ends 100 positions = one row and 36 lanes
//     further on, so what it needs from there arrives through ONE ds_bpermute per quantity (the
//     two source rows are merged on the source side: lanes >= 36 from row + 1, the others from
//     row + 2) -- no LDS memory, no store/load pairs, no address arithmetic;
//   * the '-' strand is staged one position to the right (its counts by a wave_shr:1, its
//     propensities as the table already holds them), so both strands use the same lanes and
//     indices everywhere and the observed count is one packed word;
//   * LDS holds only the packed counts (2*hw window sums) and the two propensity rows (the
//     left-to-right sum of predict.h:43-47): 3.6 KB per wavefront, so a CU holds 32 wavefronts
//     that never wait for each other; the z prefix of the window phase takes the propensities' place;
//   * rows nobody needs are skipped by scalar branches: table gathers beyond position L + 60,
//     prefix extrema of row 0, suffix extrema beyond the row of the last window start.
#include "fpt_lean_common.hpp"

#include <cstdlib>

using namespace fptd;
using namespace fptlean;

namespace {

// program-order fence for one wavefront: LDS instructions of a wavefront execute in order, so data
// written by one lane is visible to another after the next instruction -- the compiler only has
// to keep the accesses on their side of this point (no s_barrier, no s_waitcnt)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b) {  // v_pk_max_u16: both halves at once
    const u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(u32, r);
}

typedef const __attribute__((address_space(4))) lean_args wave_kargs;
#define WAVE_ARGS(name)                 \
    asm volatile("" : "+s"(ka));        \
    wave_kargs &name = *ka

// N independent inclusive wavefront scans advanced in step: a DPP operand may not be read for two
// wait states after the vector instruction that wrote it, so a lone chain costs an s_nop per step
// (and every instruction, s_nop included, takes an issue slot: the kernel is bound by those);
// with four or more chains in step there is always something else to issue
#define WAVE_SCAN_STEP_ALL(C, R, BC)                                                              \
    _Pragma("unroll") for (int k = 0; k < NA; ++k)                                                \
        xa[k] += (u32)__builtin_amdgcn_update_dpp(0, (int)xa[k], C, R, 0xf, BC);                  \
    _Pragma("unroll") for (int k = 0; k < NM; ++k) {                                              \
        const u32 y_ = (u32)__builtin_amdgcn_update_dpp(0, (int)xm[k], C, R, 0xf, BC);            \
        xm[k] = y_ > xm[k] ? y_ : xm[k];                                                          \
    }
template <int NA, int NM>
__device__ __forceinline__ void wave_scan_multi(u32 *xa /* NA prefix sums */, u32 *xm /* NM prefix maxima (unsigned) */) {
    WAVE_SCAN_STEP_ALL(0x111, 0xf, true)
    WAVE_SCAN_STEP_ALL(0x112, 0xf, true)
    WAVE_SCAN_STEP_ALL(0x114, 0xf, true)
    WAVE_SCAN_STEP_ALL(0x118, 0xf, true)
    WAVE_SCAN_STEP_ALL(0x142, 0xa, false)
    WAVE_SCAN_STEP_ALL(0x143, 0xc, false)
}
#undef WAVE_SCAN_STEP_ALL

template <int RP>
struct wave_lds {
    static constexpr int NPOS = RP * 64;
    static constexpr int RB = RP - 1;
    static constexpr int kLmax = NPOS - (2 * kPad + 7);  // 139 / 203 / 267
    static constexpr int kP0 = 40;                       // first position with a propensity slot
    static constexpr int PN = NPOS - 96;                 // slots: positions [40, 40 + PN) cover [51, Lmax + 60]
    static constexpr int nDoubles = 2 * PN;              // P+[v], P-[v-1]
    static constexpr int nWords = NPOS + 16;             // packed counts with 8 zero words either side
    static constexpr size_t bytes = (size_t)nDoubles * 8 + (size_t)nWords * 4;
    static_assert(kLmax + 60 < kP0 + PN, "propensity slots too short");
    static_assert(2 * PN >= RB * 64 + 48, "the z prefix must fit the propensity rows");
};

// one interval (tile `tile` of the table) by the calling wavefront
template <int RP>
__device__ __forceinline__ void wave_interval(wave_kargs *ka, kcoef *kc, double *smem, const int lane, const int64_t tile) {
    // The arguments are read where they are used, from the kernel-argument segment (scalar loads): the
    // pointer passes an empty asm at the head of every phase, so that the ~40 fields are not all loaded
    // ahead of the loop over the intervals and kept in (or spilled from) scalar registers across it
    WAVE_ARGS(a);
    typedef wave_lds<RP> LY;
    constexpr int RB = LY::RB, NPOS = LY::NPOS;
    double *PP = smem, *PM = smem + LY::PN, *Z = smem;
    u32 *pk = reinterpret_cast<u32 *>(smem + LY::nDoubles);

    // ---- geometry: one interval, never split (scalar loads)
    int64_t out_off, iv;
    int L;
    if (a.interval_off) {
        typedef const __attribute__((address_space(4))) fptk::lean_tile_rec krec;
        krec *r = (krec *)(a.tile_recs + tile);
        out_off = r->out_off;
        iv = r->iv;
        L = r->len;
    } else {
        iv = tile;
        L = a.interval_len;
        out_off = iv * (int64_t)L;
    }
    const int nc = L + 2 * kPad + 1;           // padded positions
    const int nrow_in = (nc + 6 + 63) >> 6;    // rows holding any input
    const int nrow_w = (nc + 63) >> 6;         // rows holding window sums
    const int prow_last = (L + 60) >> 6;       // last row with a propensity anybody reads
    const int srow_last = (L + 5) >> 6;        // last row in which a smoothing window starts
    const int nb_rows = (L + 6 + 63) >> 6;     // rows of base slots
    const double *gcp = a.counts_plus + (out_off + iv * (int64_t)(2 * kPad + 1));
    const double *gcm = a.counts_minus + (out_off + iv * (int64_t)(2 * kPad + 1));
    const uint8_t *gsq = a.seq + (out_off + iv * (int64_t)(2 * kPad + 7));

    // ---- A: every load of the interval in flight, then counts -> packed 16-bit integers ('-' one
    //      position to the right), sequence -> two bit planes in scalar registers
    double cp[RP], cm[RP];
    u32 ch[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int v = i * 64 + lane;
        cp[i] = cm[i] = 0.0;
        ch[i] = 'A';
        if (i < nrow_in) {
            if (v < nc) {
                cp[i] = gcp[v];
                cm[i] = gcm[v];
            }
            if (v < nc + 6) ch[i] = gsq[v];
        }
    }
    bool bad = false;
    if (lane < 8) {
        pk[lane] = 0;
        pk[8 + NPOS + lane] = 0;
    }
    unsigned long long m0[RP + 1], m1[RP + 1];
    m0[RP] = m1[RP] = 0;
    int im_carry = 0;  // the '-' count of the position before the row
    // what would make an input fall outside the kernel's case is OR-ed / MAX-ed up in vector registers and
    // tested once (a compare per condition and row would be a vector and a scalar instruction each):
    // the largest count as unsigned (negative ones are huge), the bits of count - (double)(int)count
    // (zero, or a signed zero, for an integer; NaN and infinities leave bits), the bases off A/C/G/T
    u32 cmax = 0, frac_bits = 0, seq_bad = 0;
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int v = i * 64 + lane;
        const int ip = (int)cp[i], im = (int)cm[i];
        const double ep = cp[i] - (double)ip, em = cm[i] - (double)im;
        frac_bits |= ((u32)__double2hiint(ep) & 0x7fffffffu) | (u32)__double2loint(ep);
        frac_bits |= ((u32)__double2hiint(em) & 0x7fffffffu) | (u32)__double2loint(em);
        cmax = max(cmax, max((u32)ip, (u32)im));
        const int ims = __builtin_amdgcn_update_dpp(im_carry, im, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        im_carry = __builtin_amdgcn_readlane(im, 63);
        pk[8 + v] = (u32)ip | ((u32)ims << 16);
        const u32 c = ch[i];
        // A, C, G, T (either case): bit (c & 0xDF) - 'A' of 0x80045; anything else has a bit outside it
        const u32 x = (c & 0xDFu) - (u32)'A';
        seq_bad |= (x >> 5) | (~(0x80045u >> (x & 31u)) & 1u);
        m0[i] = __ballot((c & 2u) != 0);
        m1[i] = __ballot((c & 4u) != 0);
    }
    bad |= (frac_bits | seq_bad) != 0 || cmax > kCountMax;
    wave_sync();

    if (LEAN_STOP(1)) return;
    // ---- B1: the propensities.  The table gathers of TWO rows are in flight at a time (eight registers;
    //      all rows at once held sixteen through the scans and cost a wavefront per SIMD): issued ahead of
    //      the scans of their rows, written to LDS (positions [40, 40 + PN)) behind them
    double2 tt[RP];
    auto gather = [&](int i) {
        tt[i] = make_double2(0.0, 0.0);
        if (i == 0 || i <= prow_last) {
            const bool low = lane < 32;
            const u32 sh = (u32)lane & 31u;
            const u32 a0 = low ? (u32)m0[i] : (u32)(m0[i] >> 32), b0 = low ? (u32)(m0[i] >> 32) : (u32)m0[i + 1];
            const u32 a1 = low ? (u32)m1[i] : (u32)(m1[i] >> 32), b1 = low ? (u32)(m1[i] >> 32) : (u32)m1[i + 1];
            const u32 f0 = __builtin_amdgcn_alignbit(b0, a0, sh) & 63u;
            const u32 f1 = __builtin_amdgcn_alignbit(b1, a1, sh) & 63u;
            tt[i] = a.table2[f0 | (f1 << 6)];  // (P+[v], P-[v-1])
        }
    };
    auto put = [&](int i) {
        if (i == 0 || i <= prow_last) {
            const int idx = i * 64 + lane - LY::kP0;
            if (idx >= 0 && idx < LY::PN) {
                PP[idx] = tt[i].x;
                PM[idx] = tt[i].y;
            }
        }
    };

    // ---- B2: 2*hw window sums and their scans, all in registers
    u32 Wk[RP], Ap[RP], Am[RP], pe_p[RP], pe_m[RP], se_p[RP], se_m[RP];
    u32 Tp[RP], Tm[RP];                       // extrema of a whole tile (scalar)
    u32 carry_p = 0, carry_m = 0, run_bits = 0;
    const int mirror = (63 - lane) << 2;
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        Wk[i] = Ap[i] = Am[i] = pe_p[i] = pe_m[i] = se_p[i] = se_m[i] = 0;
        Tp[i] = Tm[i] = 0;
        if ((i & 1) == 0) {
            gather(i);
            if (i + 1 < RP) gather(i + 1);
        }
        if (i < 2 || i < nrow_w) {  // (rows 0 and 1 always hold window sums: nc >= 112)
            const u32 *pw = pk + 8 + i * 64 + lane - kHW;
            u32 W = pw[0];
#pragma unroll
            for (int j = 1; j < 2 * kHW; ++j) W += pw[j];
            Wk[i] = W;
            const u32 wp = W & 0xffffu, wm = W >> 16;
            // prefix sums over the interval (a scalar carry from row to row) and, from row 1 on, prefix
            // extrema of the tile (a window never ENDS in row 0, and row 0 is never a middle tile):
            // (0xffff - min) | max << 16 per strand
            u32 xa[2] = {wp, wm}, xm[4] = {0xffffu - wp, wp, 0xffffu - wm, wm};
            if (i >= 1) wave_scan_multi<2, 4>(xa, xm);
            else wave_scan_multi<2, 0>(xa, xm);
            Ap[i] = xa[0] + carry_p;
            Am[i] = xa[1] + carry_m;
            carry_p = (u32)__builtin_amdgcn_readlane((int)Ap[i], 63);
            carry_m = (u32)__builtin_amdgcn_readlane((int)Am[i], 63);
            if (i >= 1) {
                pe_p[i] = xm[0] | (xm[1] << 16);
                pe_m[i] = xm[2] | (xm[3] << 16);
                Tp[i] = (u32)__builtin_amdgcn_readlane((int)pe_p[i], 63);
                Tm[i] = (u32)__builtin_amdgcn_readlane((int)pe_m[i], 63);
            }
            if (i == 0 || i <= srow_last) {  // suffix extrema: prefix scans of the mirrored tile, mirrored back
                const u32 Wr = (u32)__builtin_amdgcn_ds_bpermute(mirror, (int)W);
                const u32 rp = Wr & 0xffffu, rm = Wr >> 16;
                u32 xs[4] = {0xffffu - rp, rp, 0xffffu - rm, rm};
                wave_scan_multi<0, 4>(xs, xs);
                se_p[i] = (u32)__builtin_amdgcn_ds_bpermute(mirror, (int)(xs[0] | (xs[1] << 16)));
                se_m[i] = (u32)__builtin_amdgcn_ds_bpermute(mirror, (int)(xs[2] | (xs[3] << 16)));
            }
            // 33 equal non-zero window sums in a row?  (fpt_scan_lean.hip, phase B: the one case where
            // smoothing.h:61-69 is not S - min - max.)  Bit l of eP / eM: lanes l and l + 1 hold the same
            // non-zero sum (lane 63 is compared with 0).  A run of 33 needs 32 set bits in a row inside a
            // tile, or 33 over the ends of two tiles -- at least 16 in one of them: unless some tile has
            // 16 set bits (on dense counts hardly any has) there is nothing to look for.
            const u32 d = W ^ (u32)__builtin_amdgcn_update_dpp(0, (int)W, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
            const unsigned long long eP = __ballot((d & 0xffffu) < min(wp, 1u));   // equal and non-zero
            const unsigned long long eM = __ballot((d >> 16) < min(wm, 1u));
            run_bits = max(run_bits, (u32)__builtin_popcountll(eP | eM));
        }
        if ((i & 1) == 1 || i + 1 == RP) {
            if (i & 1) put(i - 1);
            put(i);
        }
        // rows are independent, and left alone the scheduler runs them side by side: 20 more registers,
        // two wavefronts per SIMD fewer.  Other wavefronts fill this one's gaps, not its own next row.
        __builtin_amdgcn_sched_barrier(0);
    }
    if (run_bits >= 16) {  // (scalar) the exact search: runs inside every tile, then runs across two tiles
        u32 w_first[RP], w_last[RP], w_runs[RP];
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const u32 W = Wk[i], wp = W & 0xffffu, wm = W >> 16;  // (rows without window sums hold 0: no runs)
            const u32 d = W ^ (u32)__builtin_amdgcn_update_dpp(0, (int)W, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
            constexpr unsigned long long kNotLast = 0x7fffffffffffffffull;  // lane 63 has no right neighbour here
            const unsigned long long eP = __ballot((d & 0xffffu) < min(wp, 1u)) & kNotLast;
            const unsigned long long eM = __ballot((d >> 16) < min(wm, 1u)) & kNotLast;
            unsigned long long rP = eP, rM = eM;
            rP &= rP >> 1; rP &= rP >> 2; rP &= rP >> 4; rP &= rP >> 8; rP &= rP >> 16;
            rM &= rM >> 1; rM &= rM >> 2; rM &= rM >> 4; rM &= rM >> 8; rM &= rM >> 16;
            bad |= (rP | rM) != 0;
            const u32 leadP = (u32)__builtin_ctzll(~eP) + 1u, trailP = (u32)__builtin_clzll(~(eP << 1)) + 1u;
            const u32 leadM = (u32)__builtin_ctzll(~eM) + 1u, trailM = (u32)__builtin_clzll(~(eM << 1)) + 1u;
            w_first[i] = (u32)__builtin_amdgcn_readfirstlane((int)W);
            w_last[i] = (u32)__builtin_amdgcn_readlane((int)W, 63);
            w_runs[i] = leadP | (leadM << 8) | (trailP << 16) | (trailM << 24);
        }
#pragma unroll
        for (int i = 0; i + 1 < RP; ++i) {
            const u32 last = w_last[i], first = w_first[i + 1], r0 = w_runs[i], r1 = w_runs[i + 1];
            const bool crossP = ((last ^ first) & 0xffffu) == 0 && (last & 0xffffu) != 0 && ((r0 >> 16) & 0xffu) + (r1 & 0xffu) >= 33u;
            const bool crossM = ((last ^ first) >> 16) == 0 && (last >> 16) != 0 && (r0 >> 24) + ((r1 >> 8) & 0xffu) >= 33u;
            bad |= crossP | crossM;
        }
    }
    wave_sync();

    if (LEAN_STOP(2)) {  // (keeps the scans alive)
        u32 x = 0;
#pragma unroll
        for (int i = 0; i < RP; ++i) x += Wk[i] + Ap[i] + Am[i] + pe_p[i] + pe_m[i] + se_p[i] + se_m[i] + Tp[i] + Tm[i];
        if (x == 0x12345678u) a.redo[tile] = 1;
        return;
    }
    // ---- C / D per base row: slot u = t + 6 is where the smoothing windows [u, u + 100] of BOTH strands
    //      of base t start ('+' at padded position 56 + t, '-' at 55 + t staged one to the right;
    //      detect.py:121-122); the window ends one row and 36 lanes further on
    WAVE_ARGS(ac);
    const int dm = ac.dm_ids ? uniform_load(ac.dm_ids, iv) : 0;
    const double2 *memo = ac.memo + (size_t)dm * ac.memo_exp * ac.memo_obs;
    double zrow[RB];
    const int fetch = ((lane + 36) & 63) << 2;
    const bool src_row1 = lane >= 36;   // as a SOURCE lane: serves a slot of lanes 0..27 from row + 1
    const bool three = lane >= 28;      // as a slot: the window spans three tiles, the middle one whole
    bool miss = false;
    u32 miss_e = 0, miss_k = 0;
#pragma unroll
    for (int jb = 0; jb < RB; ++jb) {
        zrow[jb] = 0.0;
        if (jb < nb_rows) {
            constexpr int kLast = RP - 1;
            const int r1 = jb + 1, r2 = jb + 2 <= kLast ? jb + 2 : kLast;  // (a valid slot never reads beyond row RP - 1)
            const u32 hiAp = (u32)__builtin_amdgcn_ds_bpermute(fetch, (int)(src_row1 ? Ap[r1] : Ap[r2]));
            const u32 hiAm = (u32)__builtin_amdgcn_ds_bpermute(fetch, (int)(src_row1 ? Am[r1] : Am[r2]));
            const u32 hiEp = (u32)__builtin_amdgcn_ds_bpermute(fetch, (int)(src_row1 ? pe_p[r1] : pe_p[r2]));
            const u32 hiEm = (u32)__builtin_amdgcn_ds_bpermute(fetch, (int)(src_row1 ? pe_m[r1] : pe_m[r2]));
            const int u = jb * 64 + lane, t = u - 6;
            lean_owner o;
            o.t = t;
            o.L = L;
            o.out_off = out_off;
            o.mine = t >= 0 && t < L && !LEAN_STOP(6);
            lean_tracks tr;
            tr.ex = tr.pv = 0.0;
            tr.k = 0;
            if (t >= 0 && t < L) {
                double e2[2];
#pragma unroll
                for (int strand = 0; strand < 2; ++strand) {
                    const u32 w = strand ? Wk[jb] >> 16 : Wk[jb] & 0xffffu;
                    const u32 S = (strand ? hiAm - Am[jb] : hiAp - Ap[jb]) + w;  // inclusive prefix at u + 100 - exclusive at u
                    const u32 mid = three ? (strand ? Tm[r1] : Tp[r1]) : 0u;
                    const u32 ext = pk_max_u16(pk_max_u16(strand ? se_m[jb] : se_p[jb], strand ? hiEm : hiEp), mid);
                    const double tsum = (double)((S + (ext & 0xffffu)) - (ext >> 16) - 0xffffu);  // S - min - max
                    const double *P = (strand ? PM : PP) + (u + 45 - LY::kP0);  // positions u+45 .. u+54, centre u+50
                    double q = P[0];
#pragma unroll
                    for (int j = 1; j < 2 * kHW; ++j) q += P[j];  // left to right, like predict.h:43-47
                    const double q99 = mul_vs(q, kc->c99);
                    double r = __builtin_amdgcn_rcp(q99);
                    r = fma(fma(-q99, r, 1.0), r, r);
                    const double x = (P[kHW] * tsum) * r;  // ~ P/Q * t/99
                    const double e = floor(x + 0.5);
                    bad |= !(fabs(x - e) < fma(x, -kc->band, 0.5));  // too close to a tie (or not a number)
                    e2[strand] = e;
                }
                tr.ex = e2[0] + e2[1];
                const u32 kw = pk[8 + u + 50];
                tr.k = (kw & 0xffffu) + (kw >> 16);
                const u32 ei = (u32)(int)tr.ex;
                bool hit = ei < (u32)ac.memo_exp && tr.k < (u32)ac.memo_obs;
                double2 pz = memo[hit ? ei * (u32)ac.memo_obs + tr.k : 0u];
                if (!hit && ac.memo2) {  // (rare: hotspots) the kept second-level table, as far as it is filled
                    const int h0 = ac.memo2_have[0], h1 = ac.memo2_have[1];
                    if (h0 >= 0 && h1 >= 0 && ei <= (u32)h0 && tr.k <= (u32)h1) {
                        pz = ac.memo2[((size_t)dm * ac.miss_rows + ei) * ac.miss_stride + tr.k];
                        hit = true;
                    }
                }
                tr.pv = pz.x;
                zrow[jb] = pz.y;
                bad |= !hit | ((__double2hiint(pz.y) & 0x7ff00000) == 0x7ff00000);  // a miss, or a non-finite z
                if (!hit && ac.miss_max && ei < (u32)ac.miss_rows && tr.k < (u32)ac.miss_stride) {
                    miss = true;
                    miss_e = max(miss_e, ei);
                    miss_k = max(miss_k, tr.k);
                }
            }
            lean_store_tracks(ac, o, tr);
        }
    }
    // the largest missed pair sizes the second-level table of the redo pass: one pair of atomics per
    // wavefront that has a miss
    if (__builtin_amdgcn_ballot_w64(miss)) {
        int me = miss ? (int)miss_e : -1, mk = miss ? (int)miss_k : -1;
#pragma unroll
        for (int d = 32; d; d >>= 1) {
            me = max(me, __shfl_xor(me, d));
            mk = max(mk, __shfl_xor(mk, d));
        }
        if (lane == 0) {
            atomicMax(&ac.miss_max[0], me);
            atomicMax(&ac.miss_max[1], mk);
        }
    }

    // ---- E: Stouffer windows (windowing.h:53-84); Z (indexed by slot) takes the propensities' place
    WAVE_ARGS(ae);
    constexpr int NT = RB * 64;
    constexpr int kEdge = NT + 32 + 15;
    if (ae.n_scales == 0 || LEAN_STOP(3)) {
    } else if (ae.n_scales == 1 && ae.max_scale <= 8) {
        wave_sync();
#pragma unroll
        for (int jb = 0; jb < RB; ++jb)
            if (jb < nb_rows) Z[16 + jb * 64 + lane] = zrow[jb];
        wave_sync();
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            if (jb < nb_rows) {
                const int u = jb * 64 + lane;
                lean_owner o;
                o.t = u - 6;
                o.L = L;
                o.out_off = out_off;
                o.mine = o.t >= 0 && o.t < L && !LEAN_STOP(6);
                bad |= lean_window_narrow<NT>(ae, kc, o, u, Z);
            }
        }
    } else {
        wave_sync();
        double carry = 0.0;
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            if (jb < nb_rows) {
                const double zr = wave_scan_f64(zrow[jb], 0.0, op_add()) + carry;
                carry = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(zr), 63),
                                         __builtin_amdgcn_readlane(__double2loint(zr), 63));
                Z[16 + jb * 64 + lane] = zr;
            }
        }
        if (lane == 0) {
            Z[15] = 0.0;
            Z[kEdge] = -1e4;
        }
        wave_sync();
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            if (jb < nb_rows) {
                const int u = jb * 64 + lane;
                lean_owner o;
                o.t = u - 6;
                o.L = L;
                o.out_off = out_off;
                o.mine = o.t >= 0 && o.t < L && !LEAN_STOP(6);
                bad |= lean_windows<NT>(ae, kc, o, u, Z);
            }
        }
    }
    if (bad) ae.redo[tile] = 1;
}

// ============================================================================================
// The same interval-per-wavefront scan as a SOFTWARE PIPELINE over the intervals of a wavefront.
//
// What a short interval costs is not its ~1,400 instructions but the memory trips around them: per-phase
// timing of k_scan_wave (FPT_ABLATE stops) shows 0.40 ms of a 0.60 ms launch with the stores
// ablated -- a wavefront waits ~2 us for its inputs and, at its end, 3-4 us for the acknowledgement of
// its stores before its slot is free; k_scan_lean's short classes lose the same third.  The vector-memory
// counter of gfx9 is in order across loads AND stores, so a wavefront that simply goes on to the next
// interval stops at its first load for the stores of the one before.  Here the order of the memory
// operations is arranged so that nothing young waits for them:
//
//   iteration t:   scans(t)  [table gathers(t) land]  ->  (exp, obs) gathers(t), all rows, consumed
//                  ->  windows(t) in registers
//                  ->  stage(t + 1) from the inputs loaded an iteration ago, table gathers(t + 1) issued
//                  ->  loads(t + 2) issued
//                  ->  stores(t), all of them, last
//
// The inputs of t + 1 are older than the stores of t - 1 they would otherwise queue behind; the table
// gathers of t + 1 are older than the stores of t; only the (exp, obs) gathers of t + 1 are younger than
// stores -- issued two microseconds earlier.  Every vector-memory operation of an iteration is issued on
// every path (buffer loads and stores: a lane without a position or a base carries an offset beyond the
// buffer and the hardware drops it; rows an interval does not need are issued empty), so the compiler's
// wait counts are exact: `vmcnt(n)` with n the stores in flight, not `vmcnt(0)`.
// For the `detect` shape: one narrow Stouffer scale, all four tracks wanted.
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
constexpr u32 kNoLane = 0xffffffffu;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void *base, u32 n_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)n_bytes, 0x00020000);
}
__device__ __forceinline__ void buffer_store_f64(__amdgpu_buffer_rsrc_t r, u32 byte_off, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ double buffer_load_f64(__amdgpu_buffer_rsrc_t r, u32 byte_off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}

struct wave_geo {
    int64_t out_off, tile;
    int L, dm;
    const double *gcp, *gcm;
    const uint8_t *gsq;
};
__device__ __forceinline__ wave_geo wave_geometry(wave_kargs *ka, int64_t tile) {
    WAVE_ARGS(a);
    wave_geo g;
    int64_t iv;
    if (a.interval_off) {
        typedef const __attribute__((address_space(4))) fptk::lean_tile_rec krec;
        krec *r = (krec *)(a.tile_recs + tile);
        g.out_off = r->out_off;
        iv = r->iv;
        g.L = r->len;
    } else {
        iv = tile;
        g.L = a.interval_len;
        g.out_off = iv * (int64_t)g.L;
    }
    g.tile = tile;
    g.gcp = a.counts_plus + (g.out_off + iv * (int64_t)(2 * kPad + 1));
    g.gcm = a.counts_minus + (g.out_off + iv * (int64_t)(2 * kPad + 1));
    g.gsq = a.seq + (g.out_off + iv * (int64_t)(2 * kPad + 7));
    g.dm = a.dm_ids ? uniform_load(a.dm_ids, iv) : 0;
    return g;
}

template <int RP>
struct wave_in {
    double cp[RP], cm[RP];
    u32 ch[RP];
};
// every load of an interval: 3 RP buffer loads, whatever its length (positions beyond it read 0)
template <int RP>
__device__ __forceinline__ void wave_issue(const wave_geo &g, int lane, wave_in<RP> &in) {
    const u32 nc = (u32)(g.L + 2 * kPad + 1);
    const __amdgpu_buffer_rsrc_t rp = buffer_of(g.gcp, nc * 8u), rm = buffer_of(g.gcm, nc * 8u), rs = buffer_of(g.gsq, nc + 6u);
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const u32 v = (u32)(i * 64 + lane);
        in.cp[i] = buffer_load_f64(rp, v * 8u);
        in.cm[i] = buffer_load_f64(rm, v * 8u);
        in.ch[i] = (u32)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)v, 0, 0);
    }
}

// counts -> packed 16-bit integers in LDS ('-' one position to the right), sequence -> bit planes
template <int RP>
__device__ __forceinline__ bool wave_stage(const wave_geo &g, int lane, const wave_in<RP> &in, u32 *pk,
                                           unsigned long long (&m0)[RP + 1], unsigned long long (&m1)[RP + 1]) {
    constexpr int NPOS = RP * 64;
    const int nseq = g.L + 2 * kPad + 7;
    if (lane < 8) {
        pk[lane] = 0;
        pk[8 + NPOS + lane] = 0;
    }
    m0[RP] = m1[RP] = 0;
    int im_carry = 0;
    u32 cmax = 0, frac_bits = 0, seq_bad = 0;
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int v = i * 64 + lane;
        const int ip = (int)in.cp[i], im = (int)in.cm[i];
        const double ep = in.cp[i] - (double)ip, em = in.cm[i] - (double)im;
        frac_bits |= ((u32)__double2hiint(ep) & 0x7fffffffu) | (u32)__double2loint(ep);
        frac_bits |= ((u32)__double2hiint(em) & 0x7fffffffu) | (u32)__double2loint(em);
        cmax = max(cmax, max((u32)ip, (u32)im));
        const int ims = __builtin_amdgcn_update_dpp(im_carry, im, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        im_carry = __builtin_amdgcn_readlane(im, 63);
        pk[8 + v] = (u32)ip | ((u32)ims << 16);
        const u32 c = v < nseq ? in.ch[i] : (u32)'A';  // (beyond the sequence the buffer load gave 0)
        const u32 x = (c & 0xDFu) - (u32)'A';
        seq_bad |= (x >> 5) | (~(0x80045u >> (x & 31u)) & 1u);
        m0[i] = __ballot((c & 2u) != 0);
        m1[i] = __ballot((c & 4u) != 0);
    }
    return (frac_bits | seq_bad) != 0 || cmax > kCountMax;
}

template <int RP>
__device__ __forceinline__ void wave_gather(const double2 *table2, int lane, const unsigned long long (&m0)[RP + 1],
                                            const unsigned long long (&m1)[RP + 1], double2 (&tt)[RP]) {
    const bool low = lane < 32;
    const u32 sh = (u32)lane & 31u;
#pragma unroll
    for (int i = 0; i < RP; ++i) {  // every row: RP gathers on every path
        const u32 a0 = low ? (u32)m0[i] : (u32)(m0[i] >> 32), b0 = low ? (u32)(m0[i] >> 32) : (u32)m0[i + 1];
        const u32 a1 = low ? (u32)m1[i] : (u32)(m1[i] >> 32), b1 = low ? (u32)(m1[i] >> 32) : (u32)m1[i + 1];
        const u32 f0 = __builtin_amdgcn_alignbit(b0, a0, sh) & 63u;
        const u32 f1 = __builtin_amdgcn_alignbit(b1, a1, sh) & 63u;
        tt[i] = table2[f0 | (f1 << 6)];
    }
}

template <int RP>
__global__ void __launch_bounds__(64) k_scan_wave_pipe(const lean_args a_) {
    extern __shared__ double smem[];
    typedef wave_lds<RP> LY;
    constexpr int RB = LY::RB;
    double *PP = smem, *PM = smem + LY::PN, *Z = smem;
    u32 *pk = reinterpret_cast<u32 *>(smem + LY::nDoubles);
    wave_kargs *ka = (wave_kargs *)__builtin_amdgcn_kernarg_segment_ptr();
    kcoef *kc = &ka->c;
    const int lane0 = threadIdx.x;
    const int per = ka->tiles_per_wave;
    const int64_t begin = ka->tile_first + (int64_t)blockIdx.x * per;
    const int64_t end = min(ka->tile_first + ka->tile_count, begin + per);
    if (begin >= end) return;

    // ---- prologue: the first interval staged, its table gathers and the second interval's loads in flight
    wave_in<RP> in;
    unsigned long long m0[RP + 1], m1[RP + 1];
    double2 tt[RP];
    wave_geo g = wave_geometry(ka, begin), gn = g;
    wave_issue<RP>(g, lane0, in);
    bool bad = wave_stage<RP>(g, lane0, in, pk, m0, m1);
    wave_sync();
    wave_gather<RP>(ka->table2, lane0, m0, m1, tt);
    // (Every path issues the same vector-memory operations in the same order, or the compiler's wait for the
    // table gathers below falls back to the shortest path's count and drains the stores after all: a
    // wavefront with one interval loads it twice, the last iteration stages its own interval again, and
    // the prologue issues the iteration's 4 RB + 1 stores with every lane out of range.)
    gn = wave_geometry(ka, min(begin + 1, end - 1));
    wave_issue<RP>(gn, lane0, in);
    {
        const __amdgpu_buffer_rsrc_t none = buffer_of(ka->exp_out, 0u);
#pragma unroll
        for (int j = 0; j < 4 * RB; ++j) buffer_store_f64(none, kNoLane, 0.0);
        __builtin_amdgcn_raw_buffer_store_b32(0, none, (int)kNoLane, 0, 0);
    }

    for (int64_t t = begin; t < end; ++t) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));  // (see k_scan_wave: lane-derived values are not kept across the loop)
        const int L = g.L;
        const int nc = L + 2 * kPad + 1;
        const int nrow_w = (nc + 63) >> 6, srow_last = (L + 5) >> 6, nb_rows = (L + 6 + 63) >> 6;

        // ---- B2: window sums and scans (registers); the propensities to LDS when their gathers have landed
        u32 Wk[RP], Ap[RP], Am[RP], pe_p[RP], pe_m[RP], se_p[RP], se_m[RP], Tp[RP], Tm[RP];
        u32 carry_p = 0, carry_m = 0, run_bits = 0;
        const int mirror = (63 - lane) << 2;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            Wk[i] = Ap[i] = Am[i] = pe_p[i] = pe_m[i] = se_p[i] = se_m[i] = 0;
            Tp[i] = Tm[i] = 0;
            if (i < 2 || i < nrow_w) {
                const u32 *pw = pk + 8 + i * 64 + lane - kHW;
                u32 W = pw[0];
#pragma unroll
                for (int j = 1; j < 2 * kHW; ++j) W += pw[j];
                Wk[i] = W;
                const u32 wp = W & 0xffffu, wm = W >> 16;
                u32 xa[2] = {wp, wm}, xm[4] = {0xffffu - wp, wp, 0xffffu - wm, wm};
                if (i >= 1) wave_scan_multi<2, 4>(xa, xm);
                else wave_scan_multi<2, 0>(xa, xm);
                Ap[i] = xa[0] + carry_p;
                Am[i] = xa[1] + carry_m;
                carry_p = (u32)__builtin_amdgcn_readlane((int)Ap[i], 63);
                carry_m = (u32)__builtin_amdgcn_readlane((int)Am[i], 63);
                if (i >= 1) {
                    pe_p[i] = xm[0] | (xm[1] << 16);
                    pe_m[i] = xm[2] | (xm[3] << 16);
                    Tp[i] = (u32)__builtin_amdgcn_readlane((int)pe_p[i], 63);
                    Tm[i] = (u32)__builtin_amdgcn_readlane((int)pe_m[i], 63);
                }
                if (i == 0 || i <= srow_last) {
                    const u32 Wr = (u32)__builtin_amdgcn_ds_bpermute(mirror, (int)W);
                    const u32 rp = Wr & 0xffffu, rm = Wr >> 16;
                    u32 xs[4] = {0xffffu - rp, rp, 0xffffu - rm, rm};
                    wave_scan_multi<0, 4>(xs, xs);
                    se_p[i] = (u32)__builtin_amdgcn_ds_bpermute(mirror, (int)(xs[0] | (xs[1] << 16)));
                    se_m[i] = (u32)__builtin_amdgcn_ds_bpermute(mirror, (int)(xs[2] | (xs[3] << 16)));
                }
                const u32 d = W ^ (u32)__builtin_amdgcn_update_dpp(0, (int)W, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
                const unsigned long long eP = __ballot((d & 0xffffu) < min(wp, 1u));
                const unsigned long long eM = __ballot((d >> 16) < min(wm, 1u));
                run_bits = max(run_bits, (u32)__builtin_popcountll(eP | eM));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (run_bits >= 16) bad = true;  // (the exact search of k_scan_wave is not repeated here: such an interval is rare on
                                         // dense counts, and on sparse ones the redo pass settles it)
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int idx = i * 64 + lane - LY::kP0;
            if (idx >= 0 && idx < LY::PN) {
                PP[idx] = tt[i].x;
                PM[idx] = tt[i].y;
            }
        }
        wave_sync();

        // ---- C / D: all rows -- expected counts, then every (exp, obs) gather issued before the first is used
        WAVE_ARGS(ac);
        const double2 *memo = ac.memo + (size_t)g.dm * ac.memo_exp * ac.memo_obs;
        const int fetch = ((lane + 36) & 63) << 2;
        const bool src_row1 = lane >= 36, three = lane >= 28;
        double ex[RB], pv[RB], zrow[RB], pw[RB];
        u32 kk[RB], ei_[RB];
        double2 pz[RB];
        bool hit[RB];
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            constexpr int kLast = RP - 1;
            const int r1 = jb + 1, r2 = jb + 2 <= kLast ? jb + 2 : kLast;
            ex[jb] = 0.0;
            kk[jb] = 0;
            if (jb == 0 || jb < nb_rows) {
                const u32 hiAp = (u32)__builtin_amdgcn_ds_bpermute(fetch, (int)(src_row1 ? Ap[r1] : Ap[r2]));
                const u32 hiAm = (u32)__builtin_amdgcn_ds_bpermute(fetch, (int)(src_row1 ? Am[r1] : Am[r2]));
                const u32 hiEp = (u32)__builtin_amdgcn_ds_bpermute(fetch, (int)(src_row1 ? pe_p[r1] : pe_p[r2]));
                const u32 hiEm = (u32)__builtin_amdgcn_ds_bpermute(fetch, (int)(src_row1 ? pe_m[r1] : pe_m[r2]));
                const int u = jb * 64 + lane, tb = u - 6;
                if (tb >= 0 && tb < L) {
                    double e2[2];
#pragma unroll
                    for (int strand = 0; strand < 2; ++strand) {
                        const u32 w = strand ? Wk[jb] >> 16 : Wk[jb] & 0xffffu;
                        const u32 S = (strand ? hiAm - Am[jb] : hiAp - Ap[jb]) + w;
                        const u32 mid = three ? (strand ? Tm[r1] : Tp[r1]) : 0u;
                        const u32 ext = pk_max_u16(pk_max_u16(strand ? se_m[jb] : se_p[jb], strand ? hiEm : hiEp), mid);
                        const double tsum = (double)((S + (ext & 0xffffu)) - (ext >> 16) - 0xffffu);
                        const double *P = (strand ? PM : PP) + (u + 45 - LY::kP0);
                        double q = P[0];
#pragma unroll
                        for (int j = 1; j < 2 * kHW; ++j) q += P[j];
                        const double q99 = mul_vs(q, kc->c99);
                        double r = __builtin_amdgcn_rcp(q99);
                        r = fma(fma(-q99, r, 1.0), r, r);
                        const double x = (P[kHW] * tsum) * r;
                        const double e = floor(x + 0.5);
                        bad |= !(fabs(x - e) < fma(x, -kc->band, 0.5));
                        e2[strand] = e;
                    }
                    ex[jb] = e2[0] + e2[1];
                    const u32 kw = pk[8 + u + 50];
                    kk[jb] = (kw & 0xffffu) + (kw >> 16);
                }
            }
            ei_[jb] = (u32)(int)ex[jb];
            hit[jb] = ei_[jb] < (u32)ac.memo_exp && kk[jb] < (u32)ac.memo_obs;
            pz[jb] = memo[hit[jb] ? ei_[jb] * (u32)ac.memo_obs + kk[jb] : 0u];  // RB gathers on every path
        }
        bool miss = false;
        u32 miss_e = 0, miss_k = 0;
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            const int tb = jb * 64 + lane - 6;
            const bool valid = tb >= 0 && tb < L;
            if (valid && !hit[jb] && ac.memo2) {
                const int h0 = ac.memo2_have[0], h1 = ac.memo2_have[1];
                if (h0 >= 0 && h1 >= 0 && ei_[jb] <= (u32)h0 && kk[jb] <= (u32)h1) {
                    pz[jb] = ac.memo2[((size_t)g.dm * ac.miss_rows + ei_[jb]) * ac.miss_stride + kk[jb]];
                    hit[jb] = true;
                }
            }
            pv[jb] = valid ? pz[jb].x : 0.0;
            zrow[jb] = valid ? pz[jb].y : 0.0;
            if (valid) {
                bad |= !hit[jb] | ((__double2hiint(pz[jb].y) & 0x7ff00000) == 0x7ff00000);
                if (!hit[jb] && ac.miss_max && ei_[jb] < (u32)ac.miss_rows && kk[jb] < (u32)ac.miss_stride) {
                    miss = true;
                    miss_e = max(miss_e, ei_[jb]);
                    miss_k = max(miss_k, kk[jb]);
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(miss)) {
            int me = miss ? (int)miss_e : -1, mk = miss ? (int)miss_k : -1;
#pragma unroll
            for (int d = 32; d; d >>= 1) {
                me = max(me, __shfl_xor(me, d));
                mk = max(mk, __shfl_xor(mk, d));
            }
            if (lane == 0) {
                atomicMax(&ac.miss_max[0], me);
                atomicMax(&ac.miss_max[1], mk);
            }
        }

        // ---- E: the one narrow Stouffer window, its p-value kept in registers
        WAVE_ARGS(ae);
        wave_sync();
#pragma unroll
        for (int jb = 0; jb < RB; ++jb)
            if (jb == 0 || jb < nb_rows) Z[16 + jb * 64 + lane] = zrow[jb];
        wave_sync();
        const int hs = ae.scales[0];
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            pw[jb] = 1.0;
            if (jb == 0 || jb < nb_rows) {
                const int u = jb * 64 + lane, tb = u - 6;
                const bool inside = tb >= hs && tb < L - hs;
                double sv = 0.0;
                if (inside)
                    for (int j = 16 + u - hs; j <= 16 + u + hs; ++j) sv += Z[j];
                const double arg = inside ? -(sv * ae.scale_rsqrt[0]) : 1e3;  // edges are 1.0 (windowing.pyx:51)
                pw[jb] = ndtr_fast_s(arg, kc);
                bad |= inside && !(arg > -kc->limit);
            }
        }
        wave_sync();  // (the next interval's staging writes the packed counts this one read)

        // ---- the next interval staged, its table gathers and the loads of the one after it in flight
        const wave_geo gcur = g;
        const bool bad_next = wave_stage<RP>(gn, lane, in, pk, m0, m1);  // (the last iteration: its own interval again)
        wave_sync();
        wave_gather<RP>(ae.table2, lane, m0, m1, tt);
        g = gn;
        gn = wave_geometry(ka, min(t + 2, end - 1));
        wave_issue<RP>(gn, lane, in);
        __builtin_amdgcn_sched_barrier(0);

        // ---- stores of this interval, last: 4 RB + 1 buffer stores on every path
        {
            const u32 n8 = (u32)gcur.L * 8u;
            const __amdgpu_buffer_rsrc_t re = buffer_of(ae.exp_out + gcur.out_off, n8), ro = buffer_of(ae.obs_out + gcur.out_off, n8),
                                         rq = buffer_of(ae.pval_out + gcur.out_off, n8), rw = buffer_of(ae.winp_out + gcur.out_off, n8);
#pragma unroll
            for (int jb = 0; jb < RB; ++jb) {
                const int tb = jb * 64 + lane - 6;
                const u32 o8 = (tb >= 0 && tb < gcur.L) ? (u32)tb * 8u : kNoLane;  // (the pipeline is not launched in ablation runs)
                buffer_store_f64(re, o8, ex[jb]);
                buffer_store_f64(ro, o8, (double)kk[jb]);
                buffer_store_f64(rq, o8, pv[jb]);
                buffer_store_f64(rw, o8, pw[jb]);
            }
            // the redo flag of the interval (the array was zeroed by this call): lane 0 of a flagged interval
            const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
            const __amdgpu_buffer_rsrc_t rr = buffer_of(ae.redo + gcur.tile, 4u);
            __builtin_amdgcn_raw_buffer_store_b32(1, rr, (int)((any_bad && lane == 0) ? 0u : kNoLane), 0, 0);
        }
        bad = bad_next;
    }
}

// Wavefronts stay and walk the table: a launch of one wavefront per interval is bound by the rate at
// which the dispatcher starts single-wavefront workgroups (measured: 2.8 ns per workgroup, 1.6 resident
// wavefronts per SIMD on average -- the kernel then runs no faster than k_scan_lean).  The grid is what
// the device holds at once; wavefront b takes tiles_per_wave consecutive tiles (neighbours in the
// table are neighbours in memory more often than not: the 128-byte lines two intervals share are
// read by the same wavefront a few microseconds apart).
template <int RP>
__global__ void __launch_bounds__(64) k_scan_wave(const lean_args a) {
    extern __shared__ double smem[];
    wave_kargs *ka = (wave_kargs *)__builtin_amdgcn_kernarg_segment_ptr();
    kcoef *kc = &ka->c;
    const int lane = threadIdx.x;
    const int per = ka->tiles_per_wave;
    const int64_t begin = ka->tile_first + (int64_t)blockIdx.x * per;
    const int64_t end = min(ka->tile_first + ka->tile_count, begin + per);
    for (int64_t t = begin; t < end; ++t) {
        // (the lane number passes an empty asm too: everything derived from it -- LDS addresses, fetch
        // lanes, masks -- is otherwise computed once ahead of the loop and held in ~20 vector registers)
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        wave_interval<RP>(ka, kc, smem, lane_t, t);
        wave_sync();  // the next interval's staging overwrites what this one's windows read
    }
}

typedef void (*wave_kernel_t)(const lean_args);
wave_kernel_t wave_kernel(int rp, bool pipe) {
    switch (rp) {
        case 4: return pipe ? k_scan_wave_pipe<4> : k_scan_wave<4>;
        case 5: return pipe ? k_scan_wave_pipe<5> : k_scan_wave<5>;
        default: return pipe ? k_scan_wave_pipe<6> : k_scan_wave<6>;
    }
}

}  // namespace

namespace fptk {

int scan_wave_max_len(int rp) { return rp * 64 - (2 * kPad + 7); }

size_t scan_wave_lds_bytes(int rp) {
    switch (rp) {
        case 4: return wave_lds<4>::bytes;
        case 5: return wave_lds<5>::bytes;
        default: return wave_lds<6>::bytes;
    }
}

void launch_scan_wave(hipStream_t st, int rp, int n_tiles, const scan_launch &sl) {
    lean_args a;
    fill_lean_args(sl, a);
    // the pipelined form for the `detect` shape (one narrow scale, every track wanted): wavefronts walk
    // runs of consecutive intervals; FPT_WAVE_PIPE=0 keeps one interval after the other
    bool pipe = sl.n_scales == 1 && sl.scales[0] <= 8 && sl.exp_out && sl.obs_out && sl.pval_out && sl.winp_out && !sl.ablate;
    if (const char *e = getenv("FPT_WAVE_PIPE")) pipe = pipe && atoi(e) != 0;
    // intervals per wavefront: few without the pipeline (the dispatcher then balances the load: a grid of
    // as many wavefronts as the device holds ended with 29 % of the slots idle), a run long enough to
    // amortise its first loads and last stores with it
    int tpw = pipe ? 8 : 1;
    if (const char *e = getenv("FPT_WAVE_TPW")) tpw = atoi(e) > 0 ? atoi(e) : tpw;
    a.tile_count = n_tiles;
    a.tiles_per_wave = tpw;
    const int grid = (int)((n_tiles + a.tiles_per_wave - 1) / a.tiles_per_wave);
    hipLaunchKernelGGL(wave_kernel(rp, pipe), dim3(grid), dim3(64), scan_wave_lds_bytes(rp), st, a);
}

}  // namespace fptk
