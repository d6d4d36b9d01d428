// fpt_scan_lean.hip -- the first pass of the fused scan in memo mode (gfx950 / CDNA4, wave64).
//
// Same path as k_scan_fused (cli/detect.py:120-130 for one tile of one interval per workgroup,
// one output base per lane), written for the case every real batch is made of -- cut counts that
// are small non-negative integers, A/C/G/T bases, (exp, obs) pairs inside the memo table, the
// `detect` window widths hw = 5 / shw = 50 / clip 0.01 -- and for nothing else.  A tile that steps
// outside that case is marked in `redo` and computed again by the general kernel
// (k_scan_fused<..., REDO>), launched right behind on the same stream; whatever this kernel wrote
// for such a tile is overwritten there.  No early exit is needed: every out-of-case input still
// indexes inside its arrays.
//
// What keeps the instruction count down (the general kernel is bound by vector-instruction issue:
// ~1,400 per base on five scales; measured, profiles/):
//   * counts live in LDS as one 32-bit word per position (strand '+' | strand '-' << 16), so a
//     2*hw window sum of both strands is nine packed integer adds (exact: window sums of integers
//     do not depend on the order), and every scan below runs on 32-bit integers on the DPP path;
//   * the sequence lives in LDS as two bit planes (bits 1 and 2 of the ASCII code: A=00 C=01 T=10
//     G=11, case-insensitive) written straight from two wavefront ballots; the 6-mer of a position
//     is six bits of each plane, and the bias table is re-indexed by that 12-bit number with the
//     forward and the reverse-complement propensity side by side, so one 16-byte gather per
//     position replaces seven byte loads, two index computations and two gathers;
//   * the trimmed sum of a smoothing window is S - min - max from per-64-position-tile prefix sums
//     and prefix / suffix extrema; the one case where smoothing.h:61-69 gives something else
//     (2nd smallest == 2nd largest: at least 99 of 101 values equal and non-zero) needs a run of
//     33 equal non-zero values -- two ballots and a dozen scalar instructions per tile for the runs
//     inside a 64-position tile, three words of LDS per tile for those across two -- and such a tile
//     goes to the general kernel.  (Round 2 took any aligned row of 16 equal non-zero values for
//     the sign: sparse counts, where a single cut makes ten equal window sums, tripped it in 16-47 %
//     of the tiles, tools/diag_sparse_redo.py; with 33 it is 0.01-0.5 %);
//   * expected = round(P/Q * t/99): Q is summed in the reference's order, the two divisions are
//     replaced by one reciprocal with one Newton step, and a base whose product lies within
//     1e-13 (relative) of a half-integer -- where the rounding of the exact operations could
//     decide -- sends its tile to the general kernel;
//   * Stouffer windows from one workgroup-wide prefix sum of z; the normal cdf by the one-formula
//     evaluation of fpt_math.hpp (ndtr_fast) with every polynomial coefficient held in scalar
//     registers: the compiler's own choice for `acc * x + c` with a 64-bit constant is two
//     v_mov_b32 and a v_fmac_f64, three vector instructions per Horner step instead of one.
#include "fpt_lean_common.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace fptd;
using namespace fptlean;

namespace {

// the table of fptm::ndtr_fast_tab (this translation unit's copy: staged in LDS per workgroup)
__device__ const double g_lean_gtab[4 * FPT_NDTR_GTAB_N + 1] = {FPT_NDTR_GTAB_LIST};

// LDS carve-up for a tile of up to NT bases (NT = lanes x bases per lane: one base per lane in every class but
// the two-bases-per-lane instance of the largest): positions are padded up to NCR = NT + 128
// (NT output bases + 2*pad + 1 padded positions + 6 sequence bases, in whole 64-position tiles)
template <int NT>
struct lean_lds {
    static constexpr int NCR = NT + 128;
    static constexpr int NROW = NT / 16;
    // doubles.  The propensities are kept for the positions phase C reads them at only -- [pad + 1 - hw, pad + nt + hw],
    // stored from kPOff on: NT + 16 slots instead of NCR (the 128-lane class then fits sixteen workgroups on a CU,
    // as many as its wavefront slots hold, instead of thirteen)
    static constexpr int kPOff = kPad + 1 - kHW - 3;   // position v sits in slot v - kPOff (slot 3 is the first in use)
    static constexpr int NPP = NT + 16;
    static constexpr int oPP = 0;                  // P+[v]
    static constexpr int oPM = oPP + NPP;          // P-[v-1]
    static constexpr int oRT = oPM + NPP;          // row totals (NROW), then row carries (NROW + 4)
    // The table form of the normal cdf's g (fptm::ndtr_fast_tab: 128 cubics, 4 KB; two 16-byte LDS reads and 9
    // instructions in place of the 15 fp64 ones of the Horner chain -- and an fp64 instruction takes two issue slots
    // on gfx950).  Round 5 kept the table in 8 KB of its own (256 cubics), a third of the residency: no gain then.
    // Round 6: the table lives in the scan arrays, which nobody reads after phase D (behind the z prefix sums: oGT
    // below) -- no LDS of its own -- and with the wave priority of phase A the kernel runs at 0.89 of vector issue,
    // where an instruction saved is time saved.
    static constexpr bool kTab = true;
    static constexpr int oTB = oRT + 2 * NROW + 4;
    static constexpr int nDoubles = oTB;
    static_assert((oTB % 2) == 0, "the 32-bit words start on a 16-byte boundary");
    // 32-bit words, after the doubles
    static constexpr int oB0 = 0;                  // sequence bit planes
    static constexpr int oB1 = oB0 + NCR / 32 + 4;
    static constexpr int oPK = oB1 + NCR / 32 + 4; // packed counts with 8 zero words either side
    static constexpr int oSP = oPK + NCR + 16;     // tile prefix sums of W+, W-
    static constexpr int oSM = oSP + NCR;
    static constexpr int oXP = oSM + NCR;          // W+: (0xffff - prefix min) | prefix max << 16
    static constexpr int oXPs = oXP + NCR;         //     same for the suffixes
    static constexpr int oXM = oXPs + NCR;
    static constexpr int oXMs = oXM + NCR;
    static constexpr int oEG = oXMs + NCR;         // per 64-position tile: first W, last W, -, -, the two equal-neighbour masks
    static constexpr int nWords = oEG + 8 * (NCR / 64) + 4;
    // the prefix sums of z (phase E, NT + 48 doubles) take the place of the six scan arrays, which
    // nobody reads after phase D: 6 * NCR words >= 2 * (NT + 48)
    static constexpr int oZB = oSP;                // in words; 8-byte aligned: see the static_assert
    static_assert((oSP % 2) == 0 && 6 * NCR >= 2 * (NT + 48), "z prefix must fit the scan arrays, aligned");
    // ... and behind them the table of g (phase E, several scales): 4 doubles per interval, read 16 bytes at a time
    static constexpr int oGT = (oZB + 2 * (NT + 48) + 3) & ~3;
    static_assert(oGT + 8 * FPT_NDTR_GTAB_N <= oEG, "the table of g must fit the scan arrays behind the z prefix");
    static constexpr int kGtPerLane = (4 * FPT_NDTR_GTAB_N + NT - 1) / NT;   // table doubles a lane carries to phase E
    static constexpr size_t bytes = (size_t)nDoubles * 8 + (size_t)nWords * 4;
};

// sum / extrema of the 101 positions [lo, hi] (hi = lo + 100) from the per-tile scans: the window
// starts in tile q0 = lo >> 6 and ends one or two tiles later
__device__ __forceinline__ u32 window_sum(const u32 *ps, int lo, int hi) {
    const int below = lo - 1, q = below >> 6;  // lo >= 5: below >= 0
    const int e1 = (q << 6) + 63, e2 = e1 + 64;
    const u32 s2 = ps[(hi >> 6) == q + 2 ? e2 : below];  // a middle tile, or a term that cancels
    return (ps[hi] - ps[below]) + ps[e1] + ((hi >> 6) == q + 2 ? s2 : 0u);
}

// geometry of one tile: output bases [t0, t0+tl) of interval iv, widened by the largest Stouffer
// half-width H on both sides for the p-values its windows need
struct lean_tile {
    int64_t out_off;  // offset of the interval in the output tracks
    int t0, tl, L, ta, nt, nc, ncs;
    const double *gcp, *gcm;
    const uint8_t *gsq;
    int dm;
};

template <typename Args>
__device__ __forceinline__ lean_tile lean_geometry(const Args &a, int64_t tile) {
    lean_tile g;
    int64_t iv;
    if (a.interval_off) {
        typedef const __attribute__((address_space(4))) fptk::lean_tile_rec krec;
        krec *r = (krec *)(a.tile_recs + tile);
        g.out_off = r->out_off;
        iv = r->iv;
        g.t0 = r->t0;
        g.tl = r->tl;
        g.L = r->len;
    } else {
        if (a.tiles_per_interval == 1) {  // a 64-bit scalar division costs ~150 instructions
            iv = tile;
            g.t0 = 0;
        } else if ((tile >> 32) == 0) {
            const uint32_t q = (uint32_t)tile / (uint32_t)a.tiles_per_interval;
            iv = q;
            g.t0 = (int)((uint32_t)tile - q * (uint32_t)a.tiles_per_interval) * a.tile_len;
        } else {
            iv = tile / a.tiles_per_interval;
            g.t0 = (int)(tile % a.tiles_per_interval) * a.tile_len;
        }
        g.L = a.interval_len;
        g.out_off = iv * (int64_t)g.L;
        g.tl = min(a.tile_len, g.L - g.t0);
    }
    const int H = a.max_scale;
    g.ta = max(0, g.t0 - H);
    g.nt = min(g.L, g.t0 + g.tl + H) - g.ta;  // bases needing p / z (<= NT)
    g.nc = g.nt + 2 * kPad + 1;                // padded positions
    g.ncs = (g.nc + 6 + 63) & ~63;             // positions staged, in whole tiles (<= NT + 128)
    g.gcp = a.counts_plus + (g.out_off + iv * (int64_t)(2 * kPad + 1) + g.ta);
    g.gcm = a.counts_minus + (g.out_off + iv * (int64_t)(2 * kPad + 1) + g.ta);
    g.gsq = a.seq + (g.out_off + iv * (int64_t)(2 * kPad + 7) + g.ta);
    g.dm = a.dm_ids ? uniform_load(a.dm_ids, iv) : 0;
    return g;
}

// the inputs of one tile as they sit in a lane's registers between the load and phase A
template <int NI>
struct lean_inputs {
    double cp[NI], cm[NI];
    u32 ch[NI];
};

// (NI trips of the workgroup's NT lanes cover the NP + 128 staged positions: two with a base per lane, three
// with two)
template <int NT, int NI>
__device__ __forceinline__ void lean_load(const lean_tile &g, int tid, lean_inputs<NI> &in, bool fake = false) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {  // every load of a lane in flight before the first is used
        const int v = i * NT + tid;
        in.cp[i] = in.cm[i] = 0.0;
        in.ch[i] = 'A';
        if (fake) {  // ablation builds: what the kernel costs without its HBM reads
            in.cp[i] = (double)((v * 7 + g.t0) % 20);
            in.cm[i] = (double)((v * 13 + g.t0) % 20);
            in.ch[i] = "ACGT"[(v * 5 + (v >> 3)) & 3];
            continue;
        }
        if (v < g.nc) {
            in.cp[i] = g.gcp[v];
            in.cm[i] = g.gcm[v];
        }
        if (v < g.nc + 6) in.ch[i] = g.gsq[v];
    }
}

// phase A of one tile: counts -> packed 16-bit integers, sequence -> two bit planes (LDS).
// Returns true if this lane saw an input outside the case the kernel handles.
template <int NT, int NI>
__device__ __forceinline__ bool lean_stage(const lean_inputs<NI> &in, int ncs, int tid, u32 *pk, u32 *bits0, u32 *bits1) {
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    bool bad = false;
    if (tid < 8) {
        pk[tid] = 0;
        pk[8 + ncs + tid] = 0;
    }
    if (tid < 4) {
        bits0[(ncs >> 5) + tid] = 0;
        bits1[(ncs >> 5) + tid] = 0;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (i * NT + wave * kWave >= ncs) break;  // whole tiles: wavefront-uniform
        const int v = i * NT + tid;
        const int ip = (int)in.cp[i], im = (int)in.cm[i];
        bad |= !((double)ip == in.cp[i]) | !((double)im == in.cm[i]) | ((u32)ip > kCountMax) | ((u32)im > kCountMax);
        pk[8 + v] = (u32)ip | ((u32)im << 16);
        const u32 ch = in.ch[i], up = ch & 0xDFu;
        bad |= !((up == 'A') | (up == 'C') | (up == 'G') | (up == 'T'));
        const unsigned long long m0 = __ballot((ch & 2u) != 0), m1 = __ballot((ch & 4u) != 0);
        if (lane == 0) {
            *reinterpret_cast<unsigned long long *>(bits0 + ((i * NT + wave * kWave) >> 5)) = m0;
            *reinterpret_cast<unsigned long long *>(bits1 + ((i * NT + wave * kWave) >> 5)) = m1;
        }
    }
    return bad;
}

// LDS arrays of one workgroup (see lean_lds)
template <int NT>
struct lean_mem {
    double *PP, *PM, *Z, *rowtot, *C, *gt;
    u32 *bits0, *bits1, *pk, *psP, *psM, *xP, *xPs, *xM, *xMs, *edge;
    __device__ __forceinline__ explicit lean_mem(double *smem) {
        typedef lean_lds<NT> LY;
        PP = smem + LY::oPP, PM = smem + LY::oPM, rowtot = smem + LY::oRT, C = rowtot + LY::NROW;
        u32 *words = reinterpret_cast<u32 *>(smem + LY::nDoubles);
        Z = reinterpret_cast<double *>(words + LY::oZB);
        gt = reinterpret_cast<double *>(words + LY::oGT);
        bits0 = words + LY::oB0, bits1 = words + LY::oB1, pk = words + LY::oPK;
        psP = words + LY::oSP, psM = words + LY::oSM;
        xP = words + LY::oXP, xPs = words + LY::oXPs, xM = words + LY::oXM, xMs = words + LY::oXMs;
        edge = words + LY::oEG;
    }
};

// ---- B: 6-mer index and propensities, 2*hw window sums, per-tile scans of the window sums.
// Returns true where an aligned row of 16 equal non-zero window sums shows up (see the header).
template <int NT, int NP, int NI>
__device__ __forceinline__ bool lean_phase_b(const lean_mem<NP> &m, const double2 *table2, int ncs, int nt, int tid,
                                             bool = false) {
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    bool bad = false;
    // the table gathers of both of a lane's positions go out first: one trip to L2 instead of two
    // in a row (a short workgroup lives ~7 us, most of it such trips)
    double2 tt[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (i * NT + wave * kWave >= ncs) break;
        const int v = i * NT + tid;
        // The propensities are read by phase C at the positions [pad + 1 - hw, pad + nt + hw] only -- the 2 hw window
        // of every base that gets a value -- while the positions beyond them, the 50 either side that only the
        // smoothing window reaches, are needed as COUNTS: their 16-byte gathers (the dearest instruction of the
        // vector-memory path: 40 cycles of the CU each, tools/micro/vmem_issue.hip) are left out
        tt[i] = make_double2(0.0, 0.0);
        if (v >= kPad + 1 - kHW && v <= kPad + nt + kHW) {
            const int w32 = v >> 5, sh = v & 31;
            const u32 f0 = __builtin_amdgcn_alignbit(m.bits0[w32 + 1], m.bits0[w32], sh) & 63u;
            const u32 f1 = __builtin_amdgcn_alignbit(m.bits1[w32 + 1], m.bits1[w32], sh) & 63u;
            tt[i] = table2[f0 | (f1 << 6)];  // (P+[v], P-[v-1]); consumed at the end of the iteration
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (i * NT + wave * kWave >= ncs) break;
        const int v = i * NT + tid;
        const u32 *pw = m.pk + 8 + v - kHW;
        u32 W = pw[0];
#pragma unroll
        for (int j = 1; j < 2 * kHW; ++j) W += pw[j];
        const int vr = v + (kWave - 1) - 2 * lane;  // the tile mirrored: suffix scans are prefix scans of it
        const u32 Wr = (u32)__builtin_amdgcn_ds_bpermute((kWave - 1 - lane) << 2, (int)W);  // W of the mirrored lane
        const u32 wp = W & 0xffffu, wm = W >> 16;
        m.psP[v] = (u32)wave_scan_i32((int)wp);
        m.psM[v] = (u32)wave_scan_i32((int)wm);
        // (both strands' extrema as packed 16-bit pairs -- a DPP move and v_pk_max_u16 per step for two scans, a byte
        // permutation per array afterwards -- was built and measured: +1 % on configs 2, 3 and 4, profiles/r06_lean_table.txt)
        const u32 rp = Wr & 0xffffu, rm = Wr >> 16;
        m.xP[v] = wave_scan_umax(0xffffu - wp) | (wave_scan_umax(wp) << 16);
        m.xM[v] = wave_scan_umax(0xffffu - wm) | (wave_scan_umax(wm) << 16);
        m.xPs[vr] = wave_scan_umax(0xffffu - rp) | (wave_scan_umax(rp) << 16);
        m.xMs[vr] = wave_scan_umax(0xffffu - rm) | (wave_scan_umax(rm) << 16);
        // 33 equal non-zero window sums in a row?  (The only way to the one case where smoothing.h:61-69
        // is not S - min - max: see the header.)  Bit l of eP / eM: lanes l and l + 1 hold the same non-zero
        // sum (lane 63 is compared with 0: never set).  Inside the tile that takes 32 set bits in a row --
        // one population count settles it for all but a few tiles, and scalar instructions are as dear as
        // fp64 ones here (tools/micro/issue.hip: 4.2 cycles of the SIMD each, not overlapped with its
        // vector work).  Across two tiles: the two masks go to LDS with the first and the last value,
        // and phase C looks at them where the values on the two sides of a boundary are equal.
        const u32 d = W ^ (u32)__builtin_amdgcn_update_dpp(0, (int)W, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
        const unsigned long long eP = __ballot((d & 0xffffu) < min(wp, 1u));  // equal and non-zero
        const unsigned long long eM = __ballot((d >> 16) < min(wm, 1u));
        if (__builtin_popcountll(eP | eM) >= 32) {
            unsigned long long rP = eP, rM = eM;
            rP &= rP >> 1; rP &= rP >> 2; rP &= rP >> 4; rP &= rP >> 8; rP &= rP >> 16;
            rM &= rM >> 1; rM &= rM >> 2; rM &= rM >> 4; rM &= rM >> 8; rM &= rM >> 16;
            bad |= (rP | rM) != 0;
        }
        {
            const u32 wlast = (u32)__builtin_amdgcn_readlane((int)W, kWave - 1);
            if (lane == 0) {
                u32 *eg = m.edge + 8 * ((i * NT + wave * kWave) >> 6);
                *reinterpret_cast<uint2 *>(eg) = make_uint2(W, wlast);
                *reinterpret_cast<uint4 *>(eg + 4) = make_uint4((u32)eP, (u32)(eP >> 32), (u32)eM, (u32)(eM >> 32));
            }
        }
        if (v >= kPad + 1 - kHW && v <= kPad + nt + kHW) {
            m.PP[v - lean_lds<NP>::kPOff] = tt[i].x;
            m.PM[v - lean_lds<NP>::kPOff] = tt[i].y;
        }
    }
    return bad;
}

// ---- C: trimmed-mean smoothing + expected counts of base t of the tile, both strands ('+' at padded
// position pad+1+t, '-' at pad+t; detect.py:121-122); D: observed count, p-value and z from the
// (exp, obs) table.  Bases beyond nt return z = 0.  A lane has BPL bases: t = tid, tid + NT, ...
template <int NP>
__device__ __forceinline__ bool lean_cd_base(const lean_mem<NP> &m, const lean_args &a, kcoef *kc, const double2 *memo,
                                             int dm, int nt, int t, lean_tracks &tr, double &z, bool &miss, u32 &miss_e) {
    bool bad = false;
    z = 0.0;
    tr.ex = tr.pv = 0.0;
    tr.k = 0;
    if (t < nt) {
        double e2[2];
#pragma unroll
        for (int strand = 0; strand < 2; ++strand) {
            const int v = kPad + t + (strand ? 0 : 1);
            const int lo = v - kSHW, hi = v + kSHW;
            const u32 S = window_sum(strand ? m.psM : m.psP, lo, hi);
            const u32 *xp = strand ? m.xM : m.xP, *xs = strand ? m.xMs : m.xPs;
            const int mid = (lo | 63) + 64;  // last position of the tile after lo's
            const u32 x1 = xs[lo], x2 = xp[hi], x3 = xp[(hi >> 6) == (lo >> 6) + 2 ? mid : hi];
            const u32 cmin = max(max(x1 & 0xffffu, x2 & 0xffffu), x3 & 0xffffu);  // 0xffff - min
            const u32 mx = max(max(x1 >> 16, x2 >> 16), x3 >> 16);
            const double tsum = (double)((S + cmin) - mx - 0xffffu);  // S - min - max
            const double *P = (strand ? m.PM : m.PP) + (kPad + t + 1 - kHW - lean_lds<NP>::kPOff);  // P[v-hw .. v+hw-1]
            double q = P[0];
#pragma unroll
            for (int j = 1; j < 2 * kHW; ++j) q += P[j];  // left to right, like predict.h:43-47
            const double q99 = mul_vs(q, kc->c99);
            double r = __builtin_amdgcn_rcp(q99);  // 2^-24; one Newton step: 2e-15, far inside the band
            r = fma(fma(-q99, r, 1.0), r, r);
            const double x = (P[kHW] * tsum) * r;  // ~ P/Q * t/99
            const double e = floor(x + 0.5);
            // too close to a tie (or not a number)?  |x - e| = 1/2 - (distance of x to the nearest half-integer)
            bad |= !(fabs(x - e) < fma(x, -kc->band, 0.5));
            e2[strand] = e;
        }
        tr.ex = e2[0] + e2[1];
        tr.k = (m.pk[8 + kPad + 1 + t] & 0xffffu) + (m.pk[8 + kPad + t] >> 16);
        const u32 ei = (u32)(int)tr.ex;
        bool hit = ei < (u32)a.memo_exp && tr.k < (u32)a.memo_obs;
        double2 pz = memo[hit ? ei * (u32)a.memo_obs + tr.k : 0u];
        if (!hit && a.memo2) {  // (rare: hotspots) the kept second-level table, as far as it is filled
            const int h0 = a.memo2_have[0], h1 = a.memo2_have[1];
            if (h0 >= 0 && h1 >= 0 && ei <= (u32)h0 && tr.k <= (u32)h1) {
                pz = a.memo2[((size_t)dm * a.miss_rows + ei) * a.miss_stride + tr.k];
                hit = true;
            }
        }
        tr.pv = pz.x;
        z = pz.y;
        bad |= !hit | ((__double2hiint(z) & 0x7ff00000) == 0x7ff00000);  // a miss, or a non-finite z
        // a pair outside the table but inside the second-level bounds
        if (!hit && a.miss_max && ei < (u32)a.miss_rows && tr.k < (u32)a.miss_stride) {
            miss_e = miss ? max(miss_e, ei) : ei;
            miss = true;
        }
    }
    return bad;
}

template <int NT, int NP, int BPL>
__device__ __forceinline__ bool lean_phase_cd(const lean_mem<NP> &m, const lean_args &a, kcoef *kc, const double2 *memo,
                                              int dm, int nt, int ncs, int tid, lean_tracks (&tr)[BPL], double (&z)[BPL]) {
    bool bad = false;
    bool miss = false;  // a lane with a base whose (exp, obs) pair lies outside the table but inside the second-level bounds
    u32 miss_e = 0, miss_k = 0;
    if (tid + 1 < (ncs >> 6)) {  // a run of equal non-zero window sums across the boundary of tiles tid, tid + 1
        const u32 *e0 = m.edge + 8 * tid;
        const u32 last = e0[1], first = e0[8];
        const bool sameP = ((last ^ first) & 0xffffu) == 0 && (last & 0xffffu) != 0;
        const bool sameM = ((last ^ first) >> 16) == 0 && (last >> 16) != 0;
        if (sameP | sameM) {  // (rare) the run that ends tile tid + the run that starts tile tid + 1
            const uint4 m0 = *reinterpret_cast<const uint4 *>(e0 + 4), m1 = *reinterpret_cast<const uint4 *>(e0 + 12);
            const unsigned long long eP0 = m0.x | ((unsigned long long)m0.y << 32), eM0 = m0.z | ((unsigned long long)m0.w << 32);
            const unsigned long long eP1 = m1.x | ((unsigned long long)m1.y << 32), eM1 = m1.z | ((unsigned long long)m1.w << 32);
            const u32 trailP = (u32)__builtin_clzll(~(eP0 << 1)) + 1u, leadP = (u32)__builtin_ctzll(~eP1) + 1u;
            const u32 trailM = (u32)__builtin_clzll(~(eM0 << 1)) + 1u, leadM = (u32)__builtin_ctzll(~eM1) + 1u;
            bad |= (sameP && trailP + leadP >= 33u) | (sameM && trailM + leadM >= 33u);
        }
    }
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
        bool mb = false;
        u32 me_b = 0;
        bad |= lean_cd_base<NP>(m, a, kc, memo, dm, nt, b * NT + tid, tr[b], z[b], mb, me_b);
        if (mb) {
            miss_e = miss ? max(miss_e, me_b) : me_b;
            miss_k = miss ? max(miss_k, tr[b].k) : tr[b].k;
            miss = true;
        }
    }
    // the largest missed pair sizes the second-level table of the redo pass: one pair of atomics per
    // wavefront that has a miss (a hotspot tile has hundreds of missing lanes; one atomic per lane
    // on the same two words cost 1.4 ms of the 24.9 ms heavy-tailed launch)
    if (__builtin_amdgcn_ballot_w64(miss)) {
        int me = miss ? (int)miss_e : -1, mk = miss ? (int)miss_k : -1;
#pragma unroll
        for (int d = 32; d; d >>= 1) {
            me = max(me, __shfl_xor(me, d));
            mk = max(mk, __shfl_xor(mk, d));
        }
        if ((tid & 63) == 0) {
            atomicMax(&a.miss_max[0], me);
            atomicMax(&a.miss_max[1], mk);
        }
    }
    return bad;
}

__device__ __forceinline__ lean_owner lean_own(const lean_tile &g, int t, bool no_stores) {  // t: the base's index in the tile
    lean_owner o;
    o.t = g.ta + t;
    o.L = g.L;
    o.out_off = g.out_off;
    o.mine = t < g.nt && o.t >= g.t0 && o.t < g.t0 + g.tl && !no_stores;
    return o;
}

// One tile per workgroup.  With its loads and stores ablated the kernel needs 3.52e8 shader cycles
// per 10^9 bases (GRBM_GUI_ACTIVE, at 2.38 GHz: the instruction-issue bound); with them 4.63e8 at
// 2.29 GHz.  Per-workgroup timelines (tools/lean_trace.py, config 3) show 12.1 us of life, 2.6 us
// of it before the inputs are staged, and 2.8 us between the last store and the successor's first
// instruction (a slot is not released until every store is acknowledged) -- but that is not where
// the extra cycles are: six forms that hide those two latencies were built and measured.  Four
// kinds of workgroups that stay and walk the tiles (the last with every vector-memory wait --
// the counter is in order across loads AND stores on gfx9 -- at least a phase younger than what
// it covers: 26.0 / 25.2 / 25.5 / 25.8 / 27.0 / 27.8 ms at 1 / 4 / 30 / 244 / 977 / 1,953 tiles
// per workgroup against 25.0; start offsets by hardware slot changed nothing), and two tiles per
// workgroup in straight-line code (both tiles' loads first, every store of both after the last
// gather, a second z array: 78.5 KB of LDS; 24.5-24.8 against 24.0-24.4 ms, arithmetic alone
// 18.6 against 18.9): all correct on the whole GPU suite, none faster.  The memory time that is
// not hidden is spread over the phases -- gathers and store issue take longer while 3.4 TB/s
// stream through the same L2 -- not concentrated at a workgroup's two ends.
// BPL: bases per lane.  1 in every size class but the largest, whose 1,024-base tiles are taken by 512 lanes with
// two bases each (the same tile, the same 52.7 KB of LDS): three workgroups of eight wavefronts share a CU where
// two of sixteen did -- 1.6 % on config 3 (24.38 -> 23.98 ms, alternating on one lease).
// FPT_LEAN_BPL2=0 brings the 1,024-lane instance back.
template <int NT, int BPL>
__global__ void __launch_bounds__(NT, BPL == 1 ? 8 : 6) k_scan_lean(const lean_args a) {
    constexpr int NP = NT * BPL;                     // bases (and output positions) of a tile
    constexpr int NI = (NP + 128 + NT - 1) / NT;     // trips of the lanes over the staged positions
    extern __shared__ double smem[];
    const lean_mem<NP> m(smem);
    constexpr int kEdge = NP + 32 + 15;

    typedef const __attribute__((address_space(4))) lean_args kargs;
    kcoef *kc = &((kargs *)__builtin_amdgcn_kernarg_segment_ptr())->c;
    const int tid = threadIdx.x;
    // The 1,024-base tiles in an XCD-contiguous order (round 6): workgroups b, b + 8, b + 16, ... -- one XCD, one L2 --
    // take NEIGHBOURING tiles, so the 128-byte line two neighbours' padded rows share (1 of the 70 a row spans) is
    // fetched once: config 3 20.79 -> 20.54 ms (three alternating pairs, profiles/r06_lean_prio.txt).  Not for the
    // smaller classes: config 2's 500-base tiles lose 7 % with it, the ragged shape 1-2 % (round 5).
    int64_t tile = a.tile_first + blockIdx.x;
    if (NP == 1024) {
        const unsigned n = gridDim.x, q = n >> 3, r = n & 7u, x = blockIdx.x & 7u, idx = blockIdx.x >> 3;
        tile = a.tile_first + (int64_t)(x * q + min(x, r) + idx);  // XCD x owns q (+ 1 for the first r) consecutive tiles
    }
    // Wave priority (round 6): the SIMD's arbiter issues from its OLDEST wavefront first, and the oldest are deep in
    // the arithmetic of phases B - E -- a workgroup that has just arrived waits behind them for the few instructions
    // that send its loads off, and the 2.6 us of their latency start late.  Raised until the inputs are staged
    // (s_setprio 3 here, 0 before the first barrier), the loads of a new tile go out at once: config 3 23.5 -> 21.3 ms,
    // config 2 0.85 -> 0.78, config 4 1.63 -> 1.52, bit-identical (profiles/r06_lean_prio.txt: the level -- 1, 2, 3 --
    // does not matter, raising it in any later phase too gives part of it back).  FPT_LEAN_PRIO=0 switches it off.
    const bool prio = a.prio != 0;
    if (prio) __builtin_amdgcn_s_setprio(3);
    LEAN_TRACE(1);
    const lean_tile g = lean_geometry(a, tile);
    const double2 *memo = a.memo + (size_t)g.dm * a.memo_exp * a.memo_obs;

    // ---- A: counts -> packed 16-bit integers, sequence -> two bit planes
    lean_inputs<NI> in;
    lean_load<NT, NI>(g, tid, in, LEAN_STOP(5) || LEAN_STOP(6));
    // the table of g for phase E: loaded behind the inputs, carried in registers (the kernel uses 36 of the 80 its
    // residency allows) until the scan arrays it will live in are free
    constexpr int KG = lean_lds<NP>::kGtPerLane;
    const bool use_tab = lean_lds<NP>::kTab && a.tab != 0;  // (FPT_LEAN_TAB=0: the Horner chain, for A/B runs)
    // (several scales only: with ONE narrow scale the table's staging -- 512 doubles per tile -- costs more than nine
    // fp64 instructions per base save: config 2 0.74 -> 0.76 ms, config 4 1.45 -> 1.55, profiles/r06_lean_table.txt)
    const bool wide = use_tab && a.n_scales > 0 && !(a.n_scales == 1 && a.max_scale <= 8);
    double gtr[KG];
#pragma unroll
    for (int i = 0; i < KG; ++i) {
        gtr[i] = 0.0;
        if (wide && i * NT + tid < 4 * FPT_NDTR_GTAB_N) gtr[i] = g_lean_gtab[i * NT + tid];
    }
    bool bad = lean_stage<NT, NI>(in, g.ncs, tid, m.pk, m.bits0, m.bits1);  // outside the case this kernel handles?
    if (prio) __builtin_amdgcn_s_setprio(0);
    LEAN_TRACE(2);
    __syncthreads();
    LEAN_TRACE(3);
    if (LEAN_STOP(1)) return;

    bad |= lean_phase_b<NT, NP, NI>(m, a.table2, g.ncs, g.nt, tid);
    __syncthreads();
    LEAN_TRACE(4);
    if (LEAN_STOP(2)) return;

    lean_tracks tr[BPL];
    double z[BPL];
    bad |= lean_phase_cd<NT, NP, BPL>(m, a, kc, memo, g.dm, g.nt, g.ncs, tid, tr, z);
    lean_owner o[BPL];
    // (the tracks are stored HERE, ahead of phase E's barriers: a store issued later holds the workgroup's slot
    // longer -- moved into phase E they cost 15-19 % on config 3)
#pragma unroll
    for (int b = 0; b < BPL; ++b) {
        o[b] = lean_own(g, b * NT + tid, LEAN_STOP(6));
        lean_store_tracks(a, o[b], tr[b]);
    }

    LEAN_TRACE(5);
    // ---- E: Stouffer windows (windowing.h:53-84)
    if (a.n_scales == 0 || LEAN_STOP(3)) {
    } else if (a.n_scales == 1 && a.max_scale <= 8) {
        __syncthreads();  // Z takes the place of the scan arrays: every lane is through phase C
#pragma unroll
        for (int b = 0; b < BPL; ++b) m.Z[16 + b * NT + tid] = z[b];
        __syncthreads();
#pragma unroll
        for (int b = 0; b < BPL; ++b) bad |= lean_window_narrow<NP>(a, kc, o[b], b * NT + tid, m.Z);
    } else {
        // one workgroup-wide prefix sum of z in two levels: rows of 16 lanes on the DPP path, the
        // NP/16 row totals scanned by the first wavefront, the carries added once behind a third
        // barrier (measured 24.7 -> 24.5 ms against adding them in every scale)
        double zr[BPL];
#pragma unroll
        for (int b = 0; b < BPL; ++b) zr[b] = lean_z_rows<NP>(z[b], b * NT + tid, m.rowtot);
        __syncthreads();
        lean_z_carries<NP>(tid, m.rowtot, m.C);
        if (use_tab) {  // every lane is through phase C: the scan arrays are free
#pragma unroll
            for (int i = 0; i < KG; ++i) {
                const int e = i * NT + tid;  // (c3, c2, c1, c0) of slot e / 4: the halves go to two arrays (lean_windows)
                if (e < 4 * FPT_NDTR_GTAB_N)
                    m.gt[((e & 2) ? 2 * FPT_NDTR_GTAB_N : 0) + 2 * (e >> 2) + (e & 1)] = gtr[i] * kc->inv_g0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < BPL; ++b) lean_z_finish<NP>(zr[b], b * NT + tid, m.C, m.Z);  // two barriers behind phase C: the scan arrays are free
        if (tid == 0) {
            m.Z[15] = 0.0;
            m.Z[kEdge] = -1e4;
        }
        __syncthreads();
        if (use_tab) {
#pragma unroll
            for (int b = 0; b < BPL; ++b) bad |= lean_windows<NP, lean_args, true>(a, kc, o[b], b * NT + tid, m.Z, m.gt);
        } else {
#pragma unroll
            for (int b = 0; b < BPL; ++b) bad |= lean_windows<NP, lean_args, false>(a, kc, o[b], b * NT + tid, m.Z, m.gt);
        }
    }
    if (bad) a.redo[tile] = 1;
    LEAN_TRACE(6);
}

typedef void (*lean_kernel_t)(const lean_args);
// The tiles of the 1,024-base class are taken by 512 lanes with two bases each (FPT_LEAN_BPL2=0, read once:
// 1,024 lanes with one)
bool lean_bpl2() {
    static const bool on = !getenv("FPT_LEAN_BPL2") || atoi(getenv("FPT_LEAN_BPL2")) != 0;
    return on;
}
lean_kernel_t lean_kernel(int nt) {
    switch (nt) {
        case 128: return k_scan_lean<128, 1>;
        case 192: return k_scan_lean<192, 1>;
        case 256: return k_scan_lean<256, 1>;
        case 384: return k_scan_lean<384, 1>;
        case 512: return k_scan_lean<512, 1>;
        case 768: return k_scan_lean<768, 1>;
        default: return lean_bpl2() ? k_scan_lean<512, 2> : k_scan_lean<1024, 1>;
    }
}


}  // namespace

namespace fptk {

size_t scan_lean_lds_bytes(int nt) {
    switch (nt) {
        case 128: return lean_lds<128>::bytes;
        case 192: return lean_lds<192>::bytes;
        case 256: return lean_lds<256>::bytes;
        case 384: return lean_lds<384>::bytes;
        case 512: return lean_lds<512>::bytes;
        case 768: return lean_lds<768>::bytes;
        default: return lean_lds<1024>::bytes;
    }
}

hipError_t scan_lean_set_lds(int nt) {
    return hipFuncSetAttribute((const void *)lean_kernel(nt), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)scan_lean_lds_bytes(nt));
}

bool scan_lean_applies_hw(int hw, int shw, int k_trim) { return hw == kHW && shw == kSHW && k_trim == 1; }

bool scan_lean_applies(const scan_launch &sl) {
    if (sl.hw != kHW || sl.shw != kSHW || sl.k_trim != 1 || !sl.memo || !sl.redo || sl.counts_only || !sl.table2)
        return false;
    if (sl.memo_exp < 1 || sl.memo_obs < 1 || (int64_t)sl.memo_exp * sl.memo_obs > 0x7fffffff) return false;
    for (int i = 0; i < sl.n_scales; ++i)
        if (sl.scales[i] > 200) return false;
    return true;
}

void launch_scan_lean(hipStream_t st, int nt, int grid, const scan_launch &sl) {
    lean_args a;
    fill_lean_args(sl, a);

#ifdef FPT_ABLATE
    static int64_t *d_trace = nullptr;
    const char *trace_path = getenv("FPT_LEAN_TRACE");
    if (trace_path) {
        if (!d_trace) (void)hipMalloc(&d_trace, (size_t)1 << 26);  // 2^20 workgroups x 8 words
        if (grid <= (1 << 20)) a.trace = d_trace;
    }
#endif
    const int lanes = (nt > 768 && lean_bpl2()) ? 512 : nt;
    hipLaunchKernelGGL(lean_kernel(nt), dim3(grid), dim3(lanes), scan_lean_lds_bytes(nt), st, a);
#ifdef FPT_ABLATE
    if (a.trace) {  // the last launch's record: 8 words per workgroup (see LEAN_TRACE)
        std::vector<int64_t> h((size_t)grid * 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), a.trace, h.size() * 8, hipMemcpyDeviceToHost);
        char path[1024];
        snprintf(path, sizeof path, "%s.%d", trace_path, nt);  // one file per workgroup size
        if (FILE *f = fopen(path, "wb")) {
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
#endif
}

// The bias table in the order the lean kernel indexes it: entry F = plane0 | plane1 << 6, where bit
// m of plane0 / plane1 is bit 1 / bit 2 of the ASCII code of base m of the 6-mer (A=00 C=01 T=10
// G=11).  .x = table[6-mer] (bias.py:101-111), .y = table[reverse complement of the 6-mer]
// (predict.pyx:47-61,150-153), both in the reference's A<C<G<T base-4 order.
void build_lean_table(const double *table4096, double *out8192) {
    static const int kStd[4] = {0, 1, 3, 2};  // plane code -> A0 C1 G2 T3
    for (int F = 0; F < 4096; ++F) {
        int fwd = 0, rev = 0;
        for (int m = 0; m < 6; ++m) {
            const int b = kStd[((F >> m) & 1) | (((F >> (6 + m)) & 1) << 1)];
            fwd = fwd * 4 + b;              // base m is digit 5-m
            rev += (3 - b) << (2 * m);      // complement of base m is digit m of the reverse complement
        }
        out8192[2 * F] = table4096[fwd];
        out8192[2 * F + 1] = table4096[rev];
    }
}

}  // namespace fptk
