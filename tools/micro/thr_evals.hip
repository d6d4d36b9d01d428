// Diagnostic: how many exact ndtr evaluations does the threshold search of k_fdr_null take per
// observed value (and per wavefront: the largest of its lanes)?  160 sorted standard-normal y, as an
// interval of the whole-genome shape has them.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -mllvm -disable-machine-licm tools/micro/thr_evals.hip -o gpurun_out/thr_evals
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../footprint_tools_amd/csrc/fpt_math.hpp"
__device__ __forceinline__ long long ordered_bits(double y) {
    const long long b = __double_as_longlong(y);
    return b < 0 ? (long long)(0x8000000000000000ull - (unsigned long long)b) : b;
}
__device__ __forceinline__ double from_ordered_bits(long long k) {
    const long long b = k < 0 ? (long long)(0x8000000000000000ull - (unsigned long long)k) : k;
    return __longlong_as_double(b);
}
__global__ void k_thr(const double *y, double *out, int *cnt, int n_total, int estimate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const double yo = y[i];
    const double P = fptm::ndtr(yo);
    if (!(P < 1.0)) {  // (ndtr never exceeds 1: no threshold to search)
        out[i] = 1e300;
        cnt[i] = 0;
        return;
    }
    long long lo = ordered_bits(yo), hi, step = 1;
    int n = 0;
    if (estimate) {  // the gallop starts at a quarter of the plateau width ulp(P) / (phi(y) ulp(y))
        const double up = __longlong_as_double(__double_as_longlong(P) + 1) - P;
        const double ay = fabs(yo);
        const double uy = ay > 0.0 ? __longlong_as_double(__double_as_longlong(ay) + 1) - ay : 4.9406564584124654e-324;
        const double w = up * 2.5066282746310002 * exp(0.5 * yo * yo) / uy;
        if (w >= 8.0) step = 1ll << min(ilogb(w) - 2, 60);
    }
    for (;;) {
        const long long c = lo + step;
        ++n;
        if (fptm::ndtr(from_ordered_bits(c)) > P) { hi = c; break; }
        lo = c; step <<= 1;
    }
    while (hi - lo > 1) {
        const long long mid = lo + ((hi - lo) >> 1);
        ++n;
        if (fptm::ndtr(from_ordered_bits(mid)) > P) hi = mid; else lo = mid;
    }
    out[i] = from_ordered_bits(hi);
    cnt[i] = n;
}
// the search on the addend before ndtr's last rounding (see ndtr_threshold_open in fpt_kernels.hip)
__device__ __forceinline__ double next_d(double x, long long k) { return from_ordered_bits(ordered_bits(x) + k); }
__global__ void k_thr_open(const double *y, double *out, int *cnt, int n_total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const double yo = y[i];
    const double P = fptm::ndtr(yo);
    int n = 0;
    double T;
    if (!(P < 1.0)) {
        out[i] = 1e300;
        cnt[i] = 0;
        return;
    }
    bool open = yo > 0.0;
    if (open) {
        double base;
        const double t0 = fptm::ndtr_addend_pos(yo, base);
        ++n;
        const double up = next_d(P, 1) - P;
        double ts = (P - base) + 0.5 * up;  // about the smallest addend whose sum rounds above P
        while (base + next_d(ts, -1) > P) ts = next_d(ts, -1);
        while (!(base + ts > P)) ts = next_d(ts, 1);
        const double phi = exp(-0.5 * yo * yo) * 0.3989422804014327;
        double a1 = yo + (ts - t0) / phi;
        if (!(a1 > yo)) a1 = next_d(yo, 1);
        // smallest a >= next(yo) with addend(a) >= ts, base unchanged: gallop from a1, then bisect
        long long k = ordered_bits(a1), klo = ordered_bits(yo);  // addend(klo) < ts
        double b2;
        double tk = fptm::ndtr_addend_pos(from_ordered_bits(k), b2);
        ++n;
        bool ok = b2 == base;
        long long lo, hi;
        if (ok && tk >= ts) {  // down: find lo with addend(lo) < ts
            hi = k;
            long long step = 1;
            for (;;) {
                long long c = hi - step;
                if (c <= klo) { lo = klo; break; }
                const double tc = fptm::ndtr_addend_pos(from_ordered_bits(c), b2);
                ++n;
                if (b2 != base) { ok = false; break; }
                if (!(tc >= ts)) { lo = c; break; }
                hi = c;
                step <<= 1;
            }
        } else if (ok) {  // up
            lo = k;
            long long step = 1;
            for (;;) {
                long long c = lo + step;
                const double tc = fptm::ndtr_addend_pos(from_ordered_bits(c), b2);
                ++n;
                if (b2 != base) { ok = false; break; }
                if (tc >= ts) { hi = c; break; }
                lo = c;
                step <<= 1;
            }
        }
        if (ok) {
            while (hi - lo > 1) {
                const long long mid = lo + ((hi - lo) >> 1);
                const double tm = fptm::ndtr_addend_pos(from_ordered_bits(mid), b2);
                ++n;
                if (tm >= ts) hi = mid; else lo = mid;
            }
            T = from_ordered_bits(hi);
        }
        open = ok;
    }
    if (!open) {  // the plain search on ndtr itself
        long long lo = ordered_bits(yo), hi, step = 1;
        for (;;) {
            const long long c = lo + step;
            ++n;
            if (fptm::ndtr(from_ordered_bits(c)) > P) { hi = c; break; }
            lo = c; step <<= 1;
        }
        while (hi - lo > 1) {
            const long long mid = lo + ((hi - lo) >> 1);
            ++n;
            if (fptm::ndtr(from_ordered_bits(mid)) > P) hi = mid; else lo = mid;
        }
        T = from_ordered_bits(hi);
    }
    out[i] = T;
    cnt[i] = n;
}

// the kernel's ndtr_threshold_from (fpt_kernels.hip), with a counter
__device__ __forceinline__ double thr_from(double y, double P, int &n) {
    long long lo = ordered_bits(y), hi, step = 1;  // ndtr(lo) <= P
    long long k = lo + 1;
    if (y > 0.0) {
        // Newton on the addend before ndtr's last rounding, to where base + t first rounds above P
        const bool central = fptm::ndtr_is_central(y);
        const double up = __longlong_as_double(__double_as_longlong(P) + 1) - P;
        double a = y;
        for (int it = 0; it < 6; ++it) {
            if (fptm::ndtr_is_central(a) != central || !(a < 40.0)) break;
            double base, ec = 1.0;
            const double t = fptm::ndtr_addend_pos(a, base, &ec);
            ++n;
            const double ts = (P - base) + 0.5 * up;
            // tail: ln q is nearly linear in a (slope -hazard = -sqrt(2/pi) / erfce); central: t itself is
            const double d = central ? (ts - t) / (exp(-0.5 * a * a) * 0.3989422804014327)
                                     : log(t / ts) * ec * 1.2533141373155003;
            const double an = a + d;
            const double ua = __longlong_as_double(__double_as_longlong(a) + 1) - a;
            if (!(an > y)) break;
            a = an;
            if (fabs(d) <= 4.0 * ua) break;
        }
        if (a > y && a < 40.0) k = ordered_bits(a);
    }
    ++n;
    if (k > lo + 1 && fptm::ndtr(from_ordered_bits(k)) > P) {  // at or beyond the end: down to it
        hi = k;
        for (;;) {
            const long long c = hi - step;
            if (c <= lo) break;
            ++n;
            if (!(fptm::ndtr(from_ordered_bits(c)) > P)) { lo = c; break; }
            hi = c;
            step <<= 1;
        }
    } else {
        if (k > lo + 1) lo = k; else --n;
        for (;;) {
            const long long c = lo + step;
            ++n;
            if (fptm::ndtr(from_ordered_bits(c)) > P) { hi = c; break; }
            lo = c;
            step <<= 1;
        }
    }
    while (hi - lo > 1) {
        const long long mid = lo + ((hi - lo) >> 1);
        ++n;
        if (fptm::ndtr(from_ordered_bits(mid)) > P) hi = mid; else lo = mid;
    }
    return from_ordered_bits(hi);
}
__global__ void k_thr_newton(const double *yv, double *out, int *cnt, int n_total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const double y = yv[i];
    const double P = fptm::ndtr(y);
    int n = 0;
    out[i] = P < 1.0 ? thr_from(y, P, n) : 1e300;
    cnt[i] = n;
}

int main() {
    const int L = 160, NI = 2000, NT = 192;
    std::mt19937_64 g(1);
    const char *shift_env = getenv("THR_SHIFT");
    const double shift = shift_env ? atof(shift_env) : 0.0;  // THR_SHIFT=1.5: windows skewed towards p = 1, as the bench's are
    std::normal_distribution<double> nd(shift, 1.0);
    std::vector<double> y((size_t)NI * NT, 0.0);
    for (int k = 0; k < NI; ++k) {
        std::vector<double> v(L);
        for (double &x : v) x = nd(g);
        std::sort(v.begin(), v.end());
        for (int i = 0; i < NT; ++i) y[(size_t)k * NT + i] = v[i < L ? i : L - 1];
    }
    double *dy, *dout; int *dc;
    hipMalloc(&dy, y.size() * 8); hipMalloc(&dout, y.size() * 8); hipMalloc(&dc, y.size() * 4);
    hipMemcpy(dy, y.data(), y.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> first;
    for (int estimate = 0; estimate < 4; ++estimate) {
        if (estimate < 2) hipLaunchKernelGGL(k_thr, dim3(NI), dim3(NT), 0, 0, dy, dout, dc, NI * NT, estimate);
        else if (estimate == 2) hipLaunchKernelGGL(k_thr_open, dim3(NI), dim3(NT), 0, 0, dy, dout, dc, NI * NT);
        else hipLaunchKernelGGL(k_thr_newton, dim3(NI), dim3(NT), 0, 0, dy, dout, dc, NI * NT);
        std::vector<int> c(y.size());
        std::vector<double> t(y.size());
        hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), dout, t.size() * 8, hipMemcpyDeviceToHost);
        long differ = 0;
        long hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // difference in ulps: <= -4, -3 .. 3, >= 4
        if (estimate) for (size_t i = 0; i < t.size(); ++i) {
            differ += t[i] != first[i];
            long long da, db;
            memcpy(&da, &t[i], 8); memcpy(&db, &first[i], 8);
            long long dd = da - db;
            if (dd) hist[dd <= -4 ? 0 : dd >= 4 ? 8 : dd + 4] += 1;
        }
        if (estimate >= 2) printf("   differences in ulps (<=-4, -3..3, >=4): %ld %ld %ld %ld | %ld %ld %ld %ld\n", hist[0], hist[1], hist[2], hist[3], hist[5], hist[6], hist[7], hist[8]);
        if (!estimate) first = t;
        double sum = 0, wsum[3] = {0, 0, 0}; int mx = 0;
        for (int k = 0; k < NI; ++k)
            for (int w = 0; w < 3; ++w) {
                int m = 0;
                for (int l = 0; l < 64; ++l) { const int i = w * 64 + l; if (i < L) { m = std::max(m, c[(size_t)k * NT + i]); sum += c[(size_t)k * NT + i]; } }
                wsum[w] += m; mx = std::max(mx, m);
            }
        printf("%s: evaluations per value: mean %.2f; largest of a wavefront: wave0 %.2f wave1 %.2f wave2 %.2f (mean over %d intervals), max %d; thresholds that differ from the plain search: %ld\n",
               estimate == 3 ? "Newton start, search on ndtr" : estimate == 2 ? "search on the addend before the last rounding" : estimate ? "gallop from the width estimate" : "gallop from 1", sum / (NI * L), wsum[0] / NI, wsum[1] / NI, wsum[2] / NI, NI, mx, differ);
    }
    return 0;
}
