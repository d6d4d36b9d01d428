// Diagnostic: how many exact ndtr evaluations does the threshold search of k_fdr_null take per
// observed value (and per wavefront: the largest of its lanes)?  160 sorted standard-normal y, as an
// interval of the whole-genome shape has them.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
// -mllvm -disable-machine-licm tools/micro/thr_evals.hip -o gpurun_out/thr_evals
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
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
int main() {
    const int L = 160, NI = 2000, NT = 192;
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd(0.0, 1.0);
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
    for (int estimate = 0; estimate < 2; ++estimate) {
        hipLaunchKernelGGL(k_thr, dim3(NI), dim3(NT), 0, 0, dy, dout, dc, NI * NT, estimate);
        std::vector<int> c(y.size());
        std::vector<double> t(y.size());
        hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), dout, t.size() * 8, hipMemcpyDeviceToHost);
        long differ = 0;
        if (estimate) for (size_t i = 0; i < t.size(); ++i) differ += t[i] != first[i];
        first = t;
        double sum = 0, wsum[3] = {0, 0, 0}; int mx = 0;
        for (int k = 0; k < NI; ++k)
            for (int w = 0; w < 3; ++w) {
                int m = 0;
                for (int l = 0; l < 64; ++l) { const int i = w * 64 + l; if (i < L) { m = std::max(m, c[(size_t)k * NT + i]); sum += c[(size_t)k * NT + i]; } }
                wsum[w] += m; mx = std::max(mx, m);
            }
        printf("%s: evaluations per value: mean %.2f; largest of a wavefront: wave0 %.2f wave1 %.2f wave2 %.2f (mean over %d intervals), max %d; thresholds that differ from the plain search: %ld\n",
               estimate ? "gallop from the width estimate" : "gallop from 1", sum / (NI * L), wsum[0] / NI, wsum[1] / NI, wsum[2] / NI, NI, mx, differ);
    }
    return 0;
}
