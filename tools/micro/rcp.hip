// accuracy of v_rcp_f64 and of one / two Newton steps (developer microbenchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    double d = x[i], r = __builtin_amdgcn_rcp(d);
    o[i] = r;
    double r1 = fma(fma(-d, r, 1.0), r, r);
    o[n + i] = r1;
    o[2 * n + i] = fma(fma(-d, r1, 1.0), r1, r1);
}
int main() {
    int n = 1 << 20; double *hx = new double[n], *ho = new double[3 * n];
    for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 40) - 20);
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(ho, dout, 3 * n * 8, hipMemcpyDeviceToHost);
    for (int s = 0; s < 3; ++s) { double m = 0; for (int i = 0; i < n; ++i) { double e = fabs(ho[s * n + i] * hx[i] - 1.0); if (e > m) m = e; } printf("step %d: max |r*d - 1| = %.3e\n", s, m); }
    return 0;
}
