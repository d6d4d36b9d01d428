// Cost of issuing vector-memory instructions on gfx950 (developer microbenchmark, not part of the library):
// every wavefront stores to / loads from its own small region again and again (cache hits), W wavefronts
// per SIMD.  Prints cycles per wave-instruction and CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/vmem_issue.hip -o /tmp/vmem && /tmp/vmem
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, double *buf) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    double *p = buf + wave * 1024 + (threadIdx.x & 63) * (MODE == 1 || MODE == 3 ? 2 : 1);
    double acc = 0.0;
    double2 acc2 = make_double2(0.0, 0.0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double *q = p + (j & 3) * 128;
            if (MODE == 0) { __builtin_nontemporal_store((double)i, q); asm volatile("" ::: "memory"); }
            if (MODE == 1) { *reinterpret_cast<double2 *>(q) = make_double2((double)i, (double)j); asm volatile("" ::: "memory"); }
            if (MODE == 2) { double v = *(volatile double *)q; acc += v; }
            if (MODE == 3) { const volatile double *vq = q; acc2.x += vq[0]; acc2.y += vq[1]; }
            if (MODE == 4) { *q = (double)i; asm volatile("" ::: "memory"); }
        }
    }
    if (acc + acc2.x + acc2.y == 12345.678) buf[0] = acc;
}
template <int MODE> void run(const char *name, double *buf, int n_cu, double ghz, int bytes_per_lane) {
    printf("%-40s", name);
    for (int w : {1, 2, 4, 8}) {
        const int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(n_cu * w), dim3(256), 0, 0, 10, buf);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(n_cu * w), dim3(256), 0, 0, iters, buf);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * ghz * 1e9;
        const double n_inst_per_cu = (double)iters * 8 * 4 * w;  // wave-instructions per CU
        printf("  w=%d: %6.1f cyc/inst/CU %6.0f GB/s", w, cyc / n_inst_per_cu, n_inst_per_cu * n_cu * 64.0 * bytes_per_lane / (ms * 1e-3) / 1e9);
    }
    printf("\n");
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    double *buf; hipMalloc(&buf, (size_t)n_cu * 8 * 4 * 1024 * 8 + 4096); hipMemset(buf, 0, (size_t)n_cu * 8 * 4 * 1024 * 8);
    printf("%s, %d CUs, %.2f GHz nominal\n", p.gcnArchName, n_cu, ghz);
    run<4>("global_store_dwordx2 (8 B/lane)", buf, n_cu, ghz, 8);
    run<0>("nontemporal store 8 B/lane", buf, n_cu, ghz, 8);
    run<1>("global_store_dwordx4 (16 B/lane)", buf, n_cu, ghz, 16);
    run<2>("global_load_dwordx2 (8 B/lane)", buf, n_cu, ghz, 8);
    run<3>("global_load_dwordx4 (16 B/lane)", buf, n_cu, ghz, 16);
    return 0;
}
