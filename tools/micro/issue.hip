// How a gfx950 SIMD issues instructions of different kinds from several wavefronts (developer
// microbenchmark, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/issue.hip -o /tmp/issue && /tmp/issue
// Every wavefront runs ITER rounds of a 64-instruction block; blocks are 256 lanes (one wavefront per
// SIMD), W blocks per CU give W wavefronts per SIMD.  Prints cycles per instruction and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R32(x) R16(x) R16(x)
#define R64(x) R16(R4(x))
template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, unsigned *out) {
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    double d0 = threadIdx.x, d1 = 1.0, d2 = 0.5, d3 = 2.0;
    const bool odd = (blockIdx.x & 1) != 0;
    unsigned long long sm = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(R16("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 1) asm volatile(R16("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        if (MODE == 2) asm volatile(R16("v_add_u32 %0, %0, %0\n s_add_u32 %2, %2, %2\n v_add_u32 %1, %1, %1\n s_add_u32 %3, %3, %3\n") : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) : : "scc");
        if (MODE == 3) {  // half the wavefronts vector-only, the other half scalar-only
            if (odd) asm volatile(R16("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            else asm volatile(R16("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
        if (MODE == 4) asm volatile(R16("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        if (MODE == 5) asm volatile(R16("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 6) asm volatile(R16("v_add_u32 %0, %0, %0\n s_nop 0\n v_add_u32 %1, %1, %1\n s_nop 0\n") : "+v"(v0), "+v"(v1));
        if (MODE == 7) asm volatile(R16("v_add_u32 %0, %0, %0\n v_readlane_b32 %2, %0, 3\n s_add_u32 %2, %2, %2\n v_add_u32 %1, %1, %2\n") : "+v"(v0), "+v"(v1), "+s"(s0) : : "scc");
        if (MODE == 8) asm volatile(R16("v_cmp_lt_u32 vcc, %0, %1\n s_and_b64 %2, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %1, %1, %0\n") : "+v"(v0), "+v"(v1), "+s"(sm) : : "vcc", "scc");
        if (MODE == 10) {  // 32 x 32 -> 64 multiply-add (what a Philox round is made of)
            unsigned long long m0 = v0, m1 = v1, m2 = v2, m3 = v3;
            asm volatile(R16("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %5, %6, 0\n v_mad_u64_u32 %2, vcc, %6, %7, 0\n v_mad_u64_u32 %3, vcc, %7, %4, 0\n") : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "vcc");
            v0 += (unsigned)(m0 + m1 + m2 + m3);
        }
        if (MODE == 11) asm volatile(R16("v_mul_lo_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 12) asm volatile(R16("v_mul_u32_u24 %0, %0, %1\n v_mul_hi_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_hi_u32_u24 %3, %3, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (MODE == 13) asm volatile(R16("v_cvt_f64_u32 %0, %2\n v_cvt_u32_f64 %2, %0\n v_cvt_f64_u32 %1, %3\n v_cvt_u32_f64 %3, %1\n") : "+v"(d0), "+v"(d1), "+v"(v0), "+v"(v1));
        if (MODE == 14) asm volatile(R16("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        if (MODE == 15) asm volatile(R16("v_cmp_le_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_le_f64 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n") : "+v"(d0), "+v"(d1), "+v"(v0), "+v"(v1) : : "vcc");
        if (MODE == 9) asm volatile(R16("v_fma_f64 %0, %0, %0, %0\n s_add_u32 %2, %2, %2\n v_fma_f64 %1, %1, %1, %1\n s_add_u32 %3, %3, %3\n") : "+v"(d0), "+v"(d1), "+s"(s0), "+s"(s1) : : "scc");
    }
    if (v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 + (unsigned)sm == 0x12345 && d0 + d1 + d2 + d3 == 7.0) out[0] = 1;
}
template <int MODE> void run(const char *name, unsigned *out, int n_cu, double ghz) {
    printf("%-44s", name);
    for (int w : {1, 2, 4, 8}) {
        const int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(n_cu * w), dim3(256), 0, 0, 10, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(n_cu * w), dim3(256), 0, 0, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // cycles per instruction of ONE wavefront's stream, and per instruction and SIMD (w wavefronts share it)
        const double cyc = ms * 1e-3 * ghz * 1e9, per_wave = cyc / (iters * 64.0);
        printf("  w=%d: %5.2f/wave-instr %5.2f/simd-instr", w, per_wave, per_wave / w);
    }
    printf("\n");
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz nominal (cycles below assume it)\n", p.gcnArchName, n_cu, ghz);
    unsigned *out; hipMalloc(&out, 64);
    run<0>("v_add_u32 x4 chains", out, n_cu, ghz);
    run<1>("s_add_u32 x4 chains", out, n_cu, ghz);
    run<2>("v_add / s_add alternating in each wave", out, n_cu, ghz);
    run<3>("half the waves v_add, half s_add", out, n_cu, ghz);
    run<4>("v_fma_f64 x4 chains", out, n_cu, ghz);
    run<5>("v_add_u32_dpp x4 chains", out, n_cu, ghz);
    run<6>("v_add / s_nop 0 alternating", out, n_cu, ghz);
    run<7>("v_add, v_readlane, s_add, v_add(sgpr) dependent", out, n_cu, ghz);
    run<8>("v_cmp, s_and, v_cndmask, v_add dependent", out, n_cu, ghz);
    run<9>("v_fma_f64 / s_add alternating in each wave", out, n_cu, ghz);
    run<10>("v_mad_u64_u32 x4", out, n_cu, ghz);
    run<11>("v_mul_lo_u32 / v_mul_hi_u32", out, n_cu, ghz);
    run<12>("v_mul_u32_u24 / v_mul_hi_u32_u24", out, n_cu, ghz);
    run<13>("v_cvt_f64_u32 / v_cvt_u32_f64", out, n_cu, ghz);
    run<14>("v_add_f64 x4 chains", out, n_cu, ghz);
    run<15>("v_cmp_le_f64 + v_cndmask", out, n_cu, ghz);
    return 0;
}
