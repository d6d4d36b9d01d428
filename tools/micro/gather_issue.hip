// Cost of divergent gathers on gfx950 (developer microbenchmark, not part of the library): every lane reads a
// pseudo-random element of a table that fits the vector L1 / the L2 (the alias and z tables of the FDR draws:
// ~20 rows of 0.5-1 KB in use per interval), W wavefronts per SIMD.  Prints CU-cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_issue.hip -o /tmp/gather && /tmp/gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// ACTIVE: lanes (of 64, a different random subset at every instruction) that take part in the gather
template <int BYTES, int SPREAD, int ACTIVE = 64>  // SPREAD: 0 = random over the table, 1 = random inside one 128-byte line per 16 lanes, 2 = coalesced
__global__ void __launch_bounds__(256) k(int iters, const char *tab, uint32_t mask, uint32_t *out) {
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            uint32_t idx = (x >> 8);
            if (SPREAD == 1) idx = ((threadIdx.x >> 4) * 977u + (i * 8 + j) * 131u) * (128 / BYTES) + (idx & (128 / BYTES - 1));
            if (SPREAD == 2) idx = threadIdx.x + (i * 8 + j) * 64;
            const uint32_t off = (idx * BYTES) & mask & ~(uint32_t)(BYTES - 1);
            if (ACTIVE < 64 && ((x >> 3) & 63u) >= (uint32_t)ACTIVE) continue;
            if (BYTES == 4) { acc += *reinterpret_cast<const uint32_t *>(tab + off); asm volatile("" ::: "memory"); }
            if (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(tab + off); acc += v.x ^ v.y; asm volatile("" ::: "memory"); }
            if (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4 *>(tab + off); acc += v.x ^ v.y ^ v.z ^ v.w; asm volatile("" ::: "memory"); }
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}
template <int BYTES, int SPREAD, int ACTIVE = 64> void run(const char *name, const char *tab, uint32_t table_bytes, uint32_t *out, int n_cu, double ghz) {
    printf("%-58s", name);
    for (int w : {2, 8}) {
        const int iters = 500;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<BYTES, SPREAD, ACTIVE>), dim3(n_cu * w), dim3(256), 0, 0, 10, tab, table_bytes - 1, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<BYTES, SPREAD, ACTIVE>), dim3(n_cu * w), dim3(256), 0, 0, iters, tab, table_bytes - 1, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("  w=%d: %6.1f cyc/inst/CU", w, cyc / ((double)iters * 8 * 4 * w));
    }
    printf("\n");
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    char *tab; hipMalloc(&tab, 1 << 22); hipMemset(tab, 1, 1 << 22);
    uint32_t *out; hipMalloc(&out, 64);
    printf("%s, %d CUs, %.2f GHz nominal (8 v_mad-class instructions per gather besides)\n", p.gcnArchName, n_cu, ghz);
    run<4, 0>("4 B, random over 16 KB", tab, 1 << 14, out, n_cu, ghz);
    run<8, 0>("8 B, random over 16 KB", tab, 1 << 14, out, n_cu, ghz);
    run<16, 0>("16 B, random over 16 KB", tab, 1 << 14, out, n_cu, ghz);
    run<8, 0, 48>("8 B, random over 16 KB, 48 of 64 lanes", tab, 1 << 14, out, n_cu, ghz);
    run<8, 0, 32>("8 B, random over 16 KB, 32 of 64 lanes", tab, 1 << 14, out, n_cu, ghz);
    run<8, 0, 16>("8 B, random over 16 KB, 16 of 64 lanes", tab, 1 << 14, out, n_cu, ghz);
    run<8, 0, 8>("8 B, random over 16 KB, 8 of 64 lanes", tab, 1 << 14, out, n_cu, ghz);
    run<16, 0, 32>("16 B, random over 16 KB, 32 of 64 lanes", tab, 1 << 14, out, n_cu, ghz);
    run<4, 0>("4 B, random over 256 KB (L2)", tab, 1 << 18, out, n_cu, ghz);
    run<8, 0>("8 B, random over 256 KB (L2)", tab, 1 << 18, out, n_cu, ghz);
    run<16, 0>("16 B, random over 256 KB (L2)", tab, 1 << 18, out, n_cu, ghz);
    run<4, 1>("4 B, four lines per wavefront (16 KB)", tab, 1 << 14, out, n_cu, ghz);
    run<8, 1>("8 B, four lines per wavefront (16 KB)", tab, 1 << 14, out, n_cu, ghz);
    run<4, 2>("4 B, coalesced (16 KB)", tab, 1 << 14, out, n_cu, ghz);
    run<8, 2>("8 B, coalesced (16 KB)", tab, 1 << 14, out, n_cu, ghz);
    run<16, 2>("16 B, coalesced (16 KB)", tab, 1 << 14, out, n_cu, ghz);
    return 0;
}
