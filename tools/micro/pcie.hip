// PCIe link rates of the box: pinned host <-> device copies of 64 MB, alone and in both directions at once
// (two streams), and a pageable source.  hipcc --offload-arch=gfx950 -O2 tools/micro/pcie.hip -o tools/micro/pcie.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
            return 1;                                                             \
        }                                                                         \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t B = (size_t)64 << 20;
    const int R = 16;
    char *h_a, *h_b, *d_a, *d_b;
    CK(hipHostMalloc((void **)&h_a, B, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&h_b, B, hipHostMallocDefault));
    CK(hipMalloc((void **)&d_a, B));
    CK(hipMalloc((void **)&d_b, B));
    memset(h_a, 1, B);
    memset(h_b, 2, B);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int w = 0; w < 2; ++w) {
        CK(hipMemcpyAsync(d_a, h_a, B, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(h_b, d_b, B, hipMemcpyDeviceToHost, s2));
    }
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int i = 0; i < R; ++i) CK(hipMemcpyAsync(d_a, h_a, B, hipMemcpyHostToDevice, s1));
    CK(hipDeviceSynchronize());
    double t1 = now();
    printf("H2D alone        %.1f GB/s\n", R * B / (t1 - t0) / 1e9);
    t0 = now();
    for (int i = 0; i < R; ++i) CK(hipMemcpyAsync(h_b, d_b, B, hipMemcpyDeviceToHost, s2));
    CK(hipDeviceSynchronize());
    t1 = now();
    printf("D2H alone        %.1f GB/s\n", R * B / (t1 - t0) / 1e9);
    t0 = now();
    for (int i = 0; i < R; ++i) {
        CK(hipMemcpyAsync(d_a, h_a, B, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(h_b, d_b, B, hipMemcpyDeviceToHost, s2));
    }
    CK(hipDeviceSynchronize());
    t1 = now();
    printf("both at once     %.1f GB/s each way, %.1f GB/s together\n", R * B / (t1 - t0) / 1e9, 2 * R * B / (t1 - t0) / 1e9);
    // small pieces (what a pipeline chunk's four output tracks are)
    const size_t P = (size_t)16 << 20;
    t0 = now();
    for (int i = 0; i < R; ++i)
        for (int k = 0; k < 4; ++k) {
            CK(hipMemcpyAsync(d_a + k * P, h_a + k * P, P, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(h_b + k * P, d_b + k * P, P, hipMemcpyDeviceToHost, s2));
        }
    CK(hipDeviceSynchronize());
    t1 = now();
    printf("both, 16 MB pieces %.1f GB/s each way\n", R * B / (t1 - t0) / 1e9);
    char *pg = (char *)malloc(B);
    memset(pg, 3, B);
    t0 = now();
    for (int i = 0; i < 4; ++i) CK(hipMemcpy(d_a, pg, B, hipMemcpyHostToDevice));
    t1 = now();
    printf("H2D pageable     %.1f GB/s\n", 4 * B / (t1 - t0) / 1e9);
    t0 = now();
    for (int i = 0; i < 4; ++i) CK(hipMemcpy(pg, d_a, B, hipMemcpyDeviceToHost));
    t1 = now();
    printf("D2H pageable     %.1f GB/s\n", 4 * B / (t1 - t0) / 1e9);
    // one core's memcpy, pageable -> pinned
    t0 = now();
    for (int i = 0; i < 4; ++i) memcpy(h_a, pg, B);
    t1 = now();
    printf("memcpy 1 thread  %.1f GB/s\n", 4 * B / (t1 - t0) / 1e9);
    // hipHostMalloc / hipHostRegister cost
    t0 = now();
    char *big;
    CK(hipHostMalloc((void **)&big, (size_t)1 << 30, hipHostMallocDefault));
    t1 = now();
    printf("hipHostMalloc 1 GiB: %.1f ms\n", (t1 - t0) * 1e3);
    char *reg = (char *)malloc((size_t)1 << 30);
    memset(reg, 0, (size_t)1 << 30);
    t0 = now();
    CK(hipHostRegister(reg, (size_t)1 << 30, hipHostRegisterDefault));
    t1 = now();
    printf("hipHostRegister 1 GiB (touched): %.1f ms\n", (t1 - t0) * 1e3);
    t0 = now();
    CK(hipHostUnregister(reg));
    t1 = now();
    printf("hipHostUnregister 1 GiB: %.1f ms\n", (t1 - t0) * 1e3);
    return 0;
}
