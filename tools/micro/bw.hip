// Streaming bandwidth by access width (developer microbenchmark, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bw.hip -o /tmp/bw && /tmp/bw
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ void k_copy(const T* __restrict__ a, T* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <typename T> __global__ void k_write(T* __restrict__ b, size_t n, T v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = v;
}
template <typename T> __global__ void k_read(const T* __restrict__ a, size_t n, T* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { T v = a[i]; if (((const int*)&v)[0] == 0x12345678) out[0] = v; }
}
// 8 tracks of doubles written per "base", like the scan kernel: each block writes 1000 bases x 8 tracks
__global__ void k_tracks8(double* __restrict__ b, size_t total) {
    size_t base = (size_t)blockIdx.x * 1000 + threadIdx.x;
    if (threadIdx.x < 1000)
        for (int s = 0; s < 8; ++s) b[s * total + base] = (double)s;
}
__global__ void k_tracks16(double2* __restrict__ b, size_t total) {   // 2 bases per lane, 500 lanes
    size_t base = (size_t)blockIdx.x * 500 + threadIdx.x;
    if (threadIdx.x < 500)
        for (int s = 0; s < 8; ++s) b[s * (total / 2) + base] = make_double2(s, s);
}
// the scan kernel's traffic and nothing else: per 1000-base interval read 2 x 1111 doubles + 1117
// bytes, write 8 tracks of 1000 doubles
__global__ void k_scanlike(const double* __restrict__ cp, const double* __restrict__ cm, const unsigned char* __restrict__ sq,
                           double* __restrict__ out, size_t total) {
    const size_t iv = blockIdx.x;
    const int tid = threadIdx.x;
    double acc = 0;
    for (int v = tid; v < 1111; v += 1024) acc += cp[iv * 1111 + v] + cm[iv * 1111 + v];
    for (int v = tid; v < 1117; v += 1024) acc += sq[iv * 1117 + v];
    if (tid < 1000)
        for (int s = 0; s < 8; ++s) out[s * total + iv * 1000 + tid] = acc + s;
}
#define T(name, bytes, ...) { hipEventRecord(e0); for (int r = 0; r < 5; ++r) { __VA_ARGS__; } hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-28s %8.1f GB/s\n", name, (double)(bytes) * 5 / ms / 1e6); }
int main() {
    size_t bytes = (size_t)4 << 30;
    void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
    size_t n8 = bytes / 8, n16 = bytes / 16, n4 = bytes / 4;
    T("copy 4B/lane", 2 * bytes, k_copy<float><<<n4 / 256, 256>>>((float*)a, (float*)b, n4));
    T("copy 8B/lane", 2 * bytes, k_copy<double><<<n8 / 256, 256>>>((double*)a, (double*)b, n8));
    T("copy 16B/lane", 2 * bytes, k_copy<double2><<<n16 / 256, 256>>>((double2*)a, (double2*)b, n16));
    T("write 8B/lane", bytes, k_write<double><<<n8 / 256, 256>>>((double*)b, n8, 1.0));
    T("write 16B/lane", bytes, k_write<double2><<<n16 / 256, 256>>>((double2*)b, n16, make_double2(1, 2)));
    T("read 8B/lane", bytes, k_read<double><<<n8 / 256, 256>>>((double*)a, n8, (double*)b));
    T("read 16B/lane", bytes, k_read<double2><<<n16 / 256, 256>>>((double2*)a, n16, (double2*)b));
    size_t nb = 60000; size_t total = nb * 1000;
    T("8 tracks, 8B/lane x1024", total * 64, k_tracks8<<<nb, 1024>>>((double*)b, total));
    T("8 tracks, 16B/lane x512", total * 64, k_tracks16<<<nb, 512>>>((double2*)b, total));
    {
        size_t nb = 400000, total = nb * 1000;   // 0.4e9 bases: 7.1 GB in, 25.6 GB out
        double *cp, *cm, *out; unsigned char* sq;
        hipMalloc(&cp, nb * 1111 * 8); hipMalloc(&cm, nb * 1111 * 8); hipMalloc(&sq, nb * 1117); hipMalloc(&out, total * 64);
        hipMemset(cp, 0, nb * 1111 * 8); hipMemset(cm, 0, nb * 1111 * 8); hipMemset(sq, 65, nb * 1117);
        T("scan-like traffic (82.9 B/base)", total * 82.887, k_scanlike<<<nb, 1024>>>(cp, cm, sq, out, total));
        hipFree(cp); hipFree(cm); hipFree(sq); hipFree(out);
    }
    }
    return 0;
}
