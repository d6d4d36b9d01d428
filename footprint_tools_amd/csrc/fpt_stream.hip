// fpt_stream.hip -- the fused scan's HBM access pattern and nothing else (gfx950).
//
// "record the box's measured stream-copy BW beside [the 8 TB/s peak]".  A plain
// device-to-device copy is not what the scan does to the memory system: per interval it reads two
// arrays of L + 2*pad + 1 doubles and L + 2*pad + 7 sequence bytes that start wherever the interval
// before ended, and writes 3 + S tracks of L doubles that lie total_bases apart.  k_stream_pattern
// issues exactly those loads and stores -- the same workgroup per interval (or per tile of a long
// one), the same lane-to-address map as k_scan_lean -- with one add per load between them, so its
// rate is what THIS box gives THIS pattern when arithmetic costs nothing.  bench.py times it in
// the invocation that times the scan and reports roofline.box_stream_GBps / frac_of_box: a slow
// box moves both numbers, a slow kernel only one.
//
// Reference counterpart: none (measurement infrastructure of the path cli/detect.py:120-130).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/fpt.h"

int fpt_internal_fail(int code, const char *fmt, ...);  // fpt_capi.cpp
hipStream_t fpt_internal_stream(fpt_ctx *c);
int fpt_internal_check_ctx(fpt_ctx *c);

namespace {

struct stream_args {
    int64_t n_intervals;
    int32_t interval_len;         // uniform batches (interval_off == nullptr)
    const int64_t *interval_off;  // ragged: n_intervals + 1 output offsets
    int32_t pad2;                 // 2 * (hw + shw)
    int32_t n_tracks;
    int64_t total_bases;
    const double *cp, *cm;
    const uint8_t *sq;
    double *out;
};

// one workgroup per interval; lanes stride over the padded positions, then over the bases
template <int NT>
__global__ void __launch_bounds__(NT) k_stream_pattern(const stream_args a) {
    const int64_t iv = blockIdx.x;
    int64_t off;
    int L;
    if (a.interval_off) {
        typedef const __attribute__((address_space(4))) int64_t koff;
        off = ((koff *)a.interval_off)[iv];
        L = (int)(((koff *)a.interval_off)[iv + 1] - off);
    } else {
        L = a.interval_len;
        off = iv * (int64_t)L;
    }
    const int nc = L + a.pad2 + 1;
    const double *cp = a.cp + off + iv * (int64_t)(a.pad2 + 1);
    const double *cm = a.cm + off + iv * (int64_t)(a.pad2 + 1);
    const uint8_t *sq = a.sq + off + iv * (int64_t)(a.pad2 + 7);
    double acc = 0.0;
    for (int v = threadIdx.x; v < nc + 6; v += NT) {
        if (v < nc) acc += cp[v] + cm[v];
        acc += (double)sq[v];
    }
    double *o = a.out + off;
    for (int t = threadIdx.x; t < L; t += NT)
        for (int s = 0; s < a.n_tracks; ++s) o[(int64_t)s * a.total_bases + t] = acc + (double)s;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int fpt_stream_pattern_dev(fpt_ctx *c, int64_t n_intervals, int32_t interval_len, const int64_t *interval_off_dev,
                           int32_t max_interval_len, int32_t pad, int32_t n_tracks, const double *counts_plus,
                           const double *counts_minus, const uint8_t *seq, double *out, int64_t total_bases,
                           int32_t reps, float *ms_out) {
    if (int rc = fpt_internal_check_ctx(c)) return rc;
    if (n_intervals < 1 || n_intervals > 0x7fffffff || pad < 0 || n_tracks < 1 || n_tracks > 3 + FPT_MAX_SCALES ||
        reps < 1 || !counts_plus || !counts_minus || !seq || !out || !ms_out || total_bases < 1)
        return fpt_internal_fail(FPT_ERR_INVALID, "fpt_stream_pattern_dev: bad arguments");
    const int longest = interval_off_dev ? max_interval_len : interval_len;
    if (longest < 1) return fpt_internal_fail(FPT_ERR_INVALID, "fpt_stream_pattern_dev: interval length must be positive");
    stream_args a;
    a.n_intervals = n_intervals;
    a.interval_len = interval_len;
    a.interval_off = interval_off_dev;
    a.pad2 = 2 * pad;
    a.n_tracks = n_tracks;
    a.total_bases = total_bases;
    a.cp = counts_plus;
    a.cm = counts_minus;
    a.sq = seq;
    a.out = out;
    hipStream_t st = fpt_internal_stream(c);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return fpt_internal_fail(FPT_ERR_HIP, "fpt_stream_pattern_dev: hipEventCreate failed");
    // the workgroup the scan would use for the longest interval of a uniform batch; ragged batches
    // (mean ~160 bases) run 256 lanes per interval
    const int nt = interval_off_dev ? 256 : (longest <= 256 ? 256 : (longest <= 512 ? 512 : 1024));
    auto launch = [&]() {
        if (nt == 256) hipLaunchKernelGGL(k_stream_pattern<256>, dim3((unsigned)n_intervals), dim3(256), 0, st, a);
        else if (nt == 512) hipLaunchKernelGGL(k_stream_pattern<512>, dim3((unsigned)n_intervals), dim3(512), 0, st, a);
        else hipLaunchKernelGGL(k_stream_pattern<1024>, dim3((unsigned)n_intervals), dim3(1024), 0, st, a);
    };
    launch();  // warm-up: page tables, clocks
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, st);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fpt_internal_fail(FPT_ERR_HIP, "fpt_stream_pattern_dev: %s", hipGetErrorString(e));
    *ms_out = ms / (float)reps;
    return FPT_OK;
}

#pragma GCC visibility pop
}
