// fpt_capi.cpp -- the C ABI of include/fpt.h on top of the HIP kernels.
// Host-side only: argument checking, device workspace, H2D/D2H staging for the
// host-buffer entry points, tile tables and launch geometry for the fused scan.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fpt_host_threads.hpp"
#include "fpt_kernels.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(FPT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

constexpr int kSlots = 16;
constexpr int kModelDoubles = 24;

}  // namespace

struct fpt_host_pipe;                       // fpt_scan_host's streams, device buffers and pinned staging (end of this file)
static void host_pipe_free(fpt_host_pipe *);

struct fpt_ctx {
    int device = 0;
    fpt_host_pipe *pipe = nullptr;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    double *d_table = nullptr;   // 4097 doubles (slot 4096 = default)
    double *d_table2 = nullptr;  // 2 x 4096 doubles: the same table in the lean scan kernel's order
    double *d_models = nullptr;  // FPT_MAX_DISPERSION_MODELS * 24
    int *d_flags = nullptr;      // error flags
    unsigned long long *d_sum = nullptr;
    bool have_table = false;
    bool have_model[FPT_MAX_DISPERSION_MODELS] = {};
    int32_t *pin_plan = nullptr;  // pinned staging of fpt_scan_dev's per-block class offsets (ragged batches), likewise
    size_t pin_plan_bytes = 0;
    hipEvent_t pin_plan_copied = nullptr;
    bool pin_plan_busy = false;
    int32_t *pin_list = nullptr;  // pinned staging of fpt_fdr_dev's interval lists, and the event behind its last copy
    size_t pin_list_bytes = 0;
    hipEvent_t pin_list_copied = nullptr;
    bool pin_list_busy = false;
    void *ws[kSlots] = {};
    size_t ws_bytes[kSlots] = {};
    uint64_t ws_gen[kSlots] = {};  // counts the (re)allocations of a slot: its contents are gone after one
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    std::vector<hipEvent_t> tev;  // event pairs of recorded scans (fpt_timing_enable)
    std::vector<hipEvent_t> marks;  // fpt_mark
    int tev_used = 0;
    int n_cu = 0;
    int memo_exp = 256, memo_obs = 256;
    // second-level table of the redo pass: capacity (entries beyond what a batch needs are never
    // computed), 16 bytes per entry and model; batches with more models than memo2_models skip it
    int memo2_rows = 4096, memo2_stride = 4096, memo2_models = 4;
    // The second-level table is a pure function of the dispersion models, and unlike the 256 x 256
    // first level it is too large to rebuild per call, so it is KEPT: d_flags[16..17] hold the
    // bounds (exp, obs) up to which it is filled for the models described by m2_*; a call that
    // misses beyond them extends it between its two passes.  The lean first pass reads it, so only
    // the first batch that reaches a new (exp, obs) range pays the redo pass for it.  Changing a
    // model of the range (fpt_set_dispersion with other values), another model range or a
    // reallocated slot empties it.
    uint64_t model_epoch[FPT_MAX_DISPERSION_MODELS] = {};
    double h_models[FPT_MAX_DISPERSION_MODELS][kModelDoubles] = {};
    uint64_t epoch_next = 1;
    int m2_dm_id = -1, m2_n_dm = 0, m2_memo_exp = 0, m2_memo_obs = 0;
    uint64_t m2_epoch[FPT_MAX_DISPERSION_MODELS] = {};
    uint64_t m2_ws_gen = 0;
    bool m2_extended = false;  // the last scan call launched k_nb_memo2: its misses become bounds at the next call
    bool posterior_direct = false;  // FPT_POSTERIOR_TABLES=0: every log-pmf evaluated in the kernel (tests compare the two)
    bool memo2_cold = false;  // FPT_MEMO2_KEEP=0: the second-level table is emptied at every call (measurements)
    bool fdr_split = true;  // fpt_fdr_dev: the per-interval set-up as a launch of its own (FPT_FDR_SPLIT=0: one launch)
    bool fdr_light = true;  // ... and the draws by the light instance first, the full one for what it leaves (FPT_FDR_LIGHT=0: full only)
    bool fdr_slices_always = false;  // (tests: also for calls of a few draws)
    bool fdr_slices = true;   // ... and, in ragged batches, intervals of more than 256 bases drawn as slices (FPT_FDR_SLICES=0: one workgroup each)
    bool fdr_light_dbuf = false;  // the light instance with two sets of z buffers (FPT_FDR_LIGHT_DBUF=1)
    bool use_lean = true;  // first pass of memo mode by k_scan_lean (FPT_SCAN_LEAN=0: the general memo-only instance)
    // size classes of a batch's tiles (k_scan_lean's workgroup sizes)
    fptk::lean_class_set classes = fptk::make_lean_classes();
    bool table_lds = false;  // general kernel: bias table staged in LDS per workgroup (FPT_TABLE_LDS=1), read at creation
    // The null sampler's table reaches further in obs: a draw beyond the table costs a gallop +
    // bisection on the direct cdf (tens of incbet evaluations), and with 100 draws per base even
    // the 1e-4 tail of the widest rows is hit in every batch.
    int fdr_memo_obs = 2048;
    // fpt_segment_count_dev -> fpt_segment_fill_dev hand-over
    int64_t seg_n_intervals = -1, seg_total = 0;
    const double *seg_track = nullptr;
    // tile-table cache of the last ragged batch (reused while the offsets and geometry match)
    int64_t plan_tiles = 0;
    int plan_max_len = 0;  // longest interval of the ragged batch in hand
    std::vector<double> beta_host;  // fpt_posterior_dev: source of its asynchronous upload
    int64_t plan_cls_count[fptk::kLeanClasses] = {};  // tiles per workgroup-size class, in table order
    int64_t last_tiles = 0;      // tiles of the most recent memo-mode scan (fpt_scan_stats)
    bool last_has_redo = false;
};

namespace {

int ws_get(fpt_ctx *c, int slot, size_t bytes, void **out) {
    if (bytes == 0) bytes = 16;
    if (c->ws_bytes[slot] < bytes) {
        if (c->ws[slot]) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipFree(c->ws[slot]));
            c->ws[slot] = nullptr;
            c->ws_bytes[slot] = 0;
        }
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&c->ws[slot], want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            want = bytes;
            e = hipMalloc(&c->ws[slot], want);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(FPT_ERR_NOMEM, "device allocation of %zu bytes failed", bytes);
        }
        c->ws_bytes[slot] = want;
        ++c->ws_gen[slot];
    }
    *out = c->ws[slot];
    return FPT_OK;
}

int check_ctx(fpt_ctx *c) {
    if (!c) return fail(FPT_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return FPT_OK;
}

int launch_ok(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FPT_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return FPT_OK;
}

// `(int)((double)w * clip)` of smoothing.h:112
int trim_k(int shw, double clip) {
    int w = shw * 2 + 1;
    return (int)((double)w * clip);
}

}  // namespace

// shared with fpt_comm.cpp (hidden visibility: not part of the C ABI)
int fpt_internal_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
hipStream_t fpt_internal_stream(fpt_ctx *c) { return c->stream; }
int fpt_internal_check_ctx(fpt_ctx *c) { return check_ctx(c); }

extern "C" {
#pragma GCC visibility push(default)

const char *fpt_last_error(void) { return g_err.c_str(); }

int fpt_version(void) { return 100; }

int fpt_device_count(int *n_out) {
    if (!n_out) return fail(FPT_ERR_INVALID, "null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *n_out = n;
    return FPT_OK;
}

int fpt_ctx_create(int device_id, fpt_ctx **out) {
    if (!out) return fail(FPT_ERR_INVALID, "null output");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(FPT_ERR_NODEVICE,
                    "no HIP device visible: libfpt_hip has no CPU fallback (needs an MI355X / gfx950)");
    }
    if (device_id < 0 || device_id >= n)
        return fail(FPT_ERR_INVALID, "device %d out of range (%d visible)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FPT_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only",
                    device_id, prop.gcnArchName);
    fpt_ctx *c = new fpt_ctx();
    if (const char *e = getenv("FPT_SCAN_LEAN")) c->use_lean = atoi(e) != 0;
    if (const char *e = getenv("FPT_MEMO2_KEEP")) c->memo2_cold = atoi(e) == 0;
    if (const char *e = getenv("FPT_POSTERIOR_TABLES")) c->posterior_direct = atoi(e) == 0;
    if (const char *e = getenv("FPT_TABLE_LDS")) c->table_lds = atoi(e) != 0;
    if (const char *e = getenv("FPT_FDR_SPLIT")) c->fdr_split = atoi(e) != 0;
    if (const char *e = getenv("FPT_FDR_LIGHT")) c->fdr_light = atoi(e) != 0;
    if (const char *e = getenv("FPT_FDR_SLICES")) c->fdr_slices = atoi(e) != 0, c->fdr_slices_always = atoi(e) == 2;
    if (const char *e = getenv("FPT_FDR_LIGHT_DBUF")) c->fdr_light_dbuf = atoi(e) != 0;
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    // any failure below releases what was created so far (fpt_ctx_destroy skips null members)
    auto init = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        HIP_TRY(hipMalloc(&c->d_table, (FPT_KMER_TABLE + 1) * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_table2, 2 * FPT_KMER_TABLE * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_models, FPT_MAX_DISPERSION_MODELS * kModelDoubles * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_flags, 24 * sizeof(int)));
        HIP_TRY(hipMalloc(&c->d_sum, 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(c->d_flags, 0, 24 * sizeof(int)));
        HIP_TRY(hipEventCreate(&c->ev0));
        HIP_TRY(hipEventCreate(&c->ev1));
        return FPT_OK;
    };
    if (int rc = init()) {
        const std::string keep = g_err;
        fpt_ctx_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return FPT_OK;
}

int fpt_ctx_destroy(fpt_ctx *c) {
    if (!c) return FPT_OK;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->stream);
    host_pipe_free(c->pipe);
    for (int i = 0; i < kSlots; ++i)
        if (c->ws[i]) (void)hipFree(c->ws[i]);
    if (c->d_table) (void)hipFree(c->d_table);
    if (c->d_table2) (void)hipFree(c->d_table2);
    if (c->d_models) (void)hipFree(c->d_models);
    if (c->d_flags) (void)hipFree(c->d_flags);
    if (c->d_sum) (void)hipFree(c->d_sum);
    if (c->pin_plan) (void)hipHostFree(c->pin_plan);
    if (c->pin_plan_copied) (void)hipEventDestroy(c->pin_plan_copied);
    if (c->pin_list) (void)hipHostFree(c->pin_list);
    if (c->pin_list_copied) (void)hipEventDestroy(c->pin_list_copied);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->tev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->marks) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return FPT_OK;
}

int fpt_ctx_set_stream(fpt_ctx *c, void *hip_stream) {
    if (int rc = check_ctx(c)) return rc;
    if (c->stream != (hipStream_t)hip_stream) {
        c->m2_dm_id = -1;  // the kept second-level table was filled in the order of the old stream
        c->m2_extended = false;
    }
    c->stream = (hipStream_t)hip_stream;  // NULL = the device's default stream, as handed over
    return FPT_OK;
}

int fpt_ctx_use_own_stream(fpt_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    if (c->stream != c->own_stream) {
        c->m2_dm_id = -1;
        c->m2_extended = false;
    }
    c->stream = c->own_stream;
    return FPT_OK;
}

int fpt_ctx_synchronize(fpt_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FPT_OK;
}

int fpt_set_bias_table(fpt_ctx *c, const double *table4096, double dflt) {
    if (int rc = check_ctx(c)) return rc;
    if (!table4096) return fail(FPT_ERR_INVALID, "null table");
    std::vector<double> t(FPT_KMER_TABLE + 1);
    std::memcpy(t.data(), table4096, FPT_KMER_TABLE * sizeof(double));
    t[FPT_KMER_TABLE] = dflt;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->d_table, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> t2(2 * FPT_KMER_TABLE);
    fptk::build_lean_table(table4096, t2.data());
    HIP_TRY(hipMemcpy(c->d_table2, t2.data(), t2.size() * sizeof(double), hipMemcpyHostToDevice));
    c->have_table = true;
    return FPT_OK;
}

int fpt_set_dispersion(fpt_ctx *c, int dm_id, const double *mu9, const double *r15) {
    if (int rc = check_ctx(c)) return rc;
    if (dm_id < 0 || dm_id >= FPT_MAX_DISPERSION_MODELS)
        return fail(FPT_ERR_INVALID, "dm_id %d out of range", dm_id);
    if (!mu9 || !r15) return fail(FPT_ERR_INVALID, "null parameters");
    double m[kModelDoubles];
    std::memcpy(m, mu9, 9 * sizeof(double));
    std::memcpy(m + 9, r15, 15 * sizeof(double));
    if (c->have_model[dm_id] && std::memcmp(m, c->h_models[dm_id], sizeof m) == 0) return FPT_OK;  // unchanged
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->d_models + (size_t)dm_id * kModelDoubles, m, sizeof m, hipMemcpyHostToDevice));
    std::memcpy(c->h_models[dm_id], m, sizeof m);
    c->model_epoch[dm_id] = c->epoch_next++;
    c->have_model[dm_id] = true;
    return FPT_OK;
}

// ---------------------------------------------------------------- host-buffer entry points

int fpt_kmer_probs(fpt_ctx *c, const uint8_t *seq, int64_t seq_len, double *fwd, double *rev) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->have_table) return fail(FPT_ERR_INVALID, "bias table not set");
    if (seq_len < 0 || (!seq && seq_len > 0)) return fail(FPT_ERR_INVALID, "bad sequence");
    int64_t n = seq_len - 6;
    if (n <= 0) return FPT_OK;
    void *d_seq, *d_f, *d_r;
    if (int rc = ws_get(c, 0, (size_t)seq_len, &d_seq)) return rc;
    if (int rc = ws_get(c, 1, (size_t)n * 8, &d_f)) return rc;
    if (int rc = ws_get(c, 2, (size_t)n * 8, &d_r)) return rc;
    HIP_TRY(hipMemcpyAsync(d_seq, seq, (size_t)seq_len, hipMemcpyHostToDevice, c->stream));
    fptk::launch_kmer_probs(c->stream, (const uint8_t *)d_seq, n, c->d_table, (double *)d_f,
                            (double *)d_r);
    if (int rc = launch_ok("k_kmer_probs")) return rc;
    if (fwd) HIP_TRY(hipMemcpyAsync(fwd, d_f, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    if (rev) HIP_TRY(hipMemcpyAsync(rev, d_r, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FPT_OK;
}

int fpt_predict(fpt_ctx *c, const double *obs, const double *probs, int64_t n_rows, int l, int hw,
                int shw, double clip, double *exp_out, double *win_out) {
    if (int rc = check_ctx(c)) return rc;
    if (n_rows < 0 || l < 0 || hw < 0 || shw < 0)
        return fail(FPT_ERR_INVALID, "negative size / window");
    if (shw > 4096) return fail(FPT_ERR_INVALID, "smoothing_half_win_width %d too large", shw);
    size_t n = (size_t)n_rows * (size_t)l;
    if (n == 0) return FPT_OK;
    if (!obs || !probs || !exp_out || !win_out) return fail(FPT_ERR_INVALID, "null buffer");
    int k = trim_k(shw, clip);
    if (shw > 0 && (k < 0 || 2 * k >= 2 * shw + 1))
        return fail(FPT_ERR_INVALID, "smoothing_clip %g trims the whole window", clip);
    void *d_o, *d_p, *d_e, *d_w;
    if (int rc = ws_get(c, 0, n * 8, &d_o)) return rc;
    if (int rc = ws_get(c, 1, n * 8, &d_p)) return rc;
    if (int rc = ws_get(c, 2, n * 8, &d_e)) return rc;
    if (int rc = ws_get(c, 3, n * 8, &d_w)) return rc;
    HIP_TRY(hipMemcpyAsync(d_o, obs, n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_p, probs, n * 8, hipMemcpyHostToDevice, c->stream));
    fptk::launch_predict_rows(c->stream, (const double *)d_o, (const double *)d_p, n_rows, l, hw, shw,
                              k, (double *)d_e, (double *)d_w);
    if (int rc = launch_ok("k_predict_rows")) return rc;
    HIP_TRY(hipMemcpyAsync(exp_out, d_e, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(win_out, d_w, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FPT_OK;
}

int fpt_nb_values(fpt_ctx *c, int what, int dm_id, const double *ex, const double *ob, int64_t n,
                  double *out) {
    if (int rc = check_ctx(c)) return rc;
    if (what < 0 || what > 2) return fail(FPT_ERR_INVALID, "bad `what` %d", what);
    if (dm_id < 0 || dm_id >= FPT_MAX_DISPERSION_MODELS || !c->have_model[dm_id])
        return fail(FPT_ERR_INVALID, "dispersion model %d not set", dm_id);
    if (n < 0) return fail(FPT_ERR_INVALID, "negative length");
    if (n == 0) return FPT_OK;
    if (!ex || !ob || !out) return fail(FPT_ERR_INVALID, "null buffer");
    void *d_e, *d_o, *d_r;
    if (int rc = ws_get(c, 0, (size_t)n * 8, &d_e)) return rc;
    if (int rc = ws_get(c, 1, (size_t)n * 8, &d_o)) return rc;
    if (int rc = ws_get(c, 2, (size_t)n * 8, &d_r)) return rc;
    HIP_TRY(hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
    HIP_TRY(hipMemcpyAsync(d_e, ex, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_o, ob, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    fptk::launch_nb_values(c->stream, what, c->d_models + (size_t)dm_id * kModelDoubles,
                           (const double *)d_e, (const double *)d_o, n, (double *)d_r, c->d_flags);
    if (int rc = launch_ok("k_nb_values")) return rc;
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(out, d_r, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&flag, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flag & 1) return fail(FPT_ERR_ZERODIV, "float division by zero in fit_r");
    return FPT_OK;
}

int fpt_nb_scalar(fpt_ctx *c, int what, const int32_t *k, const double *p, const double *r, int64_t n,
                  double *out) {
    if (int rc = check_ctx(c)) return rc;
    if (what < 0 || what > 2) return fail(FPT_ERR_INVALID, "bad `what` %d", what);
    if (n < 0) return fail(FPT_ERR_INVALID, "negative length");
    if (n == 0) return FPT_OK;
    if (!k || !p || !r || !out) return fail(FPT_ERR_INVALID, "null buffer");
    void *d_k, *d_p, *d_r, *d_o;
    if (int rc = ws_get(c, 0, (size_t)n * 4, &d_k)) return rc;
    if (int rc = ws_get(c, 1, (size_t)n * 8, &d_p)) return rc;
    if (int rc = ws_get(c, 2, (size_t)n * 8, &d_r)) return rc;
    if (int rc = ws_get(c, 3, (size_t)n * 8, &d_o)) return rc;
    HIP_TRY(hipMemcpyAsync(d_k, k, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_p, p, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_r, r, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    fptk::launch_nb_scalar(c->stream, what, (const int32_t *)d_k, (const double *)d_p,
                           (const double *)d_r, n, (double *)d_o);
    if (int rc = launch_ok("k_nb_scalar")) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FPT_OK;
}

int fpt_window(fpt_ctx *c, int op, const double *x, const double *w, int64_t n_rows, int n, int hw,
               double *out) {
    if (int rc = check_ctx(c)) return rc;
    if (op < 0 || op > 4) return fail(FPT_ERR_INVALID, "bad window op %d", op);
    if (n_rows < 0 || n < 0 || hw < 0) return fail(FPT_ERR_INVALID, "negative size / window");
    if (hw > 2048) return fail(FPT_ERR_INVALID, "half window %d too large", hw);
    size_t tot = (size_t)n_rows * (size_t)n;
    if (tot == 0) return FPT_OK;
    if (!x || !out || (op == FPT_WIN_WSTOUFFER && !w)) return fail(FPT_ERR_INVALID, "null buffer");
    void *d_x, *d_w = nullptr, *d_o;
    if (int rc = ws_get(c, 0, tot * 8, &d_x)) return rc;
    if (int rc = ws_get(c, 1, tot * 8, &d_o)) return rc;
    HIP_TRY(hipMemcpyAsync(d_x, x, tot * 8, hipMemcpyHostToDevice, c->stream));
    if (op == FPT_WIN_WSTOUFFER) {
        if (int rc = ws_get(c, 2, tot * 8, &d_w)) return rc;
        HIP_TRY(hipMemcpyAsync(d_w, w, tot * 8, hipMemcpyHostToDevice, c->stream));
    }
    fptk::launch_window_rows(c->stream, op, (const double *)d_x, (const double *)d_w, n_rows, n, hw,
                             (double *)d_o);
    if (int rc = launch_ok("k_window_rows")) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_o, tot * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FPT_OK;
}

int fpt_special(fpt_ctx *c, int fn, const double *a, const double *b, const double *x, int64_t n,
                double *out) {
    if (int rc = check_ctx(c)) return rc;
    if (fn < 0 || fn > 12) return fail(FPT_ERR_INVALID, "bad function id %d", fn);
    if (n < 0) return fail(FPT_ERR_INVALID, "negative length");
    if (n == 0) return FPT_OK;
    if (!a || !out) return fail(FPT_ERR_INVALID, "null buffer");
    if (fn == FPT_FN_INCBET && (!b || !x)) return fail(FPT_ERR_INVALID, "incbet needs a, b, x");
    if (fn == FPT_FN_CHDTRC && !x) return fail(FPT_ERR_INVALID, "chdtrc needs df (a) and x");
    void *d_a, *d_b, *d_x, *d_o;
    if (int rc = ws_get(c, 0, (size_t)n * 8, &d_a)) return rc;
    if (int rc = ws_get(c, 1, (size_t)n * 8, &d_b)) return rc;
    if (int rc = ws_get(c, 2, (size_t)n * 8, &d_x)) return rc;
    if (int rc = ws_get(c, 3, (size_t)n * 8, &d_o)) return rc;
    HIP_TRY(hipMemcpyAsync(d_a, a, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if (b) HIP_TRY(hipMemcpyAsync(d_b, b, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if (x) HIP_TRY(hipMemcpyAsync(d_x, x, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    fptk::launch_special(c->stream, fn, (const double *)d_a, (const double *)d_b, (const double *)d_x, n,
                         (double *)d_o);
    if (int rc = launch_ok("k_special")) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FPT_OK;
}

// ---------------------------------------------------------------- fused scan

int fpt_scan_dev(fpt_ctx *c, const fpt_scan_desc *d) {
    if (int rc = check_ctx(c)) return rc;
    if (!d) return fail(FPT_ERR_INVALID, "null descriptor");
    if (!c->have_table) return fail(FPT_ERR_INVALID, "bias table not set");
    const int n_dm = d->dm_ids ? d->n_dm : 1;
    if (n_dm < 1 || d->dm_id < 0 || d->dm_id + n_dm > FPT_MAX_DISPERSION_MODELS)
        return fail(FPT_ERR_INVALID, "dispersion model slots [%d, %d) out of range", d->dm_id, d->dm_id + n_dm);
    for (int i = 0; i < n_dm && d->nb_mode != FPT_NB_NONE; ++i)
        if (!c->have_model[d->dm_id + i])
            return fail(FPT_ERR_INVALID, "dispersion model %d not set", d->dm_id + i);
    if (d->n_intervals < 0) return fail(FPT_ERR_INVALID, "negative interval count");
    if (d->n_intervals == 0) return FPT_OK;
    const int hw = d->half_win_width, shw = d->smoothing_half_win_width;
    if (hw < 0 || shw < 0 || hw > 64 || shw > 1024)
        return fail(FPT_ERR_INVALID, "window widths out of range (hw=%d shw=%d)", hw, shw);
    if (d->n_scales < 0 || d->n_scales > FPT_MAX_SCALES)
        return fail(FPT_ERR_INVALID, "n_scales %d out of range", d->n_scales);
    int H = 0;
    for (int i = 0; i < d->n_scales; ++i) {
        if (d->scales[i] < 0 || d->scales[i] > 200)
            return fail(FPT_ERR_INVALID, "scale %d out of range", d->scales[i]);
        H = std::max(H, d->scales[i]);
    }
    if (!d->counts_plus || !d->counts_minus || !d->seq) return fail(FPT_ERR_INVALID, "null input");
    if (d->n_scales > 0 && !d->winp_out) return fail(FPT_ERR_INVALID, "winp_out is null");
    const int k = trim_k(shw, d->smoothing_clip);
    if (shw > 0 && (k < 0 || 2 * k >= 2 * shw + 1))
        return fail(FPT_ERR_INVALID, "smoothing_clip %g trims the whole window", d->smoothing_clip);
    const int pad = hw + shw;
    const int split_len = 1024 - 2 * H;  // tile length inside intervals longer than a workgroup

    fptk::scan_launch sl{};
    sl.n_intervals = d->n_intervals;
    sl.hw = hw;
    sl.shw = shw;
    sl.k_trim = k;
    sl.n_scales = d->n_scales;
    for (int i = 0; i < d->n_scales; ++i) sl.scales[i] = d->scales[i];
    sl.counts_plus = d->counts_plus;
    sl.counts_minus = d->counts_minus;
    sl.seq = d->seq;
    sl.table = c->d_table;
    sl.table2 = c->d_table2;
    sl.n_cu = c->n_cu;
    sl.model = c->d_models + (size_t)d->dm_id * kModelDoubles;
    sl.exp_out = d->exp_out;
    sl.obs_out = d->obs_out;
    sl.pval_out = d->pval_out;
    sl.winp_out = d->winp_out;
    sl.status_out = d->status_out;
    sl.dm_ids = d->dm_ids;

    // Tiles are grouped by the workgroup size that holds them.  The general kernel has three sizes
    // (256 / 512 / 1024 lanes); the lean first pass of memo mode has seven (fptk::kLeanNT), each a
    // sub-range of one of the three, so that short intervals leave fewer lanes idle.
    struct launch_t {
        int nt;
        int64_t first, count;
        int tile_len;
    };
    const fptk::lean_class_set &CS = c->classes;
    std::vector<launch_t> launches, lean_launches;

    auto nt_class = [](int n) { return n <= 256 ? 256 : (n <= 512 ? 512 : 1024); };
    auto lean_class = [&CS](int n) {
        int k = 0;
        while (CS.lmax[k] < n) ++k;
        return k;
    };

    if (!d->interval_off) {
        const int L = d->interval_len;
        if (L <= 0) return fail(FPT_ERR_INVALID, "interval_len must be positive");
        sl.interval_len = L;
        sl.total_bases = d->n_intervals * (int64_t)L;
        int tile_len = L <= 1024 ? L : split_len;
        int tpi = (L + tile_len - 1) / tile_len;
        sl.tiles_per_interval = tpi;
        int nt_needed = tpi == 1 ? L : std::min(L, tile_len + 2 * H);
        launches.push_back({nt_class(nt_needed), 0, d->n_intervals * (int64_t)tpi, tile_len});
        const int ucls = tpi == 1 ? lean_class(nt_needed) : std::max(lean_class(nt_needed), CS.first_split);
        lean_launches.push_back({CS.nt[ucls], 0, d->n_intervals * (int64_t)tpi, tile_len});
    } else {
        // ragged batch: tile table binned by workgroup size
        std::vector<int64_t> off_host;
        const int64_t *off = d->interval_off_host;
        if (!off) {
            off_host.resize(d->n_intervals + 1);
            HIP_TRY(hipMemcpyAsync(off_host.data(), d->interval_off,
                                   (size_t)(d->n_intervals + 1) * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            off = off_host.data();
        }
        if (d->n_intervals > 0x7fffffff) return fail(FPT_ERR_INVALID, "too many intervals");
        // one pass over the lengths: tiles per workgroup-size class (the grid sizes), where every
        // kPlanBlock intervals start inside each class, and the longest interval; the table itself is
        // made on the device (k_plan_tiles) -- a few KB are copied, nobody waits, and a job whose
        // batches all differ pays ~1 ns per interval here
        constexpr int NC = fptk::kLeanClasses, PB = fptk::kPlanBlock;
        if (1024 - H <= CS.lmax[NC - 2]) return fail(FPT_ERR_INVALID, "Stouffer half-width %d too large", H);
        const int64_t n_blocks = (d->n_intervals + PB - 1) / PB;
        const size_t stage_bytes = (size_t)std::max<int64_t>(n_blocks, 1) * NC * sizeof(int32_t);
        if (c->pin_plan_busy) {  // (the copy of the call before has long happened)
            HIP_TRY(hipEventSynchronize(c->pin_plan_copied));
            c->pin_plan_busy = false;
        }
        if (c->pin_plan_bytes < stage_bytes) {
            if (c->pin_plan) (void)hipHostFree(c->pin_plan);
            c->pin_plan = nullptr;
            c->pin_plan_bytes = 0;
            const size_t want = stage_bytes + stage_bytes / 4;
            if (hipHostMalloc((void **)&c->pin_plan, want, hipHostMallocDefault) != hipSuccess)
                return fail(FPT_ERR_NOMEM, "no pinned host memory for %zu bytes of the tile plan", want);
            c->pin_plan_bytes = want;
        }
        int64_t cls_count[NC] = {};
        int max_len = 0;
        bool bad_off = false;
        for (int64_t blk = 0; blk < n_blocks; ++blk) {
            int32_t *snap = c->pin_plan + blk * NC;
            for (int cls = 0; cls < NC; ++cls) snap[cls] = (int32_t)std::min<int64_t>(cls_count[cls], 0x7fffffff);
            const int64_t i1 = std::min<int64_t>((blk + 1) * PB, d->n_intervals);
            for (int64_t i = blk * PB; i < i1; ++i) {
                const int64_t L64 = off[i + 1] - off[i];
                bad_off |= L64 < 0 || L64 > 0x3fffffff;
                const int L = (int)L64;
                max_len = std::max(max_len, L);
                if (L > 1024 && !bad_off) {  // pieces of split_len bases, all but the last in the top class
                    const int n_full = (L - 1) / split_len;
                    cls_count[NC - 1] += n_full;
                    cls_count[std::max(lean_class(L - n_full * split_len + H), CS.first_split)] += 1;
                } else {  // (no branch on the length: the lengths of a real batch are not predictable)
                    int k = 0;
                    for (int q = 0; q < NC - 1; ++q) k += CS.lmax[q] < L ? 1 : 0;
                    cls_count[k] += L > 0 ? 1 : 0;
                }
            }
        }
        if (bad_off) return fail(FPT_ERR_INVALID, "bad interval offsets");
        int64_t n_tiles = 0;
        for (int cls = 0; cls < NC; ++cls) n_tiles += cls_count[cls];
        if (n_tiles > 0x7fffff00) return fail(FPT_ERR_INVALID, "too many tiles");
        // block_base: class-major table -> add where each class starts
        {
            int64_t at = 0;
            for (int cls = 0; cls < NC; ++cls) {
                if (at)
                    for (int64_t blk = 0; blk < n_blocks; ++blk) c->pin_plan[blk * NC + cls] += (int32_t)at;
                at += cls_count[cls];
            }
        }
        void *d_flat, *d_recs;
        if (int rc = ws_get(c, 9, ((size_t)n_tiles * 3 + 16) * 4 + stage_bytes, &d_flat)) return rc;
        if (int rc = ws_get(c, 12, std::max<size_t>((size_t)n_tiles, 1) * sizeof(fptk::lean_tile_rec), &d_recs)) return rc;
        int32_t *d_base = (int32_t *)d_flat + (size_t)n_tiles * 3 + 16;
        if (n_tiles > 0) {
            HIP_TRY(hipMemcpyAsync(d_base, c->pin_plan, stage_bytes, hipMemcpyHostToDevice, c->stream));
            if (!c->pin_plan_copied) HIP_TRY(hipEventCreateWithFlags(&c->pin_plan_copied, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(c->pin_plan_copied, c->stream));
            c->pin_plan_busy = true;
            fptk::launch_plan_tiles(c->stream, d->interval_off, d->n_intervals, n_tiles, H, split_len, CS, d_base,
                                    (int32_t *)d_flat, d_recs);
            if (int rc = launch_ok("k_plan_tiles")) return rc;
        }
        c->plan_tiles = n_tiles;
        c->plan_max_len = max_len;
        for (int cls = 0; cls < NC; ++cls) c->plan_cls_count[cls] = cls_count[cls];
        sl.total_bases = off[d->n_intervals];
        sl.interval_off = d->interval_off;
        sl.tile_iv = (const int32_t *)c->ws[9];
        sl.tile_t0 = sl.tile_iv + c->plan_tiles;
        sl.tile_tl = sl.tile_t0 + c->plan_tiles;
        sl.tile_recs = c->ws[12];
        int64_t first = 0;
        for (int cls = 0; cls < fptk::kLeanClasses; ++cls) {
            const int64_t n = c->plan_cls_count[cls];
            if (n > 0) {
                lean_launches.push_back({CS.nt[cls], first, n, split_len});
                const int nt = nt_class(CS.lmax[cls]);
                if (!launches.empty() && launches.back().nt == nt) launches.back().count += n;
                else launches.push_back({nt, first, n, split_len});
            }
            first += n;
        }
    }

    const bool rec = c->tev_used + 4 <= (int)c->tev.size();
    if (d->nb_mode < 0 || d->nb_mode > 3) return fail(FPT_ERR_INVALID, "bad nb_mode %d", d->nb_mode);
    if (d->nb_mode == FPT_NB_NONE && d->n_scales != 0)
        return fail(FPT_ERR_INVALID, "FPT_NB_NONE computes no p-values: n_scales must be 0");
    sl.counts_only = d->nb_mode == FPT_NB_NONE ? 1 : 0;
    const int64_t memo_n = (int64_t)c->memo_exp * c->memo_obs;
    const bool use_memo = d->nb_mode == FPT_NB_MEMO ||
                          (d->nb_mode == FPT_NB_AUTO && sl.total_bases >= 8 * memo_n);
    void *d_memo = nullptr;
    if (use_memo)
        if (int rc = ws_get(c, 8, (size_t)memo_n * 16 * n_dm, &d_memo)) return rc;
#ifdef FPT_ABLATE
    if (const char *e = getenv("FPT_ABLATE")) sl.ablate = atoi(e);
#endif
    // bias-table placement: through the L1/L2 caches by default (measured 25 % faster than a
    // per-workgroup LDS copy, which costs 33 KB of LDS and a third of the occupancy);
    // a context created under FPT_TABLE_LDS=1 uses the LDS-staged variant
    sl.table_global = c->table_lds ? 0 : 1;
    sl.memo = d_memo;
    sl.memo_exp = c->memo_exp;
    sl.memo_obs = c->memo_obs;

    // second-level table (built between the two passes, sized on the device by what the first
    // pass missed): only with the lean first pass, which records those maxima
    void *d_memo2 = nullptr;
    int32_t *d_miss = c->d_flags + 8;
    int memo2_state = 0;  // for k_nb_memo: 0 bounds as they are, 1 the last call's misses join them, 2 emptied
    if (use_memo && sl.table_global && c->use_lean && n_dm <= c->memo2_models && fptk::scan_lean_applies_hw(hw, shw, k)) {
        if (int rc = ws_get(c, 10, (size_t)c->memo2_rows * c->memo2_stride * 16 * n_dm, &d_memo2)) return rc;
        sl.memo2 = d_memo2;
        sl.memo2_max = d_miss;
        sl.memo2_have = c->d_flags + 16;
        sl.memo2_rows = c->memo2_rows;
        sl.memo2_stride = c->memo2_stride;
        // is what the slot holds the table of these models?
        bool same = c->m2_ws_gen == c->ws_gen[10] && c->m2_dm_id == d->dm_id && c->m2_n_dm == n_dm &&
                    c->m2_memo_exp == c->memo_exp && c->m2_memo_obs == c->memo_obs && !c->memo2_cold;
        for (int i = 0; i < n_dm && same; ++i) same = c->m2_epoch[i] == c->model_epoch[d->dm_id + i];
        memo2_state = !same ? 2 : (c->m2_extended ? 1 : 0);
        c->m2_ws_gen = c->ws_gen[10];
        c->m2_dm_id = d->dm_id;
        c->m2_n_dm = n_dm;
        c->m2_memo_exp = c->memo_exp;
        c->m2_memo_obs = c->memo_obs;
        for (int i = 0; i < n_dm; ++i) c->m2_epoch[i] = c->model_epoch[d->dm_id + i];
        c->m2_extended = false;
    }
    HIP_TRY(hipEventRecord(rec ? c->tev[c->tev_used] : c->ev0, c->stream));
    // memo mode runs two passes per size class: the light memo-only instance over every tile,
    // then the full instance over the tiles it flagged (early exit for the others)
    int64_t tiles_total = 0;
    for (const launch_t &ln : launches) tiles_total = std::max(tiles_total, ln.first + ln.count);
    void *d_redo = nullptr;
    void *d_redo_list = nullptr;
    if (use_memo && sl.table_global) {
        if (int rc = ws_get(c, 7, (size_t)tiles_total * sizeof(int32_t), &d_redo)) return rc;
        if (int rc = ws_get(c, 11, (size_t)tiles_total * sizeof(int32_t), &d_redo_list)) return rc;
    }
    if (use_memo) {
        // the table is rebuilt on every call: part of the timed work, never reused across calls.  The
        // same launch zeroes the redo flags and resets d_flags[8..15]: the largest missed pair
        // (-1, -1) and the (count, cursor) pairs of the second pass, one per workgroup size
        fptk::launch_nb_memo(c->stream, sl.model, n_dm, c->memo_exp, c->memo_obs, d_memo, (int32_t *)d_redo,
                             d_redo ? tiles_total : 0, d_miss, d_memo2 ? c->d_flags + 16 : nullptr, memo2_state,
                             c->memo2_rows, c->memo2_stride);
        if (int rc = launch_ok("k_nb_memo")) return rc;
    }
    // pass 0 (memo mode only): memo-only instance over every tile; pass 1: full instance (over
    // the flagged tiles in memo mode, over everything in direct mode)
    for (int pass = d_redo ? 0 : 1; pass < 2; ++pass) {
        const bool memo_only = pass == 0;
        const bool main_pass = memo_only || !d_redo;
        if (rec && main_pass) HIP_TRY(hipEventRecord(c->tev[c->tev_used + 1], c->stream));
        // first pass of memo mode: the lean kernel where it applies (the `detect` window widths)
        bool lean_pass = false;
        if (memo_only && c->use_lean) {
            fptk::scan_launch s2 = sl;
            s2.redo = (int32_t *)d_redo;
            // the lean kernel addresses its outputs with 32-bit byte offsets inside an interval
            const int longest = d->interval_off ? c->plan_max_len : d->interval_len;
            lean_pass = fptk::scan_lean_applies(s2) && longest < (1 << 29);
        }
        for (const launch_t &ln : lean_pass ? lean_launches : launches) {
            fptk::scan_launch s2 = sl;
            s2.tile_len = ln.tile_len;
            s2.nc_max = (ln.nt + 2 * pad + 1 + 63) & ~63;  // whole 64-position tiles
            s2.redo = (int32_t *)d_redo;
            s2.redo_list = (int32_t *)d_redo_list;
            // count and cursor of the second pass: a pair of d_flags[10..15] per workgroup size
            s2.redo_cursor = c->d_flags + 10 + 2 * (ln.nt <= 256 ? 0 : (ln.nt <= 512 ? 1 : 2));
            const bool lean = lean_pass;
            size_t lds = lean ? fptk::scan_lean_lds_bytes(ln.nt) : fptk::scan_lds_bytes(s2.nc_max, s2.table_global != 0, memo_only);
            if (lds > 160 * 1024)
                return fail(FPT_ERR_INVALID, "window padding too large for LDS (%zu bytes needed)", lds);
            if (lean) HIP_TRY(fptk::scan_lean_set_lds(ln.nt));
            else HIP_TRY(fptk::scan_set_lds(ln.nt, hw, shw, s2.table_global != 0, memo_only, !memo_only && d_redo, lds));
            for (int64_t done = 0; done < ln.count; done += 0x7fffff00) {
                int64_t n = std::min<int64_t>(ln.count - done, 0x7fffff00);
                s2.tile_first = ln.first + done;
                s2.redo_cursor_clear = done > 0 ? 1 : 0;  // a second chunk of the same size class reuses the pair
                if (lean) fptk::launch_scan_lean(c->stream, ln.nt, (int)n, s2);
                else fptk::launch_scan(c->stream, ln.nt, (int)n, lds, s2, memo_only);
                if (int rc = launch_ok("k_scan_fused")) return rc;
            }
        }
        if (rec && main_pass) HIP_TRY(hipEventRecord(c->tev[c->tev_used + 2], c->stream));
        if (memo_only && d_memo2) {  // between the passes: the table for what the first pass missed
            fptk::launch_nb_memo2(c->stream, sl.model, n_dm, d_miss, c->d_flags + 16, c->memo_exp, c->memo_obs,
                                  c->memo2_rows, c->memo2_stride, d_memo2);
            if (int rc = launch_ok("k_nb_memo2")) return rc;
            c->m2_extended = true;
        }
    }
    HIP_TRY(hipEventRecord(rec ? c->tev[c->tev_used + 3] : c->ev1, c->stream));
    c->last_tiles = tiles_total;
    c->last_has_redo = d_redo != nullptr;
    if (rec) c->tev_used += 4;
    else c->timed = true;
    return FPT_OK;
}

int fpt_scan_stats(fpt_ctx *c, int64_t *tiles_out, int64_t *redone_out, int32_t *miss_max_out) {
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    int64_t redone = 0;
    if (c->last_has_redo && c->last_tiles > 0 && c->ws[7]) {
        std::vector<int32_t> flags((size_t)c->last_tiles);
        HIP_TRY(hipMemcpy(flags.data(), c->ws[7], flags.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int32_t f : flags) redone += f != 0;
    }
    if (tiles_out) *tiles_out = c->last_tiles;
    if (redone_out) *redone_out = redone;
    if (miss_max_out) HIP_TRY(hipMemcpy(miss_max_out, c->d_flags + 8, 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return FPT_OK;
}

int fpt_timing_enable(fpt_ctx *c, int max_records) {
    if (int rc = check_ctx(c)) return rc;
    if (max_records < 0 || max_records > 100000) return fail(FPT_ERR_INVALID, "bad record count");
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (hipEvent_t e : c->tev) (void)hipEventDestroy(e);
    c->tev.clear();
    c->tev_used = 0;
    for (int i = 0; i < 4 * max_records; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->tev.push_back(e);
    }
    return FPT_OK;
}

int fpt_timing_read(fpt_ctx *c, float *ms_out, int cap, int *n_out) {
    if (int rc = check_ctx(c)) return rc;
    if (!n_out || (cap > 0 && !ms_out)) return fail(FPT_ERR_INVALID, "null output");
    HIP_TRY(hipStreamSynchronize(c->stream));
    int n = c->tev_used / 4;
    *n_out = n;
    for (int i = 0; i < n && i < cap; ++i) {
        HIP_TRY(hipEventElapsedTime(&ms_out[2 * i], c->tev[4 * i], c->tev[4 * i + 3]));
        HIP_TRY(hipEventElapsedTime(&ms_out[2 * i + 1], c->tev[4 * i + 1], c->tev[4 * i + 2]));
    }
    c->tev_used = 0;
    return FPT_OK;
}

int fpt_mark(fpt_ctx *c, int32_t *id_out) {
    if (int rc = check_ctx(c)) return rc;
    if (!id_out) return fail(FPT_ERR_INVALID, "null output");
    if (c->marks.size() >= (size_t)1 << 20) return fail(FPT_ERR_INVALID, "too many marks: fpt_marks_clear");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    c->marks.push_back(e);
    HIP_TRY(hipEventRecord(e, c->stream));
    *id_out = (int32_t)c->marks.size() - 1;
    return FPT_OK;
}

int fpt_mark_elapsed(fpt_ctx *c, int32_t from, int32_t to, float *ms_out) {
    if (int rc = check_ctx(c)) return rc;
    if (!ms_out) return fail(FPT_ERR_INVALID, "null output");
    if (from < 0 || to < 0 || (size_t)from >= c->marks.size() || (size_t)to >= c->marks.size())
        return fail(FPT_ERR_INVALID, "no such mark");
    HIP_TRY(hipEventSynchronize(c->marks[(size_t)to]));
    HIP_TRY(hipEventElapsedTime(ms_out, c->marks[(size_t)from], c->marks[(size_t)to]));
    return FPT_OK;
}

int fpt_marks_clear(fpt_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (hipEvent_t e : c->marks) (void)hipEventDestroy(e);
    c->marks.clear();
    return FPT_OK;
}

int fpt_fdr_dev(fpt_ctx *c, const fpt_fdr_desc *d) {
    if (int rc = check_ctx(c)) return rc;
    if (!d) return fail(FPT_ERR_INVALID, "null descriptor");
    const int n_dm = d->dm_ids ? d->n_dm : 1;
    if (n_dm < 1 || d->dm_id < 0 || d->dm_id + n_dm > FPT_MAX_DISPERSION_MODELS)
        return fail(FPT_ERR_INVALID, "dispersion model slots [%d, %d) out of range", d->dm_id, d->dm_id + n_dm);
    for (int i = 0; i < n_dm; ++i)
        if (!c->have_model[d->dm_id + i])
            return fail(FPT_ERR_INVALID, "dispersion model %d not set", d->dm_id + i);
    if (d->n_intervals < 0) return fail(FPT_ERR_INVALID, "negative interval count");
    if (d->n_intervals == 0) return FPT_OK;
    if (d->times < 1 || d->times > 1000000) return fail(FPT_ERR_INVALID, "times %d out of range", d->times);
    if (d->half_win_width < 0 || d->half_win_width > 200)
        return fail(FPT_ERR_INVALID, "half window %d out of range", d->half_win_width);
    if (!d->exp || !d->winp || !d->efdr_out) return fail(FPT_ERR_INVALID, "null track");
    // Intervals of up to kLdsMax bases keep their buffers in LDS; longer ones (up to kLongMax)
    // run the same kernel over buffers in global memory.
    constexpr int kLdsMax = 2048, kLongMax = 1 << 22;  // five n2-sized double buffers + three int ones must fit 160 KB
    int lmax = d->interval_len;
    // Ragged batches are binned by the power of two that holds the interval: the kernel's LDS
    // buffers are sized by the largest interval of a launch, so one 2,000-base interval in a
    // launch of 150-base ones would leave a single workgroup per compute unit (measured on the
    // whole-genome shape: 398 -> 69 ms per 7.1e7 bases)
    // and by the workgroup size that leaves the fewest lanes idle in the draw loop (one lane per
    // base and null track, intervals longer than the workgroup in several passes)
    constexpr int kClasses = 8;
    static const int cls_len[kClasses] = {64, 128, 192, 256, 384, 512, 1024, 2048};  // longest interval (kLdsMax: the last)
    static const int cls_n2[kClasses] = {64, 128, 256, 256, 512, 512, 1024, 2048};   // LDS buffers
    static const int cls_nt[kClasses] = {64, 128, 192, 256, 192, 256, 512, 512};      // lanes (512: measured slower for the 385..512 class, 2 x 8 wavefronts per CU against 3 x 4)
    auto cls_of = [&](int L) {
        int k = 0;
        while (cls_len[k] < L) ++k;
        return k;
    };
    // ragged batches: the intervals of every size class (the last: the long ones) listed back to
    // back in a pinned buffer the context keeps -- counted, then placed: no growing vectors, one
    // copy to the device that needs no wait
    int64_t cls_n[kClasses + 1] = {}, cls_at[kClasses + 2] = {};
    // slices (ragged batches, the `detect` width, split + light launches): the draws of an interval of more than
    // 256 bases by several three-wavefront workgroups (k_fdr_slice); lists and offsets ride in the pinned buffer
    // behind the interval lists: goff (int64 per interval), slice_iv and slice_start (int32 per slice, class-major)
    // (per call the slices cost two more launches per size class, a list and a memset -- 0.15 ms per 100,000
    // intervals -- and win 0.0055 ms per draw: from 32 draws per base on, or when asked for by FPT_FDR_SLICES=2)
    const bool use_slices = d->interval_off && d->half_win_width == 3 && c->fdr_split && c->fdr_light && c->fdr_slices &&
                            (d->times >= 32 || c->fdr_slices_always);
    int64_t sl_n[kClasses + 1] = {}, sl_at[kClasses + 2] = {}, ghist_total = 0;
    size_t pin_goff = 0, pin_slice_iv = 0, pin_slice_start = 0, pin_bytes = 0;
    int64_t total_host = -1;  // ragged batch: the track length, from whichever copy of the offsets was read
    if (d->interval_off) {
        if (d->n_intervals > 0x7fffff00) return fail(FPT_ERR_INVALID, "too many intervals");
        // interval lengths: from the caller's host copy of the offsets, else back from the device
        const int64_t *off = d->interval_off_host;
        std::vector<int64_t> off_back;
        if (!off) {
            off_back.resize((size_t)d->n_intervals + 1);
            HIP_TRY(hipMemcpyAsync(off_back.data(), d->interval_off, off_back.size() * 8, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            off = off_back.data();
        }
        lmax = 0;
        total_host = off[d->n_intervals];
        std::vector<uint8_t> cls((size_t)d->n_intervals);
        for (int64_t i = 0; i < d->n_intervals; ++i) {
            const int64_t L = off[i + 1] - off[i];
            if (L < 0) return fail(FPT_ERR_INVALID, "bad interval offsets");
            if (L > kLongMax)
                return fail(FPT_ERR_INVALID, "interval of %lld bases: fpt_fdr_dev handles at most %d", (long long)L, kLongMax);
            if (L > lmax) lmax = (int)L;
            const int k = L > kLdsMax ? kClasses : cls_of((int)L);
            cls[(size_t)i] = (uint8_t)k;
            cls_n[k] += 1;
            if (use_slices && k < kClasses && L > 256) sl_n[k] += fptk::fdr_slices_of((int)L, false);
        }
        for (int k = 0; k <= kClasses; ++k) cls_at[k + 1] = cls_at[k] + cls_n[k];
        for (int k = 0; k <= kClasses; ++k) sl_at[k + 1] = sl_at[k] + sl_n[k];
        pin_goff = ((size_t)d->n_intervals * sizeof(int32_t) + 7) & ~(size_t)7;
        pin_slice_iv = pin_goff + (use_slices && sl_at[kClasses] ? (size_t)d->n_intervals * sizeof(int64_t) : 0);
        pin_slice_start = pin_slice_iv + (size_t)sl_at[kClasses] * sizeof(int32_t);
        pin_bytes = pin_slice_start + (size_t)sl_at[kClasses] * sizeof(int32_t);
        const size_t need = pin_bytes;
        if (c->pin_list_busy) {  // the copy of the call before has long happened; make sure
            HIP_TRY(hipEventSynchronize(c->pin_list_copied));
            c->pin_list_busy = false;
        }
        if (c->pin_list_bytes < need) {
            if (c->pin_list) (void)hipHostFree(c->pin_list);
            c->pin_list = nullptr;
            c->pin_list_bytes = 0;
            const size_t want = need + need / 4;
            if (hipHostMalloc((void **)&c->pin_list, want, hipHostMallocDefault) != hipSuccess)
                return fail(FPT_ERR_NOMEM, "no pinned host memory for %zu bytes of interval lists", want);
            c->pin_list_bytes = want;
        }
        int64_t cur[kClasses + 1];
        for (int k = 0; k <= kClasses; ++k) cur[k] = cls_at[k];
        for (int64_t i = 0; i < d->n_intervals; ++i) c->pin_list[cur[cls[(size_t)i]]++] = (int32_t)i;
        if (use_slices && sl_at[kClasses]) {
            int64_t *goff = (int64_t *)((char *)c->pin_list + pin_goff);
            int32_t *s_iv = (int32_t *)((char *)c->pin_list + pin_slice_iv), *s_start = (int32_t *)((char *)c->pin_list + pin_slice_start);
            int64_t scur[kClasses + 1];
            for (int k = 0; k <= kClasses; ++k) scur[k] = sl_at[k];
            for (int64_t i = 0; i < d->n_intervals; ++i) {
                const int64_t L = off[i + 1] - off[i];
                const int k = cls[(size_t)i];
                if (k >= kClasses || L <= 256) {
                    goff[i] = -1;
                    continue;
                }
                goff[i] = ghist_total | ((L + 2) << 40);  // (offset, room)
                ghist_total += L + 2;
                const int slice_len = fptk::fdr_slice_positions_of((int)L, false);
                for (int64_t st0 = 0; st0 < L; st0 += slice_len) {
                    s_iv[scur[k]] = (int32_t)i;
                    s_start[scur[k]++] = (int32_t)st0;
                }
            }
        }
    } else if (lmax <= 0) {
        return fail(FPT_ERR_INVALID, "interval_len must be positive");
    } else if (lmax > kLongMax) {
        return fail(FPT_ERR_INVALID, "interval of %d bases: fpt_fdr_dev handles at most %d", lmax, kLongMax);
    }
    auto pow2 = [](int n) { int p = 64; while (p < n) p <<= 1; return p; };
    const int64_t memo_n = (int64_t)c->memo_exp * c->fdr_memo_obs;
    void *d_memo;
    if (int rc = ws_get(c, 8, (size_t)memo_n * 16 * n_dm, &d_memo)) return rc;
    const double *model = c->d_models + (size_t)d->dm_id * kModelDoubles;
    fptk::launch_nb_memo(c->stream, model, n_dm, c->memo_exp, c->fdr_memo_obs, d_memo);
    if (int rc = launch_ok("k_nb_memo")) return rc;
    void *d_alias;
    if (int rc = ws_get(c, 6, fptk::nb_alias_bytes(n_dm, c->memo_exp, c->fdr_memo_obs), &d_alias)) return rc;
    fptk::launch_nb_alias(c->stream, d_memo, n_dm, c->memo_exp, c->fdr_memo_obs, d_alias);
    if (int rc = launch_ok("k_nb_alias")) return rc;
    fptk::fdr_launch fl{};
    fl.n_intervals = d->n_intervals;
    fl.interval_len = d->interval_off ? 0 : d->interval_len;
    fl.interval_off = d->interval_off;
    fl.base_index0 = d->base_index0;
    fl.hw = d->half_win_width;
    fl.times = d->times;
    fl.seed = d->seed;
    fl.model = model;
    fl.memo = d_memo;
    fl.alias = d_alias;
    fl.n_models = n_dm;
    // (the light draw instances carry no test hooks -- caller's uniforms, null p-values out: a call with either takes
    // the full instance, one workgroup per interval)
    fl.light = c->fdr_light && !d->null_uniform && !d->null_winp_out;
    fl.light_dbuf = c->fdr_light_dbuf;
    fl.memo_exp = c->memo_exp;
    fl.memo_obs = c->fdr_memo_obs;
    fl.exp = d->exp;
    fl.winp = d->winp;
    fl.obs = d->obs;
    fl.efdr = d->efdr_out;
    fl.null_uniform = d->null_uniform;
    fl.null_out = d->null_winp_out;
    fl.dm_ids = d->dm_ids;
#ifdef FPT_ABLATE
    if (const char *e = getenv("FPT_ABLATE")) fl.ablate = atoi(e);
#endif
    // the hand-over between the set-up launch and the draw launch (the `detect` width; FPT_FDR_SPLIT=0: one launch)
    if (d->half_win_width == 3 && c->fdr_split) {
        const int64_t total = d->interval_off ? total_host : d->n_intervals * (int64_t)d->interval_len;
        if (total > 0) {
            void *ws;
            const size_t key_b = ((size_t)total * 8 + 255) & ~(size_t)255, idx_b = ((size_t)total * 2 + 255) & ~(size_t)255;
            if (int rc = ws_get(c, 14, key_b + idx_b + (size_t)d->n_intervals * 12, &ws)) return rc;
            fl.ws_key = (double *)ws;
            fl.ws_idx = (uint16_t *)((char *)ws + key_b);
            fl.ws_misc = (int32_t *)((char *)ws + key_b + idx_b);
            fl.ws_total = total;
        }
    }
    // long intervals: per-workgroup buffers in a global workspace of at most 1 GiB
    auto launch_long = [&](const int32_t *list, int64_t n_list) -> int {
        fl.n2_max = pow2(lmax);
        fl.gws_stride = (int64_t)((fptk::fdr_lds_bytes(fl.n2_max, false, true) + 255) & ~(size_t)255);
        const int64_t n_blocks = list ? n_list : d->n_intervals;
        fl.gws_blocks = std::max<int64_t>(1, std::min<int64_t>(n_blocks, ((int64_t)1 << 30) / fl.gws_stride));
        if (int rc = ws_get(c, 4, (size_t)(fl.gws_stride * fl.gws_blocks), &fl.gws)) return rc;
        fl.iv_list = list;
        fl.n_list = n_list;
        HIP_TRY(fptk::launch_fdr(c->stream, fl));
        return launch_ok("k_fdr_null (global buffers)");
    };
    if (!d->interval_off) {  // uniform batch
        if (lmax > kLdsMax) return launch_long(nullptr, 0);
        // intervals of more than 256 bases: the light draws as slices (see the ragged batches above), interval-major
        if (lmax > 256 && fl.ws_key && d->half_win_width == 3 && c->fdr_light && c->fdr_slices &&
            (d->times >= 32 || c->fdr_slices_always)) {
            void *d_counts;
            const size_t count_bytes = (size_t)d->n_intervals * ((size_t)lmax + 3) * sizeof(int32_t);
            if (int rc = ws_get(c, 15, count_bytes, &d_counts)) return rc;
            HIP_TRY(hipMemsetAsync(d_counts, 0, count_bytes, c->stream));
            fl.ghist = (int32_t *)d_counts;
            fl.gnan = (int32_t *)d_counts + (size_t)d->n_intervals * ((size_t)lmax + 2);
            fl.n_slices = d->n_intervals * (int64_t)fptk::fdr_slices_of(lmax, true);
        }
        fl.n2_max = pow2(lmax);
        fl.nt = cls_nt[cls_of(lmax)];
        fl.max_len = lmax;
        HIP_TRY(fptk::launch_fdr(c->stream, fl));
        return launch_ok("k_fdr_null");
    }
    // ragged batch: the lists to the device in one copy (pinned: it is queued, nobody waits)
    void *d_list;
    if (int rc = ws_get(c, 5, pin_bytes, &d_list)) return rc;
    HIP_TRY(hipMemcpyAsync(d_list, c->pin_list, pin_bytes, hipMemcpyHostToDevice, c->stream));
    if (use_slices && sl_at[kClasses] && fl.ws_key) {
        void *d_counts;  // the sliced intervals' counts (L + 2 each), then a count of NaN windows per interval: zeroed
        const size_t count_bytes = ((size_t)ghist_total + (size_t)d->n_intervals) * sizeof(int32_t);
        if (int rc = ws_get(c, 15, count_bytes, &d_counts)) return rc;
        HIP_TRY(hipMemsetAsync(d_counts, 0, count_bytes, c->stream));
        fl.goff = (const int64_t *)((const char *)d_list + pin_goff);
        fl.ghist = (int32_t *)d_counts;
        fl.gnan = (int32_t *)d_counts + ghist_total;
    }
    if (!c->pin_list_copied) HIP_TRY(hipEventCreateWithFlags(&c->pin_list_copied, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->pin_list_copied, c->stream));
    c->pin_list_busy = true;
    const int32_t *lists = (const int32_t *)d_list;
    for (int k = 0; k < kClasses; ++k) {
        if (cls_n[k] == 0) continue;
        fl.n2_max = cls_n2[k];
        fl.nt = cls_nt[k];
        fl.max_len = cls_len[k];
        fl.iv_list = lists + cls_at[k];
        fl.n_list = cls_n[k];
        fl.n_slices = fl.ghist ? sl_n[k] : 0;
        fl.slice_iv = (const int32_t *)((const char *)d_list + pin_slice_iv) + sl_at[k];
        fl.slice_start = (const int32_t *)((const char *)d_list + pin_slice_start) + sl_at[k];
        HIP_TRY(fptk::launch_fdr(c->stream, fl));
        if (int rc = launch_ok("k_fdr_null")) return rc;
    }
    fl.n_slices = 0;
    if (cls_n[kClasses]) return launch_long(lists + cls_at[kClasses], cls_n[kClasses]);
    return FPT_OK;
}

int fpt_set_memo_dims(fpt_ctx *c, int memo_exp, int memo_obs) {
    if (int rc = check_ctx(c)) return rc;
    if (memo_exp < 1 || memo_exp > 4096 || memo_obs < 1 || memo_obs > 4096)
        return fail(FPT_ERR_INVALID, "memo dims out of range");
    c->memo_exp = memo_exp;
    c->memo_obs = memo_obs;
    c->fdr_memo_obs = memo_obs;
    return FPT_OK;
}

int fpt_drop_kept_tables(fpt_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    c->m2_dm_id = -1;  // the next scan call finds no table of its models and starts an empty one
    c->m2_extended = false;
    return FPT_OK;
}

int fpt_last_scan_ms(fpt_ctx *c, float *ms_out) {
    if (int rc = check_ctx(c)) return rc;
    if (!ms_out) return fail(FPT_ERR_INVALID, "null output");
    if (!c->timed) return fail(FPT_ERR_INVALID, "no scan has been launched");
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms_out, c->ev0, c->ev1));
    return FPT_OK;
}

int fpt_posterior_dev(fpt_ctx *c, const fpt_posterior_desc *d) {
    if (int rc = check_ctx(c)) return rc;
    if (!d) return fail(FPT_ERR_INVALID, "null descriptor");
    if (d->n_intervals < 0) return fail(FPT_ERR_INVALID, "negative interval count");
    if (d->n_datasets < 1 || d->n_datasets > (1 << 20)) return fail(FPT_ERR_INVALID, "n_datasets %d out of range", d->n_datasets);
    if (!d->models) {  // the models of the datasets sit in consecutive slots
        if (d->dm_id < 0 || d->dm_id + d->n_datasets > FPT_MAX_DISPERSION_MODELS)
            return fail(FPT_ERR_INVALID, "dispersion model slots [%d, %d) out of range (more than %d datasets: pass `models`)",
                        d->dm_id, d->dm_id + d->n_datasets, FPT_MAX_DISPERSION_MODELS);
        for (int i = 0; i < d->n_datasets; ++i)
            if (!c->have_model[d->dm_id + i]) return fail(FPT_ERR_INVALID, "dispersion model %d not set", d->dm_id + i);
    }
    if (d->half_win_width < 0 || d->half_win_width > 32)
        return fail(FPT_ERR_INVALID, "half window %d out of range", d->half_win_width);
    if (d->n_intervals == 0 || d->total_bases == 0) return FPT_OK;
    if (d->total_bases < 0) return fail(FPT_ERR_INVALID, "negative track length");
    if (!d->betas || !d->obs || !d->exp || !d->fdr || !d->w || !d->post_out) return fail(FPT_ERR_INVALID, "null buffer");
    if (!d->interval_off) {
        if (d->interval_len <= 0) return fail(FPT_ERR_INVALID, "interval_len must be positive");
        if (d->n_intervals * (int64_t)d->interval_len != d->total_bases)
            return fail(FPT_ERR_INVALID, "total_bases does not match n_intervals x interval_len");
    }
    // the Beta priors travel through a small device buffer (slot 6: shared with the FDR pass's
    // lists, both are consumed by the launch that follows on the same stream)
    // ... and so do models handed over with the call (any number of datasets), behind the priors
    void *d_beta;
    const size_t n_beta = (size_t)d->n_datasets * 2, n_par = d->models ? (size_t)d->n_datasets * kModelDoubles : 0;
    const size_t nb = (n_beta + n_par) * sizeof(double);
    if (int rc = ws_get(c, 6, nb, &d_beta)) return rc;
    c->beta_host.assign(d->betas, d->betas + n_beta);  // stays alive for the async copy
    if (d->models) c->beta_host.insert(c->beta_host.end(), d->models, d->models + n_par);
    HIP_TRY(hipMemcpyAsync(d_beta, c->beta_host.data(), nb, hipMemcpyHostToDevice, c->stream));
    fptk::posterior_launch pl{};
    pl.n_intervals = d->n_intervals;
    pl.interval_len = d->interval_off ? 0 : d->interval_len;
    pl.interval_off = d->interval_off;
    pl.total_bases = d->total_bases;
    pl.max_len = d->interval_off ? (d->max_interval_len > 0 ? d->max_interval_len : 2048) : d->interval_len;
    pl.n_datasets = d->n_datasets;
    pl.hw = d->half_win_width;
    pl.cutoff = d->fdr_cutoff;
    pl.pseudocount = d->pseudocount;
    pl.obs = d->obs;
    pl.exp = d->exp;
    pl.fdr = d->fdr;
    pl.w = d->w;
    pl.models = d->models ? (const double *)d_beta + n_beta : c->d_models + (size_t)d->dm_id * kModelDoubles;
    pl.betas = (const double *)d_beta;
    pl.post_out = d->post_out;
    pl.prior_out = d->prior_out;
    pl.delta_out = d->delta_out;
    pl.ll_on_out = d->ll_on_out;
    pl.ll_off_out = d->ll_off_out;
    pl.status_out = d->status_out;
    // the kernel instance without the reference's sums over the segments where every model allows it (the usual case)
    pl.all_simple = true;
    for (int i = 0; i < d->n_datasets && pl.all_simple; ++i)
        pl.all_simple = fptk::posterior_model_simple(d->models ? d->models + (size_t)i * kModelDoubles : c->h_models[d->dm_id + i]);
    // tables of the unoccupied log-pmf and of lgam(k + 1), rebuilt by every call (slot 13)
    // -- where they pay and fit: 65,536 evaluations and 512 KiB per dataset, so a batch of fewer than 16,384
    // bases (the tables would cost more evaluations than they save), more than 4,096 datasets (2 GiB) or a
    // failed allocation takes the direct path -- the same records (test_posterior_batch_fuzz)
    void *d_tab = nullptr;
    if (!c->posterior_direct && d->total_bases >= 16384 && d->n_datasets <= 4096) {
        if (ws_get(c, 13, fptk::posterior_table_bytes(d->n_datasets), &d_tab) == FPT_OK) {
            pl.off_table = (double *)d_tab;
            pl.lgam_table = pl.off_table + (size_t)d->n_datasets * 256 * 256;
        }
    }
    pl.max_len_unknown = d->interval_off && d->max_interval_len <= 0;
    if (d->interval_off) {  // the list of the long intervals' further chunks (slot 14, shared with the FDR pass's workspace)
        void *d_plan;
        if (int rc = ws_get(c, 14, fptk::posterior_plan_bytes(d->total_bases, d->half_win_width), &d_plan)) return rc;
        pl.plan_ws = d_plan;
    }
    HIP_TRY(fptk::launch_posterior(c->stream, pl));
    return launch_ok("k_posterior");
}

int fpt_detect_columns_dev(fpt_ctx *c, int64_t n_intervals, int32_t interval_len, const int64_t *interval_off_dev,
                           int64_t total_bases, const int32_t *status_dev, const double *exp_dev, const double *obs_dev,
                           const double *pval_dev, const double *winp_dev, const double *efdr_dev, double *out_dev) {
    if (int rc = check_ctx(c)) return rc;
    if (n_intervals < 0 || total_bases < 0) return fail(FPT_ERR_INVALID, "negative size");
    if (total_bases == 0) return FPT_OK;
    if (!exp_dev || !obs_dev || !pval_dev || !winp_dev || !efdr_dev || !out_dev) return fail(FPT_ERR_INVALID, "null track");
    if (status_dev && !interval_off_dev && interval_len <= 0) return fail(FPT_ERR_INVALID, "interval_len must be positive");
    fptk::launch_detect_columns(c->stream, n_intervals, interval_len, interval_off_dev, status_dev, exp_dev, obs_dev,
                                pval_dev, winp_dev, efdr_dev, total_bases, out_dev);
    return launch_ok("k_detect_columns");
}

int fpt_hist2d_dev(fpt_ctx *c, const double *exp_dev, const double *obs_dev, int64_t n, int rows, int cols,
                   uint64_t *hist_dev) {
    if (int rc = check_ctx(c)) return rc;
    if (n < 0 || rows < 1 || cols < 1 || (int64_t)rows * cols > (1 << 28))
        return fail(FPT_ERR_INVALID, "bad histogram shape or length");
    if (!hist_dev || ((!exp_dev || !obs_dev) && n > 0)) return fail(FPT_ERR_INVALID, "null buffer");
    fptk::launch_hist2d(c->stream, exp_dev, obs_dev, n, rows, cols, (unsigned long long *)hist_dev);
    return launch_ok("k_hist2d");
}

static int segment_setup(fpt_ctx *c, const fpt_segment_desc *d, fptk::segment_launch *sl) {
    if (int rc = check_ctx(c)) return rc;
    if (!d) return fail(FPT_ERR_INVALID, "null descriptor");
    if (d->n_intervals < 0 || d->n_intervals > 0x7fffff00) return fail(FPT_ERR_INVALID, "bad interval count");
    if (!d->track && d->n_intervals > 0) return fail(FPT_ERR_INVALID, "null track");
    if (!d->interval_off && d->interval_len <= 0) return fail(FPT_ERR_INVALID, "interval_len must be positive");
    if (d->w < 1 || d->w > (1 << 20)) return fail(FPT_ERR_INVALID, "w %d out of range", d->w);
    sl->n_intervals = d->n_intervals;
    sl->interval_len = d->interval_off ? 0 : d->interval_len;
    sl->interval_off = d->interval_off;
    sl->track = d->track;
    sl->threshold = d->threshold;
    sl->w = d->w;
    sl->decreasing = d->decreasing ? 1 : 0;
    return FPT_OK;
}

int fpt_segment_count_dev(fpt_ctx *c, const fpt_segment_desc *d, int64_t *total_out) {
    fptk::segment_launch sl{};
    if (int rc = segment_setup(c, d, &sl)) return rc;
    if (!total_out) return fail(FPT_ERR_INVALID, "null output");
    c->seg_n_intervals = -1;
    *total_out = 0;
    const size_t n = (size_t)d->n_intervals;
    void *d_counts, *d_offsets;
    if (int rc = ws_get(c, 10, std::max<size_t>(n, 1) * sizeof(int32_t), &d_counts)) return rc;
    if (int rc = ws_get(c, 11, (n + 1) * sizeof(int64_t), &d_offsets)) return rc;
    std::vector<int32_t> counts(n);
    std::vector<int64_t> offsets(n + 1, 0);
    if (n) {
        sl.counts = (int32_t *)d_counts;
        fptk::launch_segment(c->stream, sl, false);
        if (int rc = launch_ok("k_segment (count)")) return rc;
        HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + counts[i];
    }
    HIP_TRY(hipMemcpyAsync(d_offsets, offsets.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // `offsets` is pageable host memory
    c->seg_n_intervals = d->n_intervals;
    c->seg_total = offsets[n];
    c->seg_track = d->track;
    *total_out = offsets[n];
    return FPT_OK;
}

int fpt_segment_fill_dev(fpt_ctx *c, const fpt_segment_desc *d, int64_t capacity, int32_t *seg_interval,
                         int32_t *seg_start, int32_t *seg_end, double *seg_score) {
    fptk::segment_launch sl{};
    if (int rc = segment_setup(c, d, &sl)) return rc;
    if (c->seg_n_intervals != d->n_intervals || c->seg_track != d->track)
        return fail(FPT_ERR_INVALID, "fpt_segment_fill_dev must follow fpt_segment_count_dev of the same batch");
    if (capacity < c->seg_total)
        return fail(FPT_ERR_INVALID, "capacity %lld below the %lld segments counted", (long long)capacity,
                    (long long)c->seg_total);
    if (c->seg_total == 0) return FPT_OK;
    if (!seg_interval || !seg_start || !seg_end || !seg_score) return fail(FPT_ERR_INVALID, "null output");
    sl.offsets = (const int64_t *)c->ws[11];
    sl.seg_iv = seg_interval;
    sl.seg_start = seg_start;
    sl.seg_end = seg_end;
    sl.seg_score = seg_score;
    fptk::launch_segment(c->stream, sl, true);
    return launch_ok("k_segment (fill)");
}

int fpt_synth_dev(fpt_ctx *c, uint64_t seed, int64_t pos0_counts, int64_t n_counts, double *cp,
                  double *cm, int64_t pos0_seq, int64_t n_seq, uint8_t *seq) {
    if (int rc = check_ctx(c)) return rc;
    if (n_counts < 0 || n_seq < 0) return fail(FPT_ERR_INVALID, "negative length");
    fptk::launch_synth(c->stream, seed, pos0_counts, n_counts, cp, cm, pos0_seq, n_seq, seq);
    return launch_ok("k_synth");
}

int fpt_synth_hotspots_dev(fpt_ctx *c, uint64_t seed, int64_t pos0_counts, int64_t n_counts, int32_t padded_len,
                           int32_t per_mille, double *cp, double *cm) {
    if (int rc = check_ctx(c)) return rc;
    if (n_counts < 0 || padded_len < 1 || per_mille < 0 || per_mille > 1000)
        return fail(FPT_ERR_INVALID, "bad hotspot parameters");
    fptk::launch_synth_hotspots(c->stream, seed, pos0_counts, n_counts, padded_len, per_mille, cp, cm);
    return launch_ok("k_synth_hotspots");
}

int fpt_checksum_dev(fpt_ctx *c, const double *dev, int64_t n, uint64_t *host_out) {
    if (int rc = check_ctx(c)) return rc;
    if (!host_out || n < 0 || (!dev && n > 0)) return fail(FPT_ERR_INVALID, "bad arguments");
    HIP_TRY(hipMemsetAsync(c->d_sum, 0, sizeof(unsigned long long), c->stream));
    fptk::launch_checksum(c->stream, dev, n, c->d_sum);
    if (int rc = launch_ok("k_checksum")) return rc;
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, c->d_sum, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *host_out = (uint64_t)v;
    return FPT_OK;
}

int fpt_dev_alloc(fpt_ctx *c, int64_t bytes, void **dev_out) {
    if (int rc = check_ctx(c)) return rc;
    if (!dev_out || bytes < 0) return fail(FPT_ERR_INVALID, "bad arguments");
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes > 0 ? (size_t)bytes : 16);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(FPT_ERR_NOMEM, "device allocation of %lld bytes failed", (long long)bytes);
    }
    *dev_out = p;
    return FPT_OK;
}

int fpt_dev_free(fpt_ctx *c, void *dev) {
    if (int rc = check_ctx(c)) return rc;
    if (dev) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(dev));
    }
    return FPT_OK;
}

int fpt_dev_zero(fpt_ctx *c, void *dev, int64_t bytes) {
    if (int rc = check_ctx(c)) return rc;
    if (bytes < 0 || (!dev && bytes > 0)) return fail(FPT_ERR_INVALID, "bad arguments");
    if (bytes > 0) HIP_TRY(hipMemsetAsync(dev, 0, (size_t)bytes, c->stream));
    return FPT_OK;
}

int fpt_memcpy_h2d(fpt_ctx *c, void *dev, const void *host, int64_t bytes) {
    if (int rc = check_ctx(c)) return rc;
    if (bytes < 0 || ((!dev || !host) && bytes > 0)) return fail(FPT_ERR_INVALID, "bad arguments");
    if (bytes == 0) return FPT_OK;
    HIP_TRY(hipMemcpyAsync(dev, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FPT_OK;
}

int fpt_memcpy_d2h(fpt_ctx *c, void *host, const void *dev, int64_t bytes) {
    if (int rc = check_ctx(c)) return rc;
    if (bytes < 0 || ((!dev || !host) && bytes > 0)) return fail(FPT_ERR_INVALID, "bad arguments");
    if (bytes == 0) return FPT_OK;
    HIP_TRY(hipMemcpyAsync(host, dev, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FPT_OK;
}


/* ---- host arrays in, host arrays out: the per-call form a drop-in caller uses (modeling/predict.pyx:116-163 hands
 *      numpy arrays over and gets numpy arrays back), as a three-stage pipeline over chunks of intervals */
#pragma GCC visibility pop
}  // extern "C"

namespace {
bool is_pinned(const void *p) {  // memory the copy engines reach without the runtime's help (hipHostMalloc, hipHostRegister)
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}
}  // namespace

constexpr int kPipeSlots = 3;
struct fpt_host_pipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    struct slot_t {
        char *d_in = nullptr, *d_out = nullptr, *p_small = nullptr;   // device buffers; pinned staging of the offsets / model ids
        size_t d_in_bytes = 0, d_out_bytes = 0, p_small_bytes = 0;
        hipEvent_t ev_in = nullptr, ev_scan = nullptr;
        int64_t base0 = 0, bases = 0, iv0 = 0, n_iv = 0;             // the chunk the slot holds
    } slot[kPipeSlots];
    fpt_scan_host_stats last = {};
};

static void host_pipe_free(fpt_host_pipe *p) {
    if (!p) return;
    for (auto &s : p->slot) {
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.p_small) (void)hipHostFree(s.p_small);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_scan) (void)hipEventDestroy(s.ev_scan);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
}

namespace {
int pipe_grow(char **buf, size_t *have, size_t want, bool pinned) {
    if (*have >= want) return FPT_OK;
    if (*buf) {
        if (pinned) (void)hipHostFree(*buf);
        else (void)hipFree(*buf);
        *buf = nullptr;
        *have = 0;
    }
    want += want / 8;
    hipError_t e = pinned ? hipHostMalloc((void **)buf, want, hipHostMallocDefault) : hipMalloc((void **)buf, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(FPT_ERR_NOMEM, "%s allocation of %zu bytes failed", pinned ? "pinned host" : "device", want);
    }
    *have = want;
    return FPT_OK;
}
inline size_t up16(size_t n) { return (n + 15) & ~(size_t)15; }
}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int fpt_host_alloc(fpt_ctx *c, int64_t bytes, void **host_out) {
    if (int rc = check_ctx(c)) return rc;
    if (!host_out || bytes < 0) return fail(FPT_ERR_INVALID, "bad arguments");
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes > 0 ? (size_t)bytes : 16, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FPT_ERR_NOMEM, "pinned host allocation of %lld bytes failed", (long long)bytes);
    }
    *host_out = p;
    return FPT_OK;
}

int fpt_host_free(fpt_ctx *c, void *host) {
    if (int rc = check_ctx(c)) return rc;
    if (host) HIP_TRY(hipHostFree(host));
    return FPT_OK;
}

// A fresh host array costs a page fault per 4 KiB when it is first written -- by a device-to-host copy as much as by a
// loop: 1.6 GB of new numpy output arrive at 20 GB/s instead of 56 (tools/diag_pageable_copies.py).  A team of threads
// touches the pages first (one byte per page: the array holds nothing yet), with transparent huge pages asked for
// where the kernel grants them.
int fpt_host_prefault(void *host, int64_t bytes) {
    if (bytes < 0 || (!host && bytes > 0)) return fail(FPT_ERR_INVALID, "bad arguments");
    if (bytes < ((int64_t)1 << 22)) return FPT_OK;
    char *p = (char *)host;
    const uintptr_t page = 4096;
    char *a = (char *)(((uintptr_t)p + page - 1) & ~(page - 1)), *b = (char *)(((uintptr_t)p + (uintptr_t)bytes) & ~(page - 1));
    if (b <= a) return FPT_OK;
#ifdef MADV_HUGEPAGE
    (void)madvise(a, (size_t)(b - a), MADV_HUGEPAGE);
#endif
    const int nt = std::max(1, std::min(fpt_host_cpus(), 16));
    const size_t n_pages = (size_t)(b - a) / page, per = (n_pages + nt - 1) / nt;
    std::vector<std::thread> team;
    for (int t = 0; t < nt; ++t)
        team.emplace_back([=] {
            const size_t p0 = std::min(n_pages, per * t), p1 = std::min(n_pages, per * (t + 1));
            for (size_t i = p0; i < p1; ++i) ((volatile char *)a)[i * page] = 0;
        });
    for (auto &th : team) th.join();
    return FPT_OK;
}

int fpt_scan_host_last(fpt_ctx *c, fpt_scan_host_stats *out) {
    if (int rc = check_ctx(c)) return rc;
    if (!out) return fail(FPT_ERR_INVALID, "null output");
    if (!c->pipe) return fail(FPT_ERR_INVALID, "fpt_scan_host has not run on this context");
    *out = c->pipe->last;
    return FPT_OK;
}

// Three stages, three chunks in flight: the calling thread copies a chunk's inputs to the device (stream s_in) and
// launches its scan (the context's stream, behind an event); a second thread, started per call, copies every chunk's
// results back in order (stream s_out, behind the scan's event) and frees its slot.  The caller's arrays are handed to
// hipMemcpyAsync as they are: page-locked ones are read and written by the copy engines while the call returns at once;
// pageable ones are pinned by the runtime on the way and the call BLOCKS its thread until the copy is done -- at the
// link's rate all the same (56 GB/s one way, profiles/r06_pcie_micro.txt).  Two threads therefore keep both directions of
// the link busy whatever the arrays are; one thread with pageable arrays moved 9.6e8 bases/s on config 2's shape, staging
// through pinned buffers of the library's own by a team of host threads (the first form of this call) 1.15-1.45e9 by
// what else the host's memory system was doing, this form 1.49e9 with no host copy at all (gpurun_out/r06 host-array runs).
int fpt_scan_host(fpt_ctx *c, const fpt_scan_desc *d, int64_t chunk_bases) {
    if (int rc = check_ctx(c)) return rc;
    if (!d) return fail(FPT_ERR_INVALID, "null descriptor");
    if (d->n_intervals < 0) return fail(FPT_ERR_INVALID, "negative interval count");
    if (d->n_scales < 0 || d->n_scales > FPT_MAX_SCALES) return fail(FPT_ERR_INVALID, "bad n_scales");
    if (d->half_win_width < 0 || d->smoothing_half_win_width < 0) return fail(FPT_ERR_INVALID, "negative window width");
    if (!d->counts_plus || !d->counts_minus || !d->seq) return fail(FPT_ERR_INVALID, "null input");
    const int64_t n = d->n_intervals;
    if (n == 0) return FPT_OK;
    const int64_t *off = d->interval_off ? d->interval_off : d->interval_off_host;  // HOST offsets, either field
    const bool ragged = off != nullptr;
    if (!ragged && d->interval_len <= 0) return fail(FPT_ERR_INVALID, "interval_len must be positive");
    if (ragged)
        for (int64_t i = 0; i < n; ++i)
            if (off[i + 1] < off[i]) return fail(FPT_ERR_INVALID, "bad interval offsets");
    const int pad = d->half_win_width + d->smoothing_half_win_width, S = d->n_scales;
    const int64_t L = d->interval_len;
    auto base_at = [&](int64_t i) { return ragged ? off[i] - off[0] : i * L; };
    const int64_t total = base_at(n);
    if (chunk_bases <= 0) chunk_bases = (int64_t)1 << 21;

    if (!c->pipe) {
        c->pipe = new fpt_host_pipe();
        fpt_host_pipe *np_ = c->pipe;
        HIP_TRY(hipStreamCreateWithFlags(&np_->s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&np_->s_out, hipStreamNonBlocking));
        for (auto &s : np_->slot) {
            HIP_TRY(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&s.ev_scan, hipEventDisableTiming));
        }
    }
    fpt_host_pipe *P = c->pipe;
    double *outs[3 + FPT_MAX_SCALES];
    outs[0] = d->exp_out, outs[1] = d->obs_out, outs[2] = d->pval_out;
    for (int s = 0; s < S; ++s) outs[3 + s] = d->winp_out ? d->winp_out + (size_t)s * total : nullptr;

    fpt_scan_host_stats st = {};
    st.inputs_pinned = is_pinned(d->counts_plus) && is_pinned(d->counts_minus) && is_pinned(d->seq);
    st.outputs_pinned = 1;
    for (int k = 0; k < 3 + S; ++k)
        if (outs[k] && !is_pinned(outs[k])) st.outputs_pinned = 0;
    if (d->status_out && !is_pinned(d->status_out)) st.outputs_pinned = 0;
    const auto t_start = std::chrono::steady_clock::now();
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
    };

    // the second thread: every chunk's results to the caller's arrays, in the order the chunks were issued
    struct drain_t {
        std::mutex m;
        std::condition_variable cv;
        int64_t issued = 0, drained = 0;
        bool stop = false;
        hipError_t err = hipSuccess;
    } dr;
    std::thread drain([&] {
        (void)hipSetDevice(c->device);
        for (int64_t j = 0;; ++j) {
            {
                std::unique_lock<std::mutex> g(dr.m);
                dr.cv.wait(g, [&] { return dr.issued > j || dr.stop; });
                if (dr.issued <= j) return;
            }
            fpt_host_pipe::slot_t &sl = P->slot[j % kPipeSlots];
            const size_t tb = (size_t)sl.bases * 8;
            hipError_t e = hipStreamWaitEvent(P->s_out, sl.ev_scan, 0);
            for (int t = 0; t < 3 + S && e == hipSuccess; ++t)
                if (outs[t]) e = hipMemcpyAsync(outs[t] + sl.base0, sl.d_out + (size_t)t * tb, tb, hipMemcpyDeviceToHost, P->s_out);
            if (e == hipSuccess && d->status_out)
                e = hipMemcpyAsync(d->status_out + sl.iv0, sl.d_out + (size_t)(3 + S) * tb, (size_t)sl.n_iv * 4,
                                   hipMemcpyDeviceToHost, P->s_out);
            if (e == hipSuccess) e = hipStreamSynchronize(P->s_out);
            {
                std::lock_guard<std::mutex> g(dr.m);
                if (e != hipSuccess && dr.err == hipSuccess) dr.err = e;
                dr.drained = j + 1;
            }
            dr.cv.notify_all();
        }
    });
    struct drain_guard {  // every way out of this function stops and joins the thread
        drain_t &dr;
        std::thread &th;
        ~drain_guard() {
            if (!th.joinable()) return;
            {
                std::lock_guard<std::mutex> g(dr.m);
                dr.stop = true;
            }
            dr.cv.notify_all();
            th.join();
        }
    } guard{dr, drain};

    int64_t i0 = 0;
    for (int64_t k = 0; i0 < n; ++k) {
        // the chunk: intervals [i0, i1) holding about chunk_bases output bases (at least one interval)
        int64_t i1;
        if (!ragged) {
            i1 = std::min(n, i0 + std::max<int64_t>(1, chunk_bases / L));
        } else {
            const int64_t want = base_at(i0) + chunk_bases;
            i1 = std::upper_bound(off + i0 + 1, off + n + 1, want + off[0]) - off - 1;
            i1 = std::min(n, std::max(i1, i0 + 1));
        }
        const int64_t ni = i1 - i0, b0 = base_at(i0), cb = base_at(i1) - b0;
        const int64_t c0 = b0 + i0 * (2 * pad + 1), nc = cb + ni * (2 * pad + 1);   // counts: first element, elements
        const int64_t q0 = b0 + i0 * (2 * pad + 7), nq = cb + ni * (2 * pad + 7);   // sequence bytes
        fpt_host_pipe::slot_t &sl = P->slot[k % kPipeSlots];
        {  // the slot's previous chunk (k - kPipeSlots) must have been handed over
            const auto tw = tick();
            std::unique_lock<std::mutex> g(dr.m);
            dr.cv.wait(g, [&] { return dr.drained > k - kPipeSlots; });
            st.wait_seconds += since(tw);
            if (dr.err != hipSuccess) return fail(FPT_ERR_HIP, "device-to-host copy failed: %s", hipGetErrorString(dr.err));
        }
        // layout of the chunk's input block on the device: counts+, counts-, sequence, rebased offsets, model ids
        const size_t o_cm = (size_t)nc * 8, o_sq = 2 * o_cm, o_off = o_sq + up16((size_t)nq),
                     o_dm = o_off + up16(ragged ? (size_t)(ni + 1) * 8 : 0), in_bytes = o_dm + up16(d->dm_ids ? (size_t)ni * 4 : 0);
        const size_t tb = (size_t)cb * 8, out_bytes = (size_t)(3 + S) * tb + up16((size_t)ni * 4);
        if (int rc = pipe_grow(&sl.d_in, &sl.d_in_bytes, in_bytes, false)) return rc;
        if (int rc = pipe_grow(&sl.d_out, &sl.d_out_bytes, out_bytes, false)) return rc;
        if (int rc = pipe_grow(&sl.p_small, &sl.p_small_bytes, in_bytes - o_off + 16, true)) return rc;
        const auto ti = tick();
        char *small = sl.p_small;   // the chunk's offsets, rebased to its first interval, and its model ids: made here
        if (ragged) {
            int64_t *ro = (int64_t *)small;
            for (int64_t i = 0; i <= ni; ++i) ro[i] = off[i0 + i] - off[i0];
        }
        if (d->dm_ids) memcpy(small + (o_dm - o_off), d->dm_ids + i0, (size_t)ni * 4);
        HIP_TRY(hipMemcpyAsync(sl.d_in, d->counts_plus + c0, o_cm, hipMemcpyHostToDevice, P->s_in));
        HIP_TRY(hipMemcpyAsync(sl.d_in + o_cm, d->counts_minus + c0, o_cm, hipMemcpyHostToDevice, P->s_in));
        HIP_TRY(hipMemcpyAsync(sl.d_in + o_sq, d->seq + q0, (size_t)nq, hipMemcpyHostToDevice, P->s_in));
        if (in_bytes > o_off) HIP_TRY(hipMemcpyAsync(sl.d_in + o_off, small, in_bytes - o_off, hipMemcpyHostToDevice, P->s_in));
        HIP_TRY(hipEventRecord(sl.ev_in, P->s_in));
        HIP_TRY(hipStreamWaitEvent(c->stream, sl.ev_in, 0));
        fpt_scan_desc cd = *d;
        cd.n_intervals = ni;
        cd.interval_off = ragged ? (const int64_t *)(sl.d_in + o_off) : nullptr;
        cd.interval_off_host = ragged ? (const int64_t *)small : nullptr;
        cd.counts_plus = (const double *)sl.d_in;
        cd.counts_minus = (const double *)(sl.d_in + o_cm);
        cd.seq = (const uint8_t *)(sl.d_in + o_sq);
        cd.dm_ids = d->dm_ids ? (const int32_t *)(sl.d_in + o_dm) : nullptr;
        double *dout = (double *)sl.d_out;
        cd.exp_out = dout, cd.obs_out = dout + cb, cd.pval_out = dout + 2 * cb;
        cd.winp_out = S ? dout + 3 * cb : nullptr;
        cd.status_out = (int32_t *)(sl.d_out + (size_t)(3 + S) * tb);
        HIP_TRY(hipMemsetAsync(cd.status_out, 0, (size_t)ni * 4, c->stream));
        if (int rc = fpt_scan_dev(c, &cd)) return rc;
        HIP_TRY(hipEventRecord(sl.ev_scan, c->stream));
        sl.base0 = b0, sl.bases = cb, sl.iv0 = i0, sl.n_iv = ni;
        {
            std::lock_guard<std::mutex> g(dr.m);
            dr.issued = k + 1;
        }
        dr.cv.notify_all();
        st.issue_seconds += since(ti);
        st.chunks += 1;
        st.bytes_h2d += (int64_t)in_bytes;
        st.bytes_d2h += (int64_t)out_bytes;
        i0 = i1;
    }
    {
        const auto tw = tick();
        std::unique_lock<std::mutex> g(dr.m);
        dr.cv.wait(g, [&] { return dr.drained >= st.chunks; });
        st.wait_seconds += since(tw);
    }
    if (dr.err != hipSuccess) return fail(FPT_ERR_HIP, "device-to-host copy failed: %s", hipGetErrorString(dr.err));
    HIP_TRY(hipStreamSynchronize(c->stream));
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    st.bases = total;
    P->last = st;
    return FPT_OK;
}

#pragma GCC visibility pop
}  // extern "C"
