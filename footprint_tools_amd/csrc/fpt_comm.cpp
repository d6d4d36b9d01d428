// fpt_comm.cpp -- the one collective of the sharded job: the per-base track of every rank's shard
// re-assembled over RCCL (xGMI inside a node) -- on every rank (all-gather) or on the rank that
// writes (gather to a root), on the compute stream or on the communicator's own stream beside the
// next batch's scan.  RCCL is bound directly: librccl.so is opened with dlopen at the first
// call, so the library loads and every single-GPU entry point works where RCCL is absent.
// No PyTorch, no MPI: rank 0 makes a 128-byte id (fpt_comm_unique_id), the host program carries
// it to the other ranks by whatever it has (a file, a socket), every rank calls fpt_comm_init.
//
// Reference counterpart: none in the reference's data path -- its parallelism is processes
// writing through a queue to one writer (cli/detect.py:380-411, genome_tools processors); this
// call is what re-assembles the per-base statistics track on every rank (BASELINE.json north_star).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fpt.h"

// the few declarations of rccl.h this file needs (ABI of RCCL 2.x / librccl.so.1)
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
}

int fpt_internal_fail(int code, const char *fmt, ...);           // fpt_capi.cpp
hipStream_t fpt_internal_stream(fpt_ctx *c);
int fpt_internal_check_ctx(fpt_ctx *c);

namespace {

constexpr int kNcclFloat64 = 8;

struct rccl_api {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    // what the communicator itself says about the job (fpt_comm_info); optional symbols
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    int (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

rccl_api &api() {
    static rccl_api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // FPT_RCCL_LIB names the library to bind instead (tests: a stand-in that moves the bytes between
        // processes sharing one GPU, so that rank > 0 code runs on a one-GPU box); only when set
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        if (const char *own = getenv("FPT_RCCL_LIB")) {
            a.handle = dlopen(own, RTLD_NOW | RTLD_LOCAL);
            if (!a.handle) {
                a.error = std::string("FPT_RCCL_LIB=") + own + " cannot be loaded: " + (dlerror() ? dlerror() : "?");
                return;
            }
        }
        for (const char *n : names) {
            if (a.handle) break;
            a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!a.handle) {
            a.error = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(a.handle, n);
            if (!p && a.error.empty()) a.error = std::string("librccl.so lacks ") + n;
            return p;
        };
        a.GetUniqueId = (int (*)(ncclUniqueId *))sym("ncclGetUniqueId");
        a.CommInitRank = (int (*)(ncclComm_t *, int, ncclUniqueId, int))sym("ncclCommInitRank");
        a.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
        a.AllGather = (int (*)(const void *, void *, size_t, int, ncclComm_t, hipStream_t))sym("ncclAllGather");
        a.Broadcast = (int (*)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclBroadcast");
        a.Send = (int (*)(const void *, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclSend");
        a.Recv = (int (*)(void *, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclRecv");
        a.GroupStart = (int (*)())sym("ncclGroupStart");
        a.GroupEnd = (int (*)())sym("ncclGroupEnd");
        a.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
        a.CommCount = (int (*)(const ncclComm_t, int *))dlsym(a.handle, "ncclCommCount");
        a.CommUserRank = (int (*)(const ncclComm_t, int *))dlsym(a.handle, "ncclCommUserRank");
        a.CommCuDevice = (int (*)(const ncclComm_t, int *))dlsym(a.handle, "ncclCommCuDevice");
    });
    return a;
}

int rccl_ready() {
    rccl_api &a = api();
    if (!a.error.empty()) return fpt_internal_fail(FPT_ERR_HIP, "%s", a.error.c_str());
    return FPT_OK;
}

#define NCCL_TRY(expr)                                                                               \
    do {                                                                                             \
        int r_ = (expr);                                                                             \
        if (r_ != 0)                                                                                 \
            return fpt_internal_fail(FPT_ERR_HIP, "%s failed: %s", #expr, api().GetErrorString(r_)); \
    } while (0)

}  // namespace

struct fpt_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    int device = 0;
    bool force_ragged = false;  // FPT_COMM_RAGGED=1, read once when the communicator is made
    // the communicator's own stream (the _async entry points): a collective there runs beside the scan
    // of the next batch on the context's stream; `ready` orders it behind the producer, `done` is what
    // fpt_comm_wait / fpt_comm_synchronize wait for
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done[4] = {};  // done[n % 4]: behind the n-th asynchronous collective
    int64_t n_async = 0;
};

namespace {

#define HIP_TRY_C(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fpt_internal_fail(FPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int check_counts(const fpt_comm *k, const int64_t *counts, bool *equal_out, int64_t *total_out) {
    if (!k || !k->comm) return fpt_internal_fail(FPT_ERR_INVALID, "null communicator");
    if (!counts) return fpt_internal_fail(FPT_ERR_INVALID, "null counts");
    bool equal = true;
    int64_t total = 0;
    for (int r = 0; r < k->world; ++r) {
        if (counts[r] < 0) return fpt_internal_fail(FPT_ERR_INVALID, "negative shard length");
        equal = equal && counts[r] == counts[0];
        total += counts[r];
    }
    *equal_out = equal;
    *total_out = total;
    return FPT_OK;
}

// every rank receives every shard, in rank order
int do_allgather(fpt_comm *k, hipStream_t st, const double *send, const int64_t *counts, double *recv) {
    bool equal;
    int64_t total;
    if (int rc = check_counts(k, counts, &equal, &total)) return rc;
    if (!recv) return fpt_internal_fail(FPT_ERR_INVALID, "null receive buffer");
    if (!send && counts[k->rank] > 0) return fpt_internal_fail(FPT_ERR_INVALID, "null send buffer");
    if (equal && !k->force_ragged) {
        if (counts[0] == 0) return FPT_OK;
        NCCL_TRY(api().AllGather(send, recv, (size_t)counts[0], kNcclFloat64, k->comm, st));
        return FPT_OK;
    }
    // ragged shards (intervals balanced by padded bases end on interval boundaries): one
    // broadcast per shard inside a group -- the all-gather-v idiom, no padding and no staging copy
    NCCL_TRY(api().GroupStart());
    int64_t off = 0;
    for (int r = 0; r < k->world; ++r) {
        if (counts[r] > 0) {
            int rc = api().Broadcast(r == k->rank ? (const void *)send : (const void *)(recv + off), recv + off,
                                     (size_t)counts[r], kNcclFloat64, r, k->comm, st);
            if (rc != 0) {
                (void)api().GroupEnd();
                return fpt_internal_fail(FPT_ERR_HIP, "ncclBroadcast failed: %s", api().GetErrorString(rc));
            }
        }
        off += counts[r];
    }
    NCCL_TRY(api().GroupEnd());
    return FPT_OK;
}

// ONE rank receives every shard (the rank that writes: cli/detect.py:396-408 has one writer): grouped
// ncclSend / ncclRecv -- 1/world of the all-gather's traffic, all of it on the root's links
int do_gather(fpt_comm *k, hipStream_t st, const double *send, const int64_t *counts, double *recv, int root) {
    bool equal;
    int64_t total;
    if (int rc = check_counts(k, counts, &equal, &total)) return rc;
    if (root < 0 || root >= k->world) return fpt_internal_fail(FPT_ERR_INVALID, "root %d out of range", root);
    if (!send && counts[k->rank] > 0) return fpt_internal_fail(FPT_ERR_INVALID, "null send buffer");
    if (k->rank == root && !recv && total > 0) return fpt_internal_fail(FPT_ERR_INVALID, "null receive buffer on the root");
    if (!api().Send || !api().Recv) return fpt_internal_fail(FPT_ERR_HIP, "this librccl has no ncclSend / ncclRecv");
    if (k->rank != root) {
        if (counts[k->rank] > 0) NCCL_TRY(api().Send(send, (size_t)counts[k->rank], kNcclFloat64, root, k->comm, st));
        return FPT_OK;
    }
    int64_t off = 0, my_off = 0;
    NCCL_TRY(api().GroupStart());
    for (int r = 0; r < k->world; ++r) {
        if (r == root) my_off = off;
        else if (counts[r] > 0) {
            int rc = api().Recv(recv + off, (size_t)counts[r], kNcclFloat64, r, k->comm, st);
            if (rc != 0) {
                (void)api().GroupEnd();
                return fpt_internal_fail(FPT_ERR_HIP, "ncclRecv failed: %s", api().GetErrorString(rc));
            }
        }
        off += counts[r];
    }
    NCCL_TRY(api().GroupEnd());
    if (counts[root] > 0 && send != recv + my_off)  // the root's own shard (in place: already where it belongs)
        HIP_TRY_C(hipMemcpyAsync(recv + my_off, send, (size_t)counts[root] * sizeof(double), hipMemcpyDeviceToDevice, st));
    return FPT_OK;
}

// the communicator's stream, ordered behind everything the context's stream holds now
int async_begin(fpt_ctx *c, fpt_comm *k, hipStream_t *st_out) {
    if (int rc = fpt_internal_check_ctx(c)) return rc;
    if (!k || !k->comm) return fpt_internal_fail(FPT_ERR_INVALID, "null communicator");
    if (!k->stream) {
        HIP_TRY_C(hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking));
        HIP_TRY_C(hipEventCreateWithFlags(&k->ready, hipEventDisableTiming));
        for (hipEvent_t &e : k->done) HIP_TRY_C(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIP_TRY_C(hipEventRecord(k->ready, fpt_internal_stream(c)));
    HIP_TRY_C(hipStreamWaitEvent(k->stream, k->ready, 0));
    *st_out = k->stream;
    return FPT_OK;
}
int async_end(fpt_comm *k) {
    HIP_TRY_C(hipEventRecord(k->done[k->n_async % 4], k->stream));
    ++k->n_async;
    return FPT_OK;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int fpt_comm_unique_id(uint8_t id_out[FPT_COMM_ID_BYTES]) {
    if (!id_out) return fpt_internal_fail(FPT_ERR_INVALID, "null id buffer");
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId id;
    NCCL_TRY(api().GetUniqueId(&id));
    std::memcpy(id_out, id.internal, FPT_COMM_ID_BYTES);
    return FPT_OK;
}

int fpt_comm_init(fpt_ctx *c, const uint8_t id[FPT_COMM_ID_BYTES], int world_size, int rank, fpt_comm **out) {
    if (!out) return fpt_internal_fail(FPT_ERR_INVALID, "null output");
    *out = nullptr;
    if (int rc = fpt_internal_check_ctx(c)) return rc;  // makes the context's device current
    if (!id || world_size < 1 || rank < 0 || rank >= world_size)
        return fpt_internal_fail(FPT_ERR_INVALID, "bad communicator arguments (world %d, rank %d)", world_size, rank);
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, FPT_COMM_ID_BYTES);
    fpt_comm *k = new fpt_comm();
    k->world = world_size;
    k->rank = rank;
    // FPT_COMM_RAGGED=1 sends equal shards down the ragged path too (a one-GPU box can then run it)
    const char *force = getenv("FPT_COMM_RAGGED");
    k->force_ragged = force && force[0] == '1';
    // ncclCommInitRank blocks until every rank has called it with the same id and has no timeout
    // of its own: a missing peer, or one holding another id, would hang the job.  It runs on a
    // helper thread; if it has not returned after FPT_COMM_TIMEOUT_S seconds (default 300) this
    // call fails with a message, and the helper is left behind (the process is expected to end).
    int device = 0;
    (void)hipGetDevice(&device);
    double timeout_s = 300.0;
    if (const char *e = getenv("FPT_COMM_TIMEOUT_S")) timeout_s = atof(e) > 0 ? atof(e) : timeout_s;
    struct init_state {
        std::promise<int> done;
        ncclComm_t comm = nullptr;
    };
    auto st = std::make_shared<init_state>();
    std::future<int> fut = st->done.get_future();
    std::thread([st, device, world_size, uid, rank]() {
        (void)hipSetDevice(device);
        const int r = api().CommInitRank(&st->comm, world_size, uid, rank);
        st->done.set_value(r);
    }).detach();
    if (fut.wait_for(std::chrono::duration<double>(timeout_s)) != std::future_status::ready) {
        delete k;
        return fpt_internal_fail(FPT_ERR_HIP,
                                 "ncclCommInitRank did not return within %.0f s (rank %d of %d): a rank is missing or "
                                 "holds a different communicator id", timeout_s, rank, world_size);
    }
    const int r = fut.get();
    if (r != 0) {
        delete k;
        return fpt_internal_fail(FPT_ERR_HIP, "ncclCommInitRank failed: %s", api().GetErrorString(r));
    }
    k->comm = st->comm;
    k->device = device;
    *out = k;
    return FPT_OK;
}

int fpt_comm_info(fpt_comm *k, fpt_comm_info_t *out) {
    if (!k || !k->comm) return fpt_internal_fail(FPT_ERR_INVALID, "null communicator");
    if (!out) return fpt_internal_fail(FPT_ERR_INVALID, "null output");
    std::memset(out, 0, sizeof *out);
    out->world_size = k->world;
    out->rank = k->rank;
    out->device = k->device;
    out->rccl_count = out->rccl_user_rank = out->rccl_device = -1;
    // asked of the communicator, not echoed from the arguments of fpt_comm_init: a job whose ranks
    // ended up in communicators of their own (or on one device) shows here
    if (api().CommCount) NCCL_TRY(api().CommCount(k->comm, &out->rccl_count));
    if (api().CommUserRank) NCCL_TRY(api().CommUserRank(k->comm, &out->rccl_user_rank));
    if (api().CommCuDevice) NCCL_TRY(api().CommCuDevice(k->comm, &out->rccl_device));
    HIP_TRY_C(hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof out->pci_bus_id, k->device));
    return FPT_OK;
}

int fpt_comm_destroy(fpt_comm *k) {
    if (!k) return FPT_OK;
    if (k->stream) {
        (void)hipSetDevice(k->device);
        (void)hipStreamSynchronize(k->stream);
    }
    if (k->comm && api().CommDestroy) (void)api().CommDestroy(k->comm);
    if (k->ready) (void)hipEventDestroy(k->ready);
    for (hipEvent_t e : k->done)
        if (e) (void)hipEventDestroy(e);
    if (k->stream) (void)hipStreamDestroy(k->stream);
    delete k;
    return FPT_OK;
}

int fpt_allgather_track(fpt_ctx *c, fpt_comm *k, const double *send, const int64_t *counts, double *recv) {
    if (int rc = fpt_internal_check_ctx(c)) return rc;
    return do_allgather(k, fpt_internal_stream(c), send, counts, recv);
}

int fpt_gather_track(fpt_ctx *c, fpt_comm *k, const double *send, const int64_t *counts, double *recv, int root) {
    if (int rc = fpt_internal_check_ctx(c)) return rc;
    return do_gather(k, fpt_internal_stream(c), send, counts, recv, root);
}

int fpt_allgather_track_async(fpt_ctx *c, fpt_comm *k, const double *send, const int64_t *counts, double *recv) {
    hipStream_t st;
    if (int rc = async_begin(c, k, &st)) return rc;
    if (int rc = do_allgather(k, st, send, counts, recv)) return rc;
    return async_end(k);
}

int fpt_gather_track_async(fpt_ctx *c, fpt_comm *k, const double *send, const int64_t *counts, double *recv, int root) {
    hipStream_t st;
    if (int rc = async_begin(c, k, &st)) return rc;
    if (int rc = do_gather(k, st, send, counts, recv, root)) return rc;
    return async_end(k);
}

int fpt_comm_wait(fpt_ctx *c, fpt_comm *k, int back) {
    if (int rc = fpt_internal_check_ctx(c)) return rc;
    if (!k) return fpt_internal_fail(FPT_ERR_INVALID, "null communicator");
    if (back < 0 || back > 3) return fpt_internal_fail(FPT_ERR_INVALID, "fpt_comm_wait: back must be 0..3");
    if (k->n_async > back) HIP_TRY_C(hipStreamWaitEvent(fpt_internal_stream(c), k->done[(k->n_async - 1 - back) % 4], 0));
    return FPT_OK;
}

int fpt_comm_synchronize(fpt_comm *k) {
    if (!k) return fpt_internal_fail(FPT_ERR_INVALID, "null communicator");
    if (k->n_async > 0) {
        HIP_TRY_C(hipSetDevice(k->device));
        HIP_TRY_C(hipEventSynchronize(k->done[(k->n_async - 1) % 4]));
    }
    return FPT_OK;
}

#pragma GCC visibility pop
}
