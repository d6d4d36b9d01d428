// fpt_comm.cpp -- the one collective of the sharded job: an all-gather of the per-base track
// over RCCL (xGMI inside a node), bound directly: librccl.so is opened with dlopen at the first
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
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

rccl_api &api() {
    static rccl_api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names)
            if ((a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
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
        a.GroupStart = (int (*)())sym("ncclGroupStart");
        a.GroupEnd = (int (*)())sym("ncclGroupEnd");
        a.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
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
    bool force_ragged = false;  // FPT_COMM_RAGGED=1, read once when the communicator is made
};

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
    *out = k;
    return FPT_OK;
}

int fpt_comm_destroy(fpt_comm *k) {
    if (!k) return FPT_OK;
    if (k->comm && api().CommDestroy) (void)api().CommDestroy(k->comm);
    delete k;
    return FPT_OK;
}

int fpt_allgather_track(fpt_ctx *c, fpt_comm *k, const double *send, const int64_t *counts, double *recv) {
    if (int rc = fpt_internal_check_ctx(c)) return rc;
    if (!k || !k->comm) return fpt_internal_fail(FPT_ERR_INVALID, "null communicator");
    if (!counts || !recv) return fpt_internal_fail(FPT_ERR_INVALID, "null counts / receive buffer");
    bool equal = true;
    for (int r = 0; r < k->world; ++r) {
        if (counts[r] < 0) return fpt_internal_fail(FPT_ERR_INVALID, "negative shard length");
        equal = equal && counts[r] == counts[0];
    }
    if (!send && counts[k->rank] > 0) return fpt_internal_fail(FPT_ERR_INVALID, "null send buffer");
    hipStream_t st = fpt_internal_stream(c);
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

#pragma GCC visibility pop
}
