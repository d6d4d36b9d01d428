// fakerccl.cpp -- TEST-ONLY stand-in for librccl.so: the entry points fpt_comm.cpp binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclBroadcast, ncclSend, ncclRecv,
// ncclGroupStart, ncclGroupEnd, ncclGetErrorString, ncclCommCount, ncclCommUserRank, ncclCommCuDevice) implemented between PROCESSES THAT SHARE ONE GPU:
// a rendezvous in /dev/shm keyed by the unique id, device buffers handed over as HIP IPC handles, the
// bytes moved with hipMemcpy.  RCCL refuses two ranks on one device, and a gpurun box has one: with
// FPT_RCCL_LIB pointing here everything around the collective -- rank > 0 rendezvous, shard offsets,
// in-place semantics, the grouped broadcasts with a non-root rank, send/recv matching, bench.py
// --gpus 2 -- runs end to end on that box.  It proves nothing about RCCL itself.
//
// Stricter than RCCL on purpose: every call synchronises the stream it is given and the ranks meet at
// two barriers (data ready / data taken), so a missing wait in the caller cannot hide behind timing.
//   hipcc -O2 -fPIC -shared tests/fakerccl/fakerccl.cpp -o tests/fakerccl/libfakerccl.so
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 16, kMaxOps = 64;
constexpr double kTimeoutS = 60.0;

struct slot_t {  // what a rank offers for one operation of the current call
    hipIpcMemHandle_t handle;
    uint64_t offset, bytes;
    int32_t peer;   // ncclSend: the receiver; -1: whoever takes part (broadcast / all-gather)
    int32_t valid;
};
struct shared_t {
    std::atomic<int> arrived, left;
    std::atomic<int> bar_count;
    std::atomic<int> bar_gen;
    slot_t slots[kMaxRanks][kMaxOps];
};

struct op_t {
    int kind;  // 0 broadcast (root), 1 send (peer), 2 recv (peer)
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    hipStream_t stream;
};

struct opened_t {  // a peer's allocation mapped into this process (kept until the communicator goes)
    hipIpcMemHandle_t handle;
    void *base;
};

struct comm_t {
    int world = 1, rank = 0;
    shared_t *sh = nullptr;
    std::string path;
    std::vector<op_t> ops;
    std::vector<opened_t> opened;
};

thread_local int g_group_depth = 0;
thread_local comm_t *g_group_comm = nullptr;
const char *g_last = "fakerccl: ok";

int fail(const char *m) {
    g_last = m;
    fprintf(stderr, "fakerccl: %s\n", m);
    return 1;  // ncclUnhandledCudaError
}

size_t type_size(int t) {
    switch (t) {
        case 0: case 1: return 1;
        case 6: case 9: return 2;
        case 2: case 3: case 7: return 4;
        default: return 8;
    }
}

bool barrier(comm_t *c) {
    shared_t *s = c->sh;
    const int gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == c->world) {
        s->bar_count.store(0);
        s->bar_gen.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->bar_gen.load() == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return false;
    }
    return true;
}

// publish `p` (device memory of THIS process) in my slot `i`
bool offer(comm_t *c, int i, const void *p, size_t bytes, int peer) {
    slot_t &sl = c->sh->slots[c->rank][i];
    void *base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)p) != hipSuccess) return false;
    if (hipIpcGetMemHandle(&sl.handle, base) != hipSuccess) return false;
    sl.offset = (uint64_t)((const char *)p - (const char *)base);
    sl.bytes = bytes;
    sl.peer = peer;
    sl.valid = 1;
    return true;
}

// A peer's allocation is mapped ONCE and stays mapped: a barrier of the job is a tiny all-gather, and opening and
// closing seven handles for each of them -- with the copy possibly still in flight when the handle was closed: a
// device-to-device hipMemcpy need not have finished when it returns -- made ranks fail to open a handle now and
// then ("cannot read a broadcast source of another rank", the others then waiting out the barrier's 60 s).
bool take(comm_t *c, int from, int i, void *dst, size_t bytes) {
    const slot_t &sl = c->sh->slots[from][i];
    if (!sl.valid || sl.bytes != bytes) return false;
    void *base = nullptr;
    for (const opened_t &o : c->opened)
        if (memcmp(&o.handle, &sl.handle, sizeof sl.handle) == 0) base = o.base;
    if (!base) {
        hipError_t e = hipIpcOpenMemHandle(&base, sl.handle, hipIpcMemLazyEnablePeerAccess);
        for (int tries = 0; e != hipSuccess && tries < 50; ++tries) {  // (the exporter may be busy in the driver)
            (void)hipGetLastError();
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            e = hipIpcOpenMemHandle(&base, sl.handle, hipIpcMemLazyEnablePeerAccess);
        }
        if (e != hipSuccess) {
            fprintf(stderr, "fakerccl: hipIpcOpenMemHandle (rank %d reading rank %d): %s\n", c->rank, from, hipGetErrorString(e));
            return false;
        }
        c->opened.push_back(opened_t{sl.handle, base});
    }
    hipError_t e = hipMemcpy(dst, (const char *)base + sl.offset, bytes, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();  // the copy is done before anybody is told so
    if (e != hipSuccess) fprintf(stderr, "fakerccl: copy from rank %d failed: %s\n", from, hipGetErrorString(e));
    return e == hipSuccess;
}

// every rank calls this with the same sequence of operations (collectives) or matching ones (send / recv)
int run_ops(comm_t *c) {
    std::vector<op_t> ops;
    ops.swap(c->ops);
    if (ops.empty()) return 0;
    if ((int)ops.size() > kMaxOps) return fail("too many operations in one group");
    for (const op_t &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return fail("hipStreamSynchronize failed");
    for (int i = 0; i < kMaxOps; ++i) c->sh->slots[c->rank][i].valid = 0;
    // 1. offer what others will read: slot index = position in the call sequence for collectives, the
    //    n-th send to a peer for point-to-point
    std::vector<int> send_seq(c->world, 0);
    for (size_t i = 0; i < ops.size(); ++i) {
        const op_t &o = ops[i];
        if (o.kind == 0 && o.peer == c->rank && o.bytes) {
            if (!offer(c, (int)i, o.src, o.bytes, -1)) return fail("cannot export a broadcast source (hipIpcGetMemHandle)");
        } else if (o.kind == 1 && o.bytes) {
            // slots of sends are filled from the top so that they never meet a collective's index
            const int idx = kMaxOps - 1 - (o.peer * 4 + send_seq[o.peer]++);
            if (idx < (int)ops.size() || send_seq[o.peer] > 4) return fail("too many sends in one group");
            if (!offer(c, idx, o.src, o.bytes, o.peer)) return fail("cannot export a send buffer (hipIpcGetMemHandle)");
        }
    }
    if (!barrier(c)) return fail("barrier timed out (data ready): a rank is missing");
    // 2. take
    std::vector<int> recv_seq(c->world, 0);
    for (size_t i = 0; i < ops.size(); ++i) {
        const op_t &o = ops[i];
        if (!o.bytes) continue;
        if (o.kind == 0) {
            if (o.peer == c->rank) {
                if (o.dst != o.src && hipMemcpy(o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                    return fail("local copy failed");
            } else if (!take(c, o.peer, (int)i, o.dst, o.bytes)) return fail("cannot read a broadcast source of another rank");
        } else if (o.kind == 2) {
            const int idx = kMaxOps - 1 - (c->rank * 4 + recv_seq[o.peer]++);
            if (c->sh->slots[o.peer][idx].peer != c->rank) return fail("recv without a matching send");
            if (!take(c, o.peer, idx, o.dst, o.bytes)) return fail("cannot read a send buffer of another rank");
        }
    }
    // 3. nobody reuses a source before everybody has taken
    if (!barrier(c)) return fail("barrier timed out (data taken)");
    return 0;
}

int submit(comm_t *c, const op_t &o) {
    c->ops.push_back(o);
    if (g_group_depth > 0) {
        g_group_comm = c;
        return 0;
    }
    return run_ops(c);
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

struct ncclUniqueId_ {
    char internal[128];
};

int ncclGetUniqueId(ncclUniqueId_ *id) {
    memset(id, 0, sizeof *id);
    unsigned char r[16] = {0};
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) {
        if (read(fd, r, sizeof r) != (ssize_t)sizeof r) memset(r, 0x5a, sizeof r);
        close(fd);
    }
    char *p = id->internal;
    p += sprintf(p, "fakerccl_%d_", (int)getpid());
    for (unsigned char b : r) p += sprintf(p, "%02x", b);
    return 0;
}

int ncclCommInitRank(void **comm_out, int nranks, ncclUniqueId_ id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return fail("bad rank / world size");
    id.internal[127] = 0;
    comm_t *c = new comm_t();
    c->world = nranks;
    c->rank = rank;
    c->path = std::string("/dev/shm/") + id.internal;
    const int fd = open(c->path.c_str(), O_RDWR | O_CREAT, 0600);
    if (fd < 0) return fail("cannot create the rendezvous file in /dev/shm");
    if (ftruncate(fd, sizeof(shared_t)) != 0) return fail("ftruncate failed");
    void *m = mmap(nullptr, sizeof(shared_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail("mmap failed");
    c->sh = (shared_t *)m;  // (a fresh file is all zeros: counters start at 0)
    c->sh->arrived.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->arrived.load() < nranks) {
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS)
            return fail("ncclCommInitRank: not every rank arrived");
    }
    *comm_out = c;
    return 0;
}

int ncclCommDestroy(void *comm) {
    comm_t *c = (comm_t *)comm;
    if (!c) return 0;
    for (const opened_t &o : c->opened) (void)hipIpcCloseMemHandle(o.base);
    if (c->sh) {
        if (c->sh->left.fetch_add(1) + 1 == c->world) unlink(c->path.c_str());
        munmap(c->sh, sizeof(shared_t));
    }
    delete c;
    return 0;
}

int ncclCommCount(void *comm, int *count) {
    if (!comm || !count) return fail("ncclCommCount: null argument");
    *count = ((comm_t *)comm)->world;
    return 0;
}

int ncclCommUserRank(void *comm, int *rank) {
    if (!comm || !rank) return fail("ncclCommUserRank: null argument");
    *rank = ((comm_t *)comm)->rank;
    return 0;
}

int ncclCommCuDevice(void *comm, int *device) {
    if (!comm || !device) return fail("ncclCommCuDevice: null argument");
    return hipGetDevice(device) == hipSuccess ? 0 : fail("hipGetDevice failed");
}

int ncclBroadcast(const void *send, void *recv, size_t count, int type, int root, void *comm, hipStream_t st) {
    comm_t *c = (comm_t *)comm;
    return submit(c, op_t{0, send, recv, count * type_size(type), root, st});
}

int ncclAllGather(const void *send, void *recv, size_t count, int type, void *comm, hipStream_t st) {
    comm_t *c = (comm_t *)comm;
    const size_t bytes = count * type_size(type);
    const bool own_group = g_group_depth == 0;
    if (own_group) ++g_group_depth;
    for (int r = 0; r < c->world; ++r)
        submit(c, op_t{0, send, (char *)recv + (size_t)r * bytes, bytes, r, st});
    if (own_group) {
        --g_group_depth;
        return run_ops(c);
    }
    return 0;
}

int ncclSend(const void *send, size_t count, int type, int peer, void *comm, hipStream_t st) {
    return submit((comm_t *)comm, op_t{1, send, nullptr, count * type_size(type), peer, st});
}

int ncclRecv(void *recv, size_t count, int type, int peer, void *comm, hipStream_t st) {
    return submit((comm_t *)comm, op_t{2, nullptr, recv, count * type_size(type), peer, st});
}

int ncclGroupStart() {
    ++g_group_depth;
    return 0;
}

int ncclGroupEnd() {
    if (g_group_depth <= 0) return fail("ncclGroupEnd without ncclGroupStart");
    if (--g_group_depth == 0 && g_group_comm) {
        comm_t *c = g_group_comm;
        g_group_comm = nullptr;
        return run_ops(c);
    }
    return 0;
}

const char *ncclGetErrorString(int) { return g_last; }

#pragma GCC visibility pop
}
