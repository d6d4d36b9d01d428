"""Interval-sharded scans across the GPUs of one node: one process and one context per GPU.

Intervals are independent (every window is confined to its own padded interval), so a job is
split into contiguous interval ranges balanced by padded bases (`scan.shard_intervals`), every
rank scans its own range with no communication, and the per-base statistics track is
re-assembled with ONE collective per batch: an all-gather over RCCL / xGMI (every rank holds the
track), or a gather to the rank that writes it (`gather_dev`), on the compute stream or -- the
`_async` forms -- on the communicator's own stream beside the next batch's scan.

RCCL is bound directly by the library (`fpt_comm_*`, `fpt_allgather_track` in include/fpt.h;
librccl.so through dlopen) -- no PyTorch, no MPI.  The only thing the host program has to carry
between the processes is the 128-byte communicator id that rank 0 makes; `TrackComm` does that
through a file (the ranks of one node share /tmp) whose name is unique per job and per
communicator (launcher port, launcher pid, run id, a per-process counter -- or FPT_COMM_FILE) and
which only counts while rank 0 keeps touching it, so a leftover of an aborted run is never read
as the new id.  `ncclCommInitRank` runs under a timeout (FPT_COMM_TIMEOUT_S).  Barriers and the
max-over-ranks of a timing are tiny all-gathers on the same communicator.

Reference counterpart: cli/detect.py:380-411 -- worker processes (`batch_iter(num_workers=...)`)
compute per-interval statistics and one writer thread formats them.  `sharded_deviation_stats`
is that shape with GPUs for workers: every rank computes its shard, the statistics are gathered,
rank 0 writes.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib
from .scan import DeviceArray, shard_intervals


def shard_track_sizes(lengths, bounds):
    """bases owned by each rank given interval lengths and [(first, last), ...] ranges."""
    lengths = np.asarray(lengths, dtype=np.int64)
    return [int(lengths[a:b].sum()) for a, b in bounds]


def shard_offsets(counts):
    """where each rank's slice starts in the assembled track: world_size + 1 offsets (the last one is the
    track's length) -- what fpt_allgather_track / fpt_gather_track use for `recv`, and what a caller
    needs to place its own slice for the in-place forms"""
    return np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))]).astype(np.int64)


def rank_info():
    """(rank, world_size, local_rank) from the launcher's environment (torch.distributed.run,
    mpirun or the caller's own): RANK / WORLD_SIZE / LOCAL_RANK, defaults 0 / 1 / rank."""
    rank = int(os.environ.get("RANK", os.environ.get("OMPI_COMM_WORLD_RANK", "0")))
    world = int(os.environ.get("WORLD_SIZE", os.environ.get("OMPI_COMM_WORLD_SIZE", "1")))
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("OMPI_COMM_WORLD_LOCAL_RANK", str(rank))))
    return rank, world, local


_comm_counter = [0]  # communicators made by this process so far: part of the rendezvous name


def _id_path():
    """Where rank 0 leaves the communicator id for the other ranks of THIS job and THIS
    communicator: FPT_COMM_FILE if given (a launcher-supplied unique path), else a name made of
    the launcher's rendezvous port, the launcher's pid (the parent all ranks share) and the number
    of communicators this process has made before (every rank makes them in the same order)."""
    n = _comm_counter[0]
    p = os.environ.get("FPT_COMM_FILE")
    if p:
        return p if n == 0 else "%s.%d" % (p, n)
    token = os.environ.get("TORCHELASTIC_RUN_ID", "")
    token = "".join(ch for ch in token if ch.isalnum())[:32]
    return os.path.join(os.environ.get("TMPDIR", "/tmp"),
                        "fpt_comm_%s_%d_%s_%d.id" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), token, n))


_ID_BYTES = 128
_FRESH_S = 5.0      # an id file counts only while rank 0 keeps touching it (every _BEAT_S seconds)
_BEAT_S = 0.25


def _read_fresh_id(path):
    """The id in `path` if the file is a live offer: 128 bytes, owned by this user, and touched
    within the last _FRESH_S seconds -- rank 0 keeps touching its file until every rank has joined,
    so the leftover of an aborted run (same shell, same port) is never taken for the new id."""
    try:
        st = os.stat(path)
    except OSError:
        return None
    if st.st_size != _ID_BYTES or st.st_uid != os.getuid() or time.time() - st.st_mtime > _FRESH_S:
        return None
    try:
        with open(path, "rb") as f:
            raw = f.read()
    except OSError:
        return None
    return raw if len(raw) == _ID_BYTES else None


class _id_offer(object):
    """Rank 0's side of the rendezvous: the id file under `path`, kept fresh until withdrawn."""

    def __init__(self, path, ident_bytes):
        import threading
        self.path = path
        # a leftover of another run goes first; the new file is created exclusively (a file
        # somebody else planted under the temporary name is an error) and appears complete (rename)
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
        tmp = "%s.%d.tmp" % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(bytes(ident_bytes))
        os.replace(tmp, path)
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._keep_fresh, daemon=True)
        self._thread.start()

    def _keep_fresh(self):
        while not self._stop.wait(_BEAT_S):
            try:
                os.utime(self.path)
            except OSError:
                return

    def withdraw(self):
        self._stop.set()
        self._thread.join()
        try:
            os.remove(self.path)
        except OSError:
            pass


def _await_id(path, timeout_s):
    """The other ranks' side: wait for a live offer under `path`."""
    t0 = time.time()
    while True:
        raw = _read_fresh_id(path)
        if raw is not None:
            return raw
        if time.time() - t0 > timeout_s:
            raise TimeoutError("no live communicator id at %s after %.0f s (is rank 0 running?)" % (path, timeout_s))
        time.sleep(0.01)


class TrackComm(object):
    """RCCL communicator of the job's ranks, for the track all-gather (and barriers / timing).

    ctx   : this rank's _lib.Context (its GPU)
    rank, world : default from the environment (rank_info)
    path  : file through which rank 0 hands the communicator id to the others
    timeout_s : how long a rank waits for the id, and (through FPT_COMM_TIMEOUT_S, read by the
                library) how long ncclCommInitRank may take before the call fails instead of hanging
    """

    def __init__(self, ctx, rank=None, world=None, path=None, timeout_s=300.0):
        r, w, _ = rank_info()
        self.ctx, self.L = ctx, ctx.L
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        path = path or _id_path()
        _comm_counter[0] += 1
        ident = (C.c_uint8 * _ID_BYTES)()
        offer = None
        if self.rank == 0:
            _lib.check(self.L.fpt_comm_unique_id(ident))
            if self.world > 1:
                offer = _id_offer(path, bytes(ident))
        else:
            ident = (C.c_uint8 * _ID_BYTES).from_buffer_copy(_await_id(path, timeout_s))
        h = C.c_void_p()
        os.environ.setdefault("FPT_COMM_TIMEOUT_S", "%d" % max(1, int(timeout_s)))
        # RCCL prints a version banner on stdout while rank 0 initialises; programs that print
        # a result on stdout (bench.py: ONE JSON line) get it on stderr instead
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            rc = self.L.fpt_comm_init(ctx.h, ident, self.world, self.rank, C.byref(h))
            C.CDLL(None).fflush(None)  # the banner sits in the C library's buffer: push it out now
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            if offer is not None:  # every rank is through init (or it failed): withdraw the offer
                offer.withdraw()
        _lib.check(rc)
        self.h = h
        self._scratch = DeviceArray(ctx, 8 * (self.world + 1))
        self._path = path
        self.barrier()

    def close(self):
        if getattr(self, "h", None):
            self.ctx.synchronize()
            self.L.fpt_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the collective
    def _counts(self, counts):
        """one int64 count per rank, contiguous (the library reads counts[0 .. world) unconditionally)"""
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        if counts.ndim != 1 or counts.size != self.world:
            raise ValueError("need one count per rank")
        return counts

    def allgather_dev(self, send_ptr, counts, recv_ptr):
        """Enqueue the all-gather of a per-base track on the context's stream: rank r contributes
        counts[r] doubles at device pointer send_ptr, every rank receives sum(counts) doubles at
        recv_ptr in rank order.  Does not synchronise."""
        counts = self._counts(counts)
        _lib.check(self.L.fpt_allgather_track(self.ctx.h, self.h, send_ptr, counts.ctypes.data, recv_ptr))

    def gather_dev(self, send_ptr, counts, recv_ptr, root=0):
        """The same shards to ONE rank (the writer): recv_ptr is read on `root` only (None elsewhere)."""
        counts = self._counts(counts)
        _lib.check(self.L.fpt_gather_track(self.ctx.h, self.h, send_ptr, counts.ctypes.data, recv_ptr, int(root)))

    def allgather_dev_async(self, send_ptr, counts, recv_ptr):
        """allgather_dev on the communicator's own stream, behind what the context's stream holds now:
        the track of one batch travels while the next is scanned (into another buffer).  `wait` /
        `synchronize` before the buffers are touched again."""
        counts = self._counts(counts)
        _lib.check(self.L.fpt_allgather_track_async(self.ctx.h, self.h, send_ptr, counts.ctypes.data, recv_ptr))

    def gather_dev_async(self, send_ptr, counts, recv_ptr, root=0):
        counts = self._counts(counts)
        _lib.check(self.L.fpt_gather_track_async(self.ctx.h, self.h, send_ptr, counts.ctypes.data, recv_ptr, int(root)))

    def wait(self, back=0):
        """the context's stream waits for an asynchronous collective: the last one enqueued (back=0), the
        one before it (1: what a job with two track buffers in turn waits for before its next scan), ..."""
        _lib.check(self.L.fpt_comm_wait(self.ctx.h, self.h, int(back)))

    def synchronize(self):
        """the host waits for the last asynchronous collective"""
        _lib.check(self.L.fpt_comm_synchronize(self.h))

    def allgather_host(self, value):
        """one double per rank -> list of all ranks' values (a tiny all-gather; synchronises)"""
        mine = np.array([float(value)])
        _lib.check(self.L.fpt_memcpy_h2d(self.ctx.h, self._scratch.ptr + 8 * self.world, mine.ctypes.data, 8))
        self.allgather_dev(self._scratch.ptr + 8 * self.world, np.ones(self.world, np.int64), self._scratch.ptr)
        self.ctx.synchronize()
        return self._scratch.download(np.float64, self.world).tolist()

    def barrier(self):
        self.allgather_host(0.0)

    def info(self):
        """this rank as the communicator sees it: world / rank as given, what RCCL itself reports
        (ncclCommCount, ncclCommUserRank, ncclCommCuDevice; -1 where the bound library lacks them),
        the device ordinal and its PCI bus id"""
        ci = _lib.CommInfo()
        _lib.check(self.L.fpt_comm_info(self.h, C.byref(ci)))
        return dict(world_size=ci.world_size, rank=ci.rank, device=ci.device, rccl_count=ci.rccl_count,
                    rccl_user_rank=ci.rccl_user_rank, rccl_device=ci.rccl_device,
                    pci_bus_id=ci.pci_bus_id.decode("ascii", "replace"))

    def job_info(self):
        """`info()` of every rank, in rank order, gathered over the communicator itself (four tiny
        all-gathers: the bus id travels as its domain / bus / device / function numbers)"""
        me = self.info()
        try:
            dom, bus, rest = me["pci_bus_id"].split(":")
            dev, fn = rest.split(".")
            bdf = (int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(dev, 16) << 3) | int(fn, 16)
        except ValueError:
            bdf = -1
        cols = [self.allgather_host(v) for v in (me["rccl_count"], me["rccl_user_rank"], me["rccl_device"], bdf)]
        out = []
        for r in range(self.world):
            b = int(cols[3][r])
            out.append(dict(rank=r, rccl_count=int(cols[0][r]), rccl_user_rank=int(cols[1][r]),
                            device=int(cols[2][r]),
                            pci_bus_id=("%04x:%02x:%02x.%x" % (b >> 16, (b >> 8) & 0xff, (b >> 3) & 0x1f, b & 7)) if b >= 0 else None))
        return out

    def max_over_ranks(self, value):
        return max(self.allgather_host(value))

    def allgather_rows(self, local, counts):
        """Host convenience: `local` is this rank's (counts[rank], k) float64 matrix; returns the
        (sum(counts), k) matrix of all ranks in rank order (through device memory and RCCL)."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        k = local.shape[1] if local.ndim == 2 else 1
        counts = np.asarray(counts, dtype=np.int64)
        if local.size != counts[self.rank] * k:
            raise ValueError("local block has %d values, expected %d" % (local.size, counts[self.rank] * k))
        total = int(counts.sum())
        recv = DeviceArray(self.ctx, max(total * k, 1) * 8)
        off = int(counts[:self.rank].sum()) * k * 8
        if local.size:
            _lib.check(self.L.fpt_memcpy_h2d(self.ctx.h, recv.ptr + off, local.ctypes.data, local.size * 8))
        self.allgather_dev(recv.ptr + off, counts * k, recv.ptr)  # in place: the slice sits where it belongs
        self.ctx.synchronize()
        out = recv.download(np.float64, total * k).reshape(total, k)
        recv.free()
        return out


def sharded_deviation_stats(intervals, read_func, fasta_func, bm, dm, gather=None, rank=None, world=None,
                            batch_size=4096, stats_cls=None, **kw):
    """`detect.deviation_stats` over the ranks of a job (the shape of cli/detect.py:380-411 with
    GPUs for workers): the interval list is cut into contiguous ranges balanced by padded bases,
    this rank computes the records of its range, the per-base statistics are gathered, and every
    rank returns the records of ALL intervals in list order (rank 0 writes them).

    gather(local_matrix, counts) -> full matrix; default: TrackComm(ctx).allgather_rows.
    stats_cls: the per-rank driver (default detect.deviation_stats; tests pass a CPU stand-in).  The
    null draws of the FDR pass are keyed by global base index, so the records do not depend on
    the number of ranks."""
    if stats_cls is None:
        from .detect import deviation_stats as stats_cls

    r, w, _ = rank_info()
    rank = r if rank is None else int(rank)
    world = w if world is None else int(world)
    ds = stats_cls(intervals, read_func, fasta_func, bm, dm, batch_size=batch_size, **kw)
    lens = np.array([iv.end - iv.start for iv in ds.intervals], dtype=np.int64)
    bounds = shard_intervals(lens, world, ds.padding)
    a, b = bounds[rank]
    blocks = []
    for s in range(a, b, int(batch_size)):
        blocks += [rec["stats"] for rec in ds.compute(range(s, min(s + int(batch_size), b)))]
    ncol = 5 if dm else 2
    local = np.concatenate(blocks) if blocks else np.zeros((0, ncol))
    counts = shard_track_sizes(lens, bounds)
    if gather is None:
        comm = TrackComm(ds._scanner().ctx, rank, world)
        full = comm.allgather_rows(local, counts)
        comm.close()
    else:
        full = gather(local, counts)
    off = np.concatenate([[0], np.cumsum(lens)])
    return [{"interval": iv, "stats": full[off[i]:off[i + 1]]} for i, iv in enumerate(ds.intervals)]
