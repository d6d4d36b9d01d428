"""ctypes binding of libfpt_hip.so (C ABI: include/fpt.h).  No PyTorch, no cffi."""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FPT_LIB_PATH") or os.path.join(HERE, "libfpt_hip.so")

FPT_OK, FPT_ERR_INVALID, FPT_ERR_HIP, FPT_ERR_NODEVICE, FPT_ERR_ZERODIV, FPT_ERR_NOMEM = 0, -1, -2, -3, -4, -5
WIN_SUM, WIN_PRODUCT, WIN_FISHER, WIN_STOUFFER, WIN_WSTOUFFER = range(5)
NB_CDF, NB_LOGPMF, NB_PMF = range(3)
NB_AUTO, NB_DIRECT, NB_MEMO, NB_NONE = range(4)
FN = dict(gamma=0, lgam=1, ndtr=2, ndtri=3, log1p=4, erf=5, erfc=6, incbet=7, chdtrc=8, ndtr_window=9, ndtr_window_tab=10, log_fast=11,
          log1p_fast=12)
MAX_SCALES = 8
MAX_DM = 64

EXPORTS = [
    "fpt_last_error", "fpt_version", "fpt_device_count", "fpt_ctx_create", "fpt_ctx_destroy",
    "fpt_ctx_set_stream", "fpt_ctx_use_own_stream", "fpt_ctx_synchronize", "fpt_set_bias_table", "fpt_set_dispersion",
    "fpt_kmer_probs", "fpt_predict", "fpt_nb_values", "fpt_nb_scalar", "fpt_window", "fpt_special",
    "fpt_scan_dev", "fpt_scan_stats", "fpt_synth_dev", "fpt_synth_hotspots_dev", "fpt_checksum_dev", "fpt_dev_alloc", "fpt_dev_free",
    "fpt_dev_zero", "fpt_format_stats", "fpt_format_stats_batch", "fpt_memcpy_h2d", "fpt_memcpy_d2h", "fpt_last_scan_ms", "fpt_timing_enable", "fpt_timing_read", "fpt_mark", "fpt_mark_elapsed", "fpt_marks_clear",
    "fpt_bam_open", "fpt_bam_close", "fpt_bam_n_refs", "fpt_bam_ref", "fpt_bam_read", "fpt_bam_has_index", "fpt_bam_seek_region", "fpt_bam_read_raw", "fpt_cut_counts_dev", "fpt_seq_gather_dev",
    "fpt_track_open", "fpt_track_close", "fpt_track_n_refs", "fpt_track_ref", "fpt_track_fetch", "fpt_track_fetch_rows", "fpt_track_writer_open", "fpt_track_writer_set_level", "fpt_track_writer_write", "fpt_track_writer_write_stats", "fpt_track_writer_close",
    "fpt_comm_unique_id", "fpt_comm_init", "fpt_comm_destroy", "fpt_allgather_track", "fpt_gather_track",
    "fpt_allgather_track_async", "fpt_gather_track_async", "fpt_comm_wait", "fpt_comm_synchronize", "fpt_comm_info",
    "fpt_scan_host", "fpt_scan_host_last", "fpt_host_alloc", "fpt_host_free", "fpt_host_prefault",
    "fpt_stream_pattern_dev", "fpt_set_memo_dims", "fpt_drop_kept_tables", "fpt_fdr_dev", "fpt_posterior_dev", "fpt_detect_columns_dev", "fpt_hist2d_dev", "fpt_segment_count_dev", "fpt_segment_fill_dev",
]


class CommInfo(C.Structure):
    """struct fpt_comm_info_t of include/fpt.h"""
    _fields_ = [("world_size", C.c_int32), ("rank", C.c_int32), ("device", C.c_int32),
                ("rccl_count", C.c_int32), ("rccl_user_rank", C.c_int32), ("rccl_device", C.c_int32),
                ("pci_bus_id", C.c_char * 32)]


class ScanDesc(C.Structure):
    """struct fpt_scan_desc of include/fpt.h"""
    _fields_ = [
        ("n_intervals", C.c_int64),
        ("interval_len", C.c_int32),
        ("interval_off", C.c_void_p),
        ("interval_off_host", C.c_void_p),
        ("half_win_width", C.c_int32),
        ("smoothing_half_win_width", C.c_int32),
        ("smoothing_clip", C.c_double),
        ("n_scales", C.c_int32),
        ("scales", C.c_int32 * MAX_SCALES),
        ("dm_id", C.c_int32),
        ("dm_ids", C.c_void_p),
        ("n_dm", C.c_int32),
        ("nb_mode", C.c_int32),
        ("counts_plus", C.c_void_p),
        ("counts_minus", C.c_void_p),
        ("seq", C.c_void_p),
        ("exp_out", C.c_void_p),
        ("obs_out", C.c_void_p),
        ("pval_out", C.c_void_p),
        ("winp_out", C.c_void_p),
        ("status_out", C.c_void_p),
    ]


class ScanHostStats(C.Structure):
    """struct fpt_scan_host_stats of include/fpt.h"""
    _fields_ = [("seconds", C.c_double), ("bases", C.c_int64), ("chunks", C.c_int64), ("bytes_h2d", C.c_int64),
                ("bytes_d2h", C.c_int64), ("inputs_pinned", C.c_int32), ("outputs_pinned", C.c_int32),
                ("wait_seconds", C.c_double), ("stage_seconds", C.c_double), ("issue_seconds", C.c_double)]


class FdrDesc(C.Structure):
    """struct fpt_fdr_desc of include/fpt.h"""
    _fields_ = [
        ("n_intervals", C.c_int64),
        ("interval_len", C.c_int32),
        ("interval_off", C.c_void_p),
        ("base_index0", C.c_int64),
        ("half_win_width", C.c_int32),
        ("times", C.c_int32),
        ("seed", C.c_uint64),
        ("dm_id", C.c_int32),
        ("dm_ids", C.c_void_p),
        ("n_dm", C.c_int32),
        ("exp", C.c_void_p),
        ("winp", C.c_void_p),
        ("efdr_out", C.c_void_p),
        ("null_uniform", C.c_void_p),
        ("null_winp_out", C.c_void_p),
        ("obs", C.c_void_p),
        ("interval_off_host", C.c_void_p),
    ]


class PosteriorDesc(C.Structure):
    """struct fpt_posterior_desc of include/fpt.h"""
    _fields_ = [
        ("n_intervals", C.c_int64),
        ("interval_len", C.c_int32),
        ("interval_off", C.c_void_p),
        ("total_bases", C.c_int64),
        ("max_interval_len", C.c_int32),
        ("n_datasets", C.c_int32),
        ("dm_id", C.c_int32),
        ("half_win_width", C.c_int32),
        ("fdr_cutoff", C.c_double),
        ("pseudocount", C.c_double),
        ("betas", C.c_void_p),
        ("obs", C.c_void_p),
        ("exp", C.c_void_p),
        ("fdr", C.c_void_p),
        ("w", C.c_void_p),
        ("post_out", C.c_void_p),
        ("prior_out", C.c_void_p),
        ("delta_out", C.c_void_p),
        ("ll_on_out", C.c_void_p),
        ("ll_off_out", C.c_void_p),
        ("status_out", C.c_void_p),
        ("models", C.c_void_p),
    ]


class SegmentDesc(C.Structure):
    """struct fpt_segment_desc of include/fpt.h"""
    _fields_ = [
        ("n_intervals", C.c_int64),
        ("interval_len", C.c_int32),
        ("interval_off", C.c_void_p),
        ("track", C.c_void_p),
        ("threshold", C.c_double),
        ("w", C.c_int32),
        ("decreasing", C.c_int32),
    ]


_lib = None
_lock = threading.RLock()


def load():
    """Load the HIP library; fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "footprint_tools_amd: %s is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C footprint_tools_amd/csrc` (needs hipcc; target gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
        L.fpt_last_error.restype = C.c_char_p
        L.fpt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.fpt_ctx_create.argtypes = [i32, C.POINTER(vp)]
        L.fpt_ctx_destroy.argtypes = [vp]
        L.fpt_ctx_set_stream.argtypes = [vp, vp]
        L.fpt_ctx_use_own_stream.argtypes = [vp]
        L.fpt_ctx_synchronize.argtypes = [vp]
        L.fpt_set_bias_table.argtypes = [vp, vp, dbl]
        L.fpt_set_dispersion.argtypes = [vp, i32, vp, vp]
        L.fpt_kmer_probs.argtypes = [vp, vp, i64, vp, vp]
        L.fpt_predict.argtypes = [vp, vp, vp, i64, i32, i32, i32, dbl, vp, vp]
        L.fpt_nb_values.argtypes = [vp, i32, i32, vp, vp, i64, vp]
        L.fpt_nb_scalar.argtypes = [vp, i32, vp, vp, vp, i64, vp]
        L.fpt_window.argtypes = [vp, i32, vp, vp, i64, i32, i32, vp]
        L.fpt_special.argtypes = [vp, i32, vp, vp, vp, i64, vp]
        L.fpt_scan_dev.argtypes = [vp, C.POINTER(ScanDesc)]
        L.fpt_fdr_dev.argtypes = [vp, C.POINTER(FdrDesc)]
        L.fpt_posterior_dev.argtypes = [vp, C.POINTER(PosteriorDesc)]
        L.fpt_seq_gather_dev.argtypes = [vp, vp, i64, vp, i64, vp]
        L.fpt_detect_columns_dev.argtypes = [vp, i64, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp]
        L.fpt_hist2d_dev.argtypes = [vp, vp, vp, i64, i32, i32, vp]
        L.fpt_segment_count_dev.argtypes = [vp, C.POINTER(SegmentDesc), C.POINTER(C.c_int64)]
        L.fpt_segment_fill_dev.argtypes = [vp, C.POINTER(SegmentDesc), i64, vp, vp, vp, vp]
        L.fpt_synth_dev.argtypes = [vp, C.c_uint64, i64, i64, vp, vp, i64, i64, vp]
        L.fpt_comm_unique_id.argtypes = [vp]
        L.fpt_comm_init.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        L.fpt_comm_destroy.argtypes = [vp]
        L.fpt_allgather_track.argtypes = [vp, vp, vp, vp, vp]
        if hasattr(L, "fpt_gather_track"):  # (absent from older builds loaded through FPT_LIB_PATH for A/B runs)
            L.fpt_gather_track.argtypes = [vp, vp, vp, vp, vp, i32]
            L.fpt_allgather_track_async.argtypes = [vp, vp, vp, vp, vp]
            L.fpt_gather_track_async.argtypes = [vp, vp, vp, vp, vp, i32]
            L.fpt_comm_wait.argtypes = [vp, vp, i32]
            L.fpt_comm_synchronize.argtypes = [vp]
        if hasattr(L, "fpt_comm_info"):
            L.fpt_comm_info.argtypes = [vp, C.POINTER(CommInfo)]
        L.fpt_scan_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32 * 2)]
        L.fpt_synth_hotspots_dev.argtypes = [vp, C.c_uint64, i64, i64, i32, i32, vp, vp]
        L.fpt_checksum_dev.argtypes = [vp, vp, i64, C.POINTER(C.c_uint64)]
        L.fpt_dev_alloc.argtypes = [vp, i64, C.POINTER(vp)]
        L.fpt_dev_free.argtypes = [vp, vp]
        L.fpt_dev_zero.argtypes = [vp, vp, i64]
        L.fpt_format_stats.argtypes = [C.c_char_p, i64, vp, i64, i32, vp, i64, C.c_char, i32, vp, i64, C.POINTER(i64)]
        L.fpt_format_stats_batch.argtypes = [i64, vp, i32, vp, vp, vp, vp, i32, C.c_char, i32, vp, i64, C.POINTER(i64)]
        L.fpt_track_writer_write_stats.argtypes = [vp, i64, vp, i32, vp, vp, vp, vp, i32, i32]
        L.fpt_memcpy_h2d.argtypes = [vp, vp, vp, i64]
        L.fpt_memcpy_d2h.argtypes = [vp, vp, vp, i64]
        L.fpt_last_scan_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.fpt_set_memo_dims.argtypes = [vp, i32, i32]
        L.fpt_drop_kept_tables.argtypes = [vp]
        L.fpt_timing_enable.argtypes = [vp, i32]
        L.fpt_scan_host.argtypes = [vp, vp, i64]
        L.fpt_scan_host_last.argtypes = [vp, vp]
        L.fpt_host_alloc.argtypes = [vp, i64, C.POINTER(C.c_void_p)]
        L.fpt_host_free.argtypes = [vp, vp]
        L.fpt_host_prefault.argtypes = [vp, i64]
        if hasattr(L, "fpt_stream_pattern_dev"):  # (absent from older builds loaded through FPT_LIB_PATH for A/B runs)
            L.fpt_stream_pattern_dev.argtypes = [vp, i64, i32, vp, i32, i32, i32, vp, vp, vp, vp, i64, i32, C.POINTER(C.c_float)]
        L.fpt_timing_read.argtypes = [vp, vp, i32, C.POINTER(C.c_int)]
        L.fpt_mark.argtypes = [vp, C.POINTER(C.c_int32)]
        L.fpt_mark_elapsed.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
        L.fpt_marks_clear.argtypes = [vp]
        _lib = L
    return _lib


class FptError(RuntimeError):
    pass


def check(rc):
    if rc == FPT_OK:
        return
    msg = load().fpt_last_error().decode("utf-8", "replace")
    if rc == FPT_ERR_ZERODIV:
        raise ZeroDivisionError(msg or "float division")
    if rc == FPT_ERR_INVALID:
        raise ValueError(msg)
    if rc == FPT_ERR_NOMEM:
        raise MemoryError(msg)
    raise FptError("libfpt_hip error %d: %s" % (rc, msg))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data


def batch_text_args(chroms, starts, row_off, table):
    """arguments of fpt_format_stats_batch / fpt_track_writer_write_stats from a batch's interval
    names, starts, row offsets and (rows, columns) matrix: (char*[] of the distinct names, ids,
    starts, offsets, matrix)"""
    uniq = {}
    ids = np.fromiter((uniq.setdefault(c, len(uniq)) for c in chroms), dtype=np.int32, count=len(chroms))
    for c in uniq:
        if not str(c).isascii():
            raise ValueError("chromosome name %r is not ASCII" % (c,))
    names = (C.c_char_p * max(len(uniq), 1))(*[str(c).encode() for c in uniq])
    st = np.ascontiguousarray(starts, dtype=np.int64)
    off = np.ascontiguousarray(row_off, dtype=np.int64)
    m = np.ascontiguousarray(table, dtype=np.float64)
    if m.ndim != 2 or off.size != st.size + 1 or ids.size != st.size or (off.size and (off[0] < 0 or off[-1] > m.shape[0])):
        raise ValueError("batch of %d intervals: offsets / starts / matrix do not fit together" % st.size)
    return names, ids, st, off, m


class Context(object):
    """One per (process, GPU): stream, bias table, dispersion-model slots, workspace."""

    def __init__(self, device=None):
        L = load()
        if device is None:
            device = int(os.environ.get("FPT_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = C.c_int(0)
            L.fpt_device_count(C.byref(n))
            if n.value > 0:
                device %= n.value
        h = C.c_void_p()
        check(L.fpt_ctx_create(int(device), C.byref(h)))
        self.L, self.h, self.device = L, h, int(device)
        self._table_key = None
        self._dm_slots = {}
        self._slot_key = {}  # slot -> the model it holds
        self._dm_lists = {}
        self._dm_next = 0
        self._lock = threading.RLock()
        self._dev_pool, self._dev_pool_bytes = {}, 0  # free list of scan.DeviceArray

    def close(self):
        if self.h:
            for ptrs in self._dev_pool.values():
                for p in ptrs:
                    self.L.fpt_dev_free(self.h, p)
            self._dev_pool, self._dev_pool_bytes = {}, 0
            self.L.fpt_ctx_destroy(self.h)
            self.h = None

    def trim_pool(self):
        """Give the device buffers on scan.DeviceArray's free list back to the driver (the list keeps up to
        8 GiB per context for reuse; a host that has finished its large batches calls this instead of
        closing the context)."""
        with self._lock:
            self.synchronize()
            for ptrs in self._dev_pool.values():
                for p in ptrs:
                    self.L.fpt_dev_free(self.h, p)
            self._dev_pool, self._dev_pool_bytes = {}, 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- model state -----------------------------------------------------------------
    def set_bias_table(self, table, dflt=1e-6):
        table = f64(table)
        if table.shape != (4096,):
            raise ValueError("bias table must have 4096 entries")
        key = (table.tobytes(), float(dflt))
        with self._lock:
            if key != self._table_key:
                check(self.L.fpt_set_bias_table(self.h, ptr(table), float(dflt)))
                self._table_key = key

    def dispersion_slot(self, mu_params, r_params):
        """Slot id holding this (mu_params, r_params); uploads on first use (LRU of 64 slots)."""
        mu, r = f64(mu_params).ravel(), f64(r_params).ravel()
        if mu.size != 9 or r.size != 15:
            raise ValueError("mu_params needs 9 and r_params 15 values")
        key = (mu.tobytes(), r.tobytes())
        with self._lock:
            slot = self._dm_slots.get(key)
            if slot is None or self._slot_key.get(slot) != key:
                slot = self._dm_next % MAX_DM
                self._dm_next += 1
                for k in [k for k, v in self._dm_slots.items() if v == slot]:
                    del self._dm_slots[k]
                for k in [k for k, f in self._dm_lists.items() if f <= slot < f + len(k)]:
                    del self._dm_lists[k]
                check(self.L.fpt_set_dispersion(self.h, slot, ptr(mu), ptr(r)))
                self._dm_slots[key] = slot
                self._slot_key[slot] = key
            return slot

    def dispersion_slots(self, models):
        """A list of (mu_params, r_params) in CONSECUTIVE slots (for per-interval models); returns
        the first slot.  The same list again returns the same slots without touching the device
        (an upload synchronises the stream, which scan_dev / fdr_dev promise not to do)."""
        n = len(models)
        if n < 1 or n > MAX_DM:
            raise ValueError("need 1..%d models" % MAX_DM)
        packed = [(f64(mu).ravel(), f64(r).ravel()) for mu, r in models]
        key = tuple((mu.tobytes(), r.tobytes()) for mu, r in packed)
        with self._lock:
            # a cached list is valid while every one of its slots still holds its model -- checked by what
            # the SLOTS hold (a list with the same model twice has two slots for one key: a map from model
            # to slot cannot vouch for it, and such a list was uploaded again at every call, overwriting
            # slots a live scanner used)
            hit = self._dm_lists.get(key)
            if hit is not None and all(self._slot_key.get(hit + i) == k for i, k in enumerate(key)):
                return hit
            first = 0 if self._dm_next % MAX_DM + n > MAX_DM else self._dm_next % MAX_DM
            for i, (mu, r) in enumerate(packed):
                slot = first + i
                for k in [k for k, v in self._dm_slots.items() if v == slot]:
                    del self._dm_slots[k]
                check(self.L.fpt_set_dispersion(self.h, slot, ptr(mu), ptr(r)))
                self._dm_slots[key[i]] = slot
                self._slot_key[slot] = key[i]
            self._dm_next = first + n
            # lists that covered an overwritten slot are dead (their check above would fail for ever): dropped,
            # so the cache holds at most what the 64 slots can vouch for
            for k in [k for k, f in self._dm_lists.items() if f < first + n and first < f + len(k)]:
                del self._dm_lists[k]
            self._dm_lists[key] = first
            return first

    def synchronize(self):
        check(self.L.fpt_ctx_synchronize(self.h))

    def pinned_empty(self, shape, dtype=np.float64):
        """An uninitialised numpy array in page-locked host memory (fpt_host_alloc): arrays of this kind are read
        and written by the copy engines directly when handed to FootprintScanner.scan.  Freed with the array."""
        import weakref
        dt = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        p = C.c_void_p()
        check(self.L.fpt_host_alloc(self.h, max(n, 16), C.byref(p)))
        buf = (C.c_char * max(n, 16)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        L, h, addr = self.L, self.h, p.value
        weakref.finalize(buf, lambda: L.fpt_host_free(h, C.c_void_p(addr)) if h else None)
        return arr

    def scan_host_last(self):
        st = ScanHostStats()
        check(self.L.fpt_scan_host_last(self.h, C.byref(st)))
        return dict((k, getattr(st, k)) for k, _ in ScanHostStats._fields_)

    def set_stream(self, hip_stream):
        """Run on a caller-owned hipStream_t (integer handle).  0 is the device's default (null)
        stream -- what `torch.cuda.current_stream().cuda_stream` is on the default stream -- and
        work is then ordered with it; None goes back to the context's own non-blocking stream.
        (The stream in use is drained first: buffers on scan.DeviceArray's free list are handed out
        again on the assumption that one stream orders everything.)"""
        self.synchronize()
        if hip_stream is None:
            check(self.L.fpt_ctx_use_own_stream(self.h))
        else:
            check(self.L.fpt_ctx_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def scan_stats(self):
        """(tiles, tiles redone by the general kernel, (max exp, max obs) that missed the first-level
        table) of the most recent memo-mode scan; synchronises."""
        t, r, m = C.c_int64(0), C.c_int64(0), (C.c_int32 * 2)(-1, -1)
        check(self.L.fpt_scan_stats(self.h, C.byref(t), C.byref(r), C.byref(m)))
        return t.value, r.value, (m[0], m[1])

    def drop_kept_tables(self):
        """Empty the second-level (exp, obs) table memo mode keeps across calls (see fpt.h)."""
        check(self.L.fpt_drop_kept_tables(self.h))

    def mark(self):
        """a HIP event on the context's stream; returns its number (see mark_elapsed)"""
        i = C.c_int32(-1)
        check(self.L.fpt_mark(self.h, C.byref(i)))
        return i.value

    def mark_elapsed(self, a, b):
        """milliseconds on the device between marks a and b (waits for b)"""
        ms = C.c_float(0.0)
        check(self.L.fpt_mark_elapsed(self.h, int(a), int(b), C.byref(ms)))
        return float(ms.value)

    def marks_clear(self):
        check(self.L.fpt_marks_clear(self.h))

    def timing_enable(self, max_records):
        check(self.L.fpt_timing_enable(self.h, int(max_records)))

    def timing_read(self, cap=100000):
        """(sequence_ms, dominant_pass_ms) arrays of the scans recorded since timing_enable /
        the last read (synchronises)."""
        buf = np.zeros(2 * cap, dtype=np.float32)
        n = C.c_int(0)
        check(self.L.fpt_timing_read(self.h, buf.ctypes.data, cap, C.byref(n)))
        m = min(n.value, cap)
        pairs = buf[:2 * m].astype(np.float64).reshape(m, 2)
        return pairs[:, 0], pairs[:, 1]


_default = None


def get_ctx():
    """Process-wide default context (created on first use)."""
    global _default
    if _default is None:
        with _lock:
            if _default is None:
                _default = Context()
    return _default


def gpu_available():
    try:
        n = C.c_int(0)
        load().fpt_device_count(C.byref(n))
        return n.value > 0
    except (ImportError, OSError):
        return False
