"""Batched, HBM-resident footprint scan: many intervals per launch.

The reference runs `prediction.compute -> strand merge -> dm.p_values -> stouffers_z` one
interval at a time (cli/detect.py:120-130).  `FootprintScanner` runs the same sequence for a
whole batch of intervals in one fused HIP kernel (fpt_scan_dev of include/fpt.h) and keeps
inputs and output tracks resident in HBM.

Layout (pad = half_win_width + smoothing_half_win_width), interval i of length L_i, output
offset off_i = sum_{j<i} L_j:
    counts_plus / counts_minus : float64, interval i at [off_i + i*(2*pad+1), +L_i+2*pad+1)
    seq                        : uint8 ASCII, interval i at [off_i + i*(2*pad+7), +L_i+2*pad+7)
    exp, obs, pval             : float64 tracks of sum(L_i)
    winp                       : (n_scales, sum(L_i)) float64
which is what prediction.compute fetches per interval (predict.pyx:132-140), back to back.
"""
import ctypes as C

import numpy as np

from . import _lib


class DeviceArray(object):
    """A device allocation made through the C ABI (for hosts without their own allocator).

    Allocations of up to 256 MiB come from, and go back to, a free list the context keeps (sizes
    rounded up to a power of two, at most 8 GiB held): the file-to-file driver makes two dozen
    arrays per batch, and hipMalloc / hipFree synchronise the device -- 26 frees per batch were 8 %
    of its time.  Everything runs on the context's one stream, so a buffer handed out again is
    only touched after the work that used it before."""

    _POOL_MAX_ITEM, _POOL_MAX_TOTAL = 256 << 20, 8 << 30

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self._bucket = None
        if 0 < self.nbytes <= self._POOL_MAX_ITEM:
            b = 4096
            while b < self.nbytes:
                b <<= 1
            self._bucket = b
            free = ctx._dev_pool.get(b)
            if free:
                self.ptr = free.pop()
                ctx._dev_pool_bytes -= b
                return
        p = C.c_void_p()
        _lib.check(ctx.L.fpt_dev_alloc(ctx.h, self._bucket or self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, host):
        host = np.ascontiguousarray(host)
        assert host.nbytes <= self.nbytes
        _lib.check(self.ctx.L.fpt_memcpy_h2d(self.ctx.h, self.ptr, host.ctypes.data, host.nbytes))
        return self

    def zero(self):
        """all bytes 0, on the context's stream"""
        _lib.check(self.ctx.L.fpt_dev_zero(self.ctx.h, self.ptr, self.nbytes))
        return self

    def download(self, dtype, count, offset_bytes=0):
        out = np.empty(int(count), dtype=dtype)
        if int(offset_bytes) < 0 or int(offset_bytes) + out.nbytes > self.nbytes:
            raise ValueError("download outside the allocation")
        _lib.check(self.ctx.L.fpt_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr + int(offset_bytes),
                                             out.nbytes))
        return out

    def free(self):
        if self.ptr:
            ctx = self.ctx
            held = ctx._dev_pool_bytes
            if self._bucket and held + self._bucket <= self._POOL_MAX_TOTAL and ctx.h:
                ctx._dev_pool.setdefault(self._bucket, []).append(self.ptr)
                ctx._dev_pool_bytes = held + self._bucket
            else:
                ctx.L.fpt_dev_free(ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def shard_intervals(lengths, world_size, pad):
    """Contiguous interval ranges per rank, balanced by cumulative padded bases
    sum(L_i + 2*pad + 1).  `lengths` is an int (uniform count is then given by
    `lengths=(n, L)`) or an array of interval lengths.  Returns a list of (first, last)."""
    if isinstance(lengths, tuple):
        n, L = lengths
        cost = np.full(int(n), int(L) + 2 * pad + 1, dtype=np.int64)
    else:
        cost = np.asarray(lengths, dtype=np.int64) + 2 * pad + 1
    n = cost.size
    cum = np.concatenate([[0], np.cumsum(cost)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        i = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(max(i, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


class FootprintScanner(object):
    def __init__(self, bias_table, dispersion, half_win_width=5, smoothing_half_win_width=50,
                 smoothing_clip=0.01, scales=(3,), default_propensity=1e-6, ctx=None,
                 nb_mode="auto"):
        """bias_table: 4096 propensities in 2-bit order (bias_model.table());
        dispersion: object with mu_params / r_params (modeling.dispersion.dispersion_model)."""
        self.ctx = ctx or _lib.get_ctx()
        self.table = _lib.f64(bias_table)
        self.dflt = float(default_propensity)
        # one model, or a list of models selected per interval through `dm_ids`
        self.models = list(dispersion) if isinstance(dispersion, (list, tuple)) else [dispersion]
        if dispersion is None:  # expected / observed counts only (learn_dm has no model yet)
            if nb_mode != "none" or len(scales):
                raise ValueError('a scanner without a dispersion model needs nb_mode="none" and scales=()')
            self.mu = self.r = None
        else:
            self.mu = _lib.f64(self.models[0].mu_params).ravel()
            self.r = _lib.f64(self.models[0].r_params).ravel()
        self.hw, self.shw, self.clip = int(half_win_width), int(smoothing_half_win_width), float(smoothing_clip)
        self.scales = tuple(int(s) for s in scales)
        if len(self.scales) > _lib.MAX_SCALES:
            raise ValueError("at most %d scales" % _lib.MAX_SCALES)
        self.pad = self.hw + self.shw
        # how the per-base NB p-value is evaluated (fpt_nb_mode of include/fpt.h)
        self.nb_mode = {"auto": _lib.NB_AUTO, "direct": _lib.NB_DIRECT, "memo": _lib.NB_MEMO,
                        "none": _lib.NB_NONE}[nb_mode]

    # ---- geometry ------------------------------------------------------------------------
    def padded_len(self, L):
        return int(L) + 2 * self.pad + 1

    def input_sizes(self, n_intervals, total_bases):
        """(#count elements per strand, #sequence bytes) for a batch."""
        return (total_bases + n_intervals * (2 * self.pad + 1),
                total_bases + n_intervals * (2 * self.pad + 7))

    # ---- device-pointer level ----------------------------------------------------------
    def scan_dev(self, n_intervals, counts_plus, counts_minus, seq, exp_out=None, obs_out=None,
                 pval_out=None, winp_out=None, interval_len=None, interval_off_dev=None,
                 interval_off_host=None, status_out=None, dm_ids_dev=None):
        """Enqueue the fused scan on device pointers (ints); does not synchronise.
        dm_ids_dev: device int32[n_intervals] of indices into the scanner's model list."""
        ctx = self.ctx
        ctx.set_bias_table(self.table, self.dflt)
        d = _lib.ScanDesc()
        d.n_intervals = int(n_intervals)
        d.interval_len = int(interval_len or 0)
        d.interval_off = interval_off_dev
        self._keep = None
        if interval_off_host is not None:
            self._keep = np.ascontiguousarray(interval_off_host, dtype=np.int64)
            d.interval_off_host = self._keep.ctypes.data
        d.half_win_width, d.smoothing_half_win_width = self.hw, self.shw
        d.smoothing_clip = self.clip
        d.n_scales = len(self.scales)
        for i, s in enumerate(self.scales):
            d.scales[i] = s
        d.dm_id = self._model_slot()
        d.dm_ids, d.n_dm = dm_ids_dev, len(self.models)
        d.nb_mode = self.nb_mode
        d.counts_plus, d.counts_minus, d.seq = counts_plus, counts_minus, seq
        d.exp_out, d.obs_out, d.pval_out, d.winp_out = exp_out, obs_out, pval_out, winp_out
        d.status_out = status_out
        _lib.check(ctx.L.fpt_scan_dev(ctx.h, C.byref(d)))

    def _model_slot(self):
        ctx = self.ctx
        if self.mu is None:
            return 0
        if len(self.models) == 1:
            return ctx.dispersion_slot(self.mu, self.r)
        return ctx.dispersion_slots([(m.mu_params, m.r_params) for m in self.models])

    def last_kernel_ms(self):
        ms = C.c_float()
        _lib.check(self.ctx.L.fpt_last_scan_ms(self.ctx.h, C.byref(ms)))
        return ms.value

    # ---- numpy level ---------------------------------------------------------------------
    def _desc(self, n_intervals):
        """a scan descriptor with this scanner's geometry, model slots and scales"""
        d = _lib.ScanDesc()
        d.n_intervals = int(n_intervals)
        d.half_win_width, d.smoothing_half_win_width = self.hw, self.shw
        d.smoothing_clip = self.clip
        d.n_scales = len(self.scales)
        for i, s in enumerate(self.scales):
            d.scales[i] = s
        d.dm_id = self._model_slot()
        d.n_dm = len(self.models)
        d.nb_mode = self.nb_mode
        return d

    def scan(self, counts_plus, counts_minus, seq, interval_len=None, interval_off=None, dm_ids=None, pinned_out=False,
             chunk_bases=0, out=None):
        """Host arrays in, dict of host arrays out (exp, obs, pval, winp[S], status): the per-call form a drop-in
        caller uses (the reference's prediction.compute / dm.p_values / windowing.stouffers_z take and return numpy
        arrays).  One fpt_scan_host call: the batch travels in chunks through a three-stage pipeline (host-to-device
        copy, scan, device-to-host copy; two threads), the two directions of the link and the kernel overlapping.
        The arrays are used where they lie, pageable or page-locked (`ctx.pinned_empty`; outputs with
        pinned_out=True): both move at the link's rate.
        dm_ids: optional per-interval index into the scanner's list of dispersion models.
        out: the dict an earlier call of the same shape returned -- its arrays are written again instead of new ones
        being made (a fresh numpy array costs a page fault per 4 KiB when first written, a fresh page-locked one
        ~170 ms per GiB to allocate: a caller that loops over batches reuses them)."""
        ctx = self.ctx
        cp, cm = _lib.f64(counts_plus).ravel(), _lib.f64(counts_minus).ravel()
        if isinstance(seq, str):
            seq = seq.encode("ascii", "replace")
        sq = (np.frombuffer(bytes(seq), dtype=np.uint8) if isinstance(seq, (bytes, bytearray))
              else np.ascontiguousarray(seq, dtype=np.uint8).ravel())
        if interval_off is not None:
            off = np.ascontiguousarray(interval_off, dtype=np.int64)
            n_iv, total = off.size - 1, int(off[-1] - off[0])
        else:
            lp = self.padded_len(interval_len)
            if cp.size % lp:
                raise ValueError("counts length is not a multiple of the padded interval length")
            n_iv = cp.size // lp
            total = n_iv * int(interval_len)
            off = None
        n_c, n_s = self.input_sizes(n_iv, total)
        if cp.size != n_c or cm.size != n_c or sq.size != n_s:
            raise ValueError("input sizes do not match the batch layout (counts %d/%d, seq %d/%d)"
                             % (cp.size, n_c, sq.size, n_s))
        S = len(self.scales)
        ids = None
        if dm_ids is not None:
            ids = np.ascontiguousarray(dm_ids, dtype=np.int32)
            if ids.size != n_iv or (n_iv and (ids.min() < 0 or ids.max() >= len(self.models))):
                raise ValueError("dm_ids needs one valid model index per interval")
        empty = ctx.pinned_empty if pinned_out else (lambda shape, dtype=np.float64: np.empty(shape, dtype=dtype))
        if out is not None:
            flat, status = out.get("_flat"), out.get("status")
            if flat is None or status is None or flat.size != (3 + S) * total or status.size != n_iv:
                raise ValueError("`out` is not the result of a scan of this shape")
        else:
            flat = empty((3 + S) * total)
            status = empty(max(n_iv, 1), np.int32)[:n_iv]
            if not pinned_out:  # a new numpy array: its pages touched by a team of threads before the copies land in it
                _lib.check(ctx.L.fpt_host_prefault(flat.ctypes.data, flat.nbytes))
        status[:] = 0
        if n_iv and total:
            ctx.set_bias_table(self.table, self.dflt)
            d = self._desc(n_iv)
            d.interval_len = int(interval_len or 0)
            d.interval_off = _lib.ptr(off)
            d.interval_off_host = _lib.ptr(off)
            d.dm_ids = _lib.ptr(ids)
            d.counts_plus, d.counts_minus, d.seq = cp.ctypes.data, cm.ctypes.data, sq.ctypes.data
            d.exp_out, d.obs_out = flat.ctypes.data, flat.ctypes.data + total * 8
            d.pval_out = flat.ctypes.data + 2 * total * 8
            d.winp_out = flat.ctypes.data + 3 * total * 8 if S else None
            d.status_out = status.ctypes.data
            _lib.check(ctx.L.fpt_scan_host(ctx.h, C.byref(d), int(chunk_bases)))
        pval = flat[2 * total:3 * total]
        if self.nb_mode == _lib.NB_NONE:  # counts only: no p-values were computed
            pval = np.full(total, np.nan)
        return dict(exp=flat[:total], obs=flat[total:2 * total], pval=pval,
                    winp=flat[3 * total:].reshape(S, total), status=status, _flat=flat)

    # ---- empirical FDR (cli/detect.py:132-135) ------------------------------------------
    def fdr_dev(self, n_intervals, exp, winp, efdr_out, times=100, seed=0, half_win_width=3,
                interval_len=None, interval_off_dev=None, base_index0=0, null_uniform=None, dm_ids_dev=None,
                null_winp_out=None, obs=None, interval_off_host=None):
        """Enqueue the null sampling + ranking on device pointers; does not synchronise.
        obs: the observed counts track (device pointer): ties between null and observed windows
        are then decided exactly, as in the reference (see fpt_fdr_desc.obs).  interval_off_host:
        the offsets once more as a host int64 array (saves their way back from the device)."""
        ctx = self.ctx
        d = _lib.FdrDesc()
        d.n_intervals = int(n_intervals)
        d.interval_len = int(interval_len or 0)
        d.interval_off = interval_off_dev
        d.base_index0 = int(base_index0)
        d.half_win_width, d.times, d.seed = int(half_win_width), int(times), int(seed)
        d.dm_id = self._model_slot()
        d.dm_ids, d.n_dm = dm_ids_dev, len(self.models)
        d.exp, d.winp, d.efdr_out, d.null_uniform = exp, winp, efdr_out, null_uniform
        d.null_winp_out = null_winp_out
        d.obs = obs
        off_h = None
        if interval_off_host is not None and interval_off_dev is not None:
            off_h = np.ascontiguousarray(interval_off_host, dtype=np.int64)
            if off_h.size != int(n_intervals) + 1:
                raise ValueError("interval_off_host must hold n_intervals + 1 offsets")
            d.interval_off_host = off_h.ctypes.data
        _lib.check(ctx.L.fpt_fdr_dev(ctx.h, C.byref(d)))  # (off_h is read before the call returns)

    def fdr(self, exp, winp, times=100, seed=0, half_win_width=3, interval_len=None, interval_off=None,
            base_index0=0, null_uniform=None, dm_ids=None, return_null=False, obs=None, host_offsets=True):
        """Empirical FDR of observed window p-values (host arrays in / out).  return_null=True
        also returns the (total_bases, times) null window p-values (detect.py:133).  obs: the
        observed counts the p-values were made from (exact ties, see fdr_dev)."""
        ctx = self.ctx
        exp, winp = _lib.f64(exp).ravel(), _lib.f64(winp).ravel()
        total = exp.size
        if interval_off is not None:
            off = np.ascontiguousarray(interval_off, dtype=np.int64)
            n_iv = off.size - 1
        else:
            off, n_iv = None, total // int(interval_len)
        bufs = []
        try:
            d_e = DeviceArray(ctx, max(exp.nbytes, 16)).upload(exp); bufs.append(d_e)
            d_w = DeviceArray(ctx, max(winp.nbytes, 16)).upload(winp); bufs.append(d_w)
            d_o = DeviceArray(ctx, max(total * 8, 16)); bufs.append(d_o)
            d_off = d_u = None
            if off is not None:
                d_off = DeviceArray(ctx, off.nbytes).upload(off); bufs.append(d_off)
            if null_uniform is not None:
                nu = _lib.f64(null_uniform).ravel()
                if nu.size != total * times:
                    raise ValueError("null_uniform needs total_bases * times values")
                d_u = DeviceArray(ctx, nu.nbytes).upload(nu); bufs.append(d_u)
            d_dm = d_n = d_obs = None
            if obs is not None:
                ob = _lib.f64(obs).ravel()
                if ob.size != total:
                    raise ValueError("obs needs one value per base")
                d_obs = DeviceArray(ctx, max(ob.nbytes, 16)).upload(ob); bufs.append(d_obs)
            if dm_ids is not None:
                ids = np.ascontiguousarray(dm_ids, dtype=np.int32)
                if ids.size != n_iv or (ids.size and (ids.min() < 0 or ids.max() >= len(self.models))):
                    raise ValueError("dm_ids needs one valid model index per interval")
                d_dm = DeviceArray(ctx, max(ids.nbytes, 16)).upload(ids); bufs.append(d_dm)
            if return_null:
                d_n = DeviceArray(ctx, max(total * times * 8, 16)); bufs.append(d_n)
            self.fdr_dev(n_iv, d_e.ptr, d_w.ptr, d_o.ptr, times, seed, half_win_width,
                         interval_len=interval_len, interval_off_dev=d_off.ptr if d_off else None,
                         base_index0=base_index0, null_uniform=d_u.ptr if d_u else None,
                         dm_ids_dev=d_dm.ptr if d_dm else None, null_winp_out=d_n.ptr if d_n else None,
                         obs=d_obs.ptr if d_obs else None,
                         interval_off_host=off if host_offsets else None)  # (False: the call fetches them back itself)
            ctx.synchronize()
            ef = d_o.download(np.float64, total)
            if d_n:
                return ef, d_n.download(np.float64, total * times).reshape(total, times)
            return ef
        finally:
            for b in bufs:
                b.free()

    # ---- (exp, obs) histogram of `ftd learn_dm` (cli/learn_dm.py:276-287) --------------
    def histogram(self, exp, obs, dims=(200, 1000)):
        """hist[int(exp), int(obs)] += 1 over host arrays; pairs outside `dims` are ignored."""
        ctx = self.ctx
        exp, obs = _lib.f64(exp).ravel(), _lib.f64(obs).ravel()
        if exp.size != obs.size:
            raise ValueError("exp and obs differ in length")
        rows, cols = int(dims[0]), int(dims[1])
        bufs = []
        try:
            d_e = DeviceArray(ctx, max(exp.nbytes, 16)).upload(exp); bufs.append(d_e)
            d_o = DeviceArray(ctx, max(obs.nbytes, 16)).upload(obs); bufs.append(d_o)
            d_h = DeviceArray(ctx, rows * cols * 8).upload(np.zeros(rows * cols, np.uint64)); bufs.append(d_h)
            _lib.check(ctx.L.fpt_hist2d_dev(ctx.h, d_e.ptr, d_o.ptr, exp.size, rows, cols, d_h.ptr))
            ctx.synchronize()
            return d_h.download(np.uint64, rows * cols).reshape(rows, cols).astype(np.int64)
        finally:
            for b in bufs:
                b.free()

    # ---- footprint calling: utils.segment over a whole track (cli/utils.py:204) -----------
    def segment_dev(self, track, n_intervals, threshold, w=3, decreasing=True, interval_len=None,
                    interval_off_dev=None):
        """Segments of a DEVICE track (pointer) -> dict of host arrays interval / start / end /
        score, ordered by interval and position; start / end are relative to the interval."""
        ctx = self.ctx
        d = _lib.SegmentDesc()
        d.n_intervals, d.interval_len, d.interval_off = int(n_intervals), int(interval_len or 0), interval_off_dev
        d.track, d.threshold, d.w, d.decreasing = track, float(threshold), int(w), int(bool(decreasing))
        total = C.c_int64()
        _lib.check(ctx.L.fpt_segment_count_dev(ctx.h, C.byref(d), C.byref(total)))
        n = total.value
        out = dict(interval=np.empty(0, np.int32), start=np.empty(0, np.int32), end=np.empty(0, np.int32),
                   score=np.empty(0, np.float64))
        if n == 0:
            return out
        buf = DeviceArray(ctx, n * 20)
        try:
            _lib.check(ctx.L.fpt_segment_fill_dev(ctx.h, C.byref(d), n, buf.ptr + 8 * n, buf.ptr + 12 * n,
                                                  buf.ptr + 16 * n, buf.ptr))
            ctx.synchronize()
            out["score"] = buf.download(np.float64, n)
            out["interval"] = buf.download(np.int32, n, 8 * n)
            out["start"] = buf.download(np.int32, n, 12 * n)
            out["end"] = buf.download(np.int32, n, 16 * n)
        finally:
            buf.free()
        return out

    def segment(self, track, threshold, w=3, decreasing=True, interval_len=None, interval_off=None):
        """`utils.segment` of every interval of a host track in one launch pair."""
        ctx = self.ctx
        track = _lib.f64(track).ravel()
        bufs = [DeviceArray(ctx, max(track.nbytes, 16)).upload(track)]
        try:
            d_off = None
            if interval_off is not None:
                off = np.ascontiguousarray(interval_off, dtype=np.int64)
                d_off = DeviceArray(ctx, off.nbytes).upload(off)
                bufs.append(d_off)
                n_iv = off.size - 1
            else:
                n_iv = track.size // int(interval_len)
            return self.segment_dev(bufs[0].ptr, n_iv, threshold, w, decreasing, interval_len=interval_len,
                                    interval_off_dev=d_off.ptr if d_off else None)
        finally:
            for b in bufs:
                b.free()

    # ---- synthetic workload (BASELINE.json configs 1-3) ---------------------------------
    def synth_dev(self, seed, n_intervals, interval_len, counts_plus, counts_minus, seq,
                  first_interval=0):
        """Fill device buffers with the counter-hash workload for intervals
        [first_interval, first_interval + n_intervals) of a uniform batch."""
        lp = self.padded_len(interval_len)
        _lib.check(self.ctx.L.fpt_synth_dev(self.ctx.h, int(seed), first_interval * lp, n_intervals * lp,
                                            counts_plus, counts_minus, first_interval * (lp + 6),
                                            n_intervals * (lp + 6), seq))

    def synth_hotspots_dev(self, seed, n_intervals, interval_len, counts_plus, counts_minus, per_mille,
                           first_interval=0):
        """Heavy-tailed variant of the synthetic workload: add hotspot bursts (peaks of 100..499
        cuts per strand in `per_mille`/1000 of the intervals) to counts made by synth_dev."""
        lp = self.padded_len(interval_len)
        _lib.check(self.ctx.L.fpt_synth_hotspots_dev(self.ctx.h, int(seed), first_interval * lp, n_intervals * lp,
                                                     lp, int(per_mille), counts_plus, counts_minus))

    def checksum_dev(self, ptr, n):
        out = C.c_uint64()
        _lib.check(self.ctx.L.fpt_checksum_dev(self.ctx.h, ptr, int(n), C.byref(out)))
        return out.value
