"""Diagnostic (GPU box): how many tiles does the first pass hand to the general kernel on sparse
counts (Poisson, the shape of real DNase-I data away from hotspots) and what does a step cost?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from footprint_tools_amd import _lib
from footprint_tools_amd.scan import DeviceArray, FootprintScanner

table, DM = bench.load_models()
ctx = _lib.Context(0)
n_iv, L, scales = 20000, 1000, (3, 5, 10, 20, 40)
S = len(scales)
sc = FootprintScanner(table, DM, bench.HW, bench.SHW, bench.CLIP, scales, ctx=ctx, nb_mode="memo")
l = sc.padded_len(L)
total = n_iv * L
t8 = total * 8
rs = np.random.RandomState(1)
d_cp, d_cm, d_sq = DeviceArray(ctx, n_iv * l * 8), DeviceArray(ctx, n_iv * l * 8), DeviceArray(ctx, n_iv * (l + 6))
d_out = DeviceArray(ctx, (3 + S) * t8)
sc.synth_dev(1, n_iv, L, d_cp.ptr, d_cm.ptr, d_sq.ptr)  # (sequence; the counts are replaced below)
for lam in (0.02, 0.1, 0.5, 2.0, "uniform 0..19"):
    if isinstance(lam, float):
        d_cp.upload(rs.poisson(lam, n_iv * l).astype(np.float64))
        d_cm.upload(rs.poisson(lam, n_iv * l).astype(np.float64))
    else:
        sc.synth_dev(1, n_iv, L, d_cp.ptr, d_cm.ptr, d_sq.ptr)

    def step():
        sc.scan_dev(n_iv, d_cp.ptr, d_cm.ptr, d_sq.ptr, d_out.ptr, d_out.ptr + t8, d_out.ptr + 2 * t8, d_out.ptr + 3 * t8,
                    interval_len=L)
    step(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 5
    tiles, redone, miss = ctx.scan_stats()
    print("counts %s: %.3f ms per %d bases = %.3g bases/s; tiles redone %d of %d (%.2f %%)"
          % (lam, dt * 1e3, total, total / dt, redone, tiles, 100.0 * redone / max(tiles, 1)))
    # the empirical FDR of the same batch (100 draws per base)
    d_ef = DeviceArray(ctx, t8)

    def fstep():
        sc.fdr_dev(n_iv, d_out.ptr, d_out.ptr + 3 * t8, d_ef.ptr, times=100, seed=1, half_win_width=3, interval_len=L,
                   obs=d_out.ptr + t8)
    fstep(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fstep()
    ctx.synchronize()
    dtf = (time.perf_counter() - t0) / 3
    print("     FDR with 100 draws: %.2f ms = %.3g bases/s" % (dtf * 1e3, total / dtf))
    d_ef.free()
