"""Diagnostic (GPU box): a DHS-like batch -- sparse background (Poisson 0.05 per base and strand), a
150-base hotspot of Poisson(10) counts in 30 % of the intervals, one with Poisson(150): tiles handed
to the general kernel and the rate, first call and later calls (the second-level table is kept)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from footprint_tools_amd import _lib
from footprint_tools_amd.scan import DeviceArray, FootprintScanner

table, DM = bench.load_models()
ctx = _lib.Context(0)
n_iv, L, scales = 20000, 1000, (3,)
S = len(scales)
sc = FootprintScanner(table, DM, bench.HW, bench.SHW, bench.CLIP, scales, ctx=ctx, nb_mode="memo")
l = sc.padded_len(L)
total = n_iv * L
t8 = total * 8
rs = np.random.RandomState(1)
d_cp, d_cm, d_sq = DeviceArray(ctx, n_iv * l * 8), DeviceArray(ctx, n_iv * l * 8), DeviceArray(ctx, n_iv * (l + 6))
d_out = DeviceArray(ctx, (3 + S) * t8)
sc.synth_dev(1, n_iv, L, d_cp.ptr, d_cm.ptr, d_sq.ptr)
cp, cm = (rs.poisson(0.05, (n_iv, l)).astype(np.float64) for _ in range(2))
hot = np.nonzero(rs.rand(n_iv) < 0.3)[0]
for i in hot:
    a = int(rs.randint(100, l - 250))
    lam = 150.0 if rs.rand() < 0.03 else 10.0
    cp[i, a:a + 150] += rs.poisson(lam, 150)
    cm[i, a:a + 150] += rs.poisson(lam, 150)
d_cp.upload(cp.ravel()); d_cm.upload(cm.ravel())


def step():
    sc.scan_dev(n_iv, d_cp.ptr, d_cm.ptr, d_sq.ptr, d_out.ptr, d_out.ptr + t8, d_out.ptr + 2 * t8, d_out.ptr + 3 * t8,
                interval_len=L)


ctx.drop_kept_tables()
for k in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    tiles, redone, miss = ctx.scan_stats()
    print("call %d: %.3f ms = %.3g bases/s; tiles redone %d of %d (%.2f %%), largest pair outside the tables %s"
          % (k, dt * 1e3, total / dt, redone, tiles, 100.0 * redone / tiles, miss))
