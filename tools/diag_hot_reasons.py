"""Diagnostic (GPU box): what do the tiles that the first pass flags in the heavy-tailed workload
look like -- window arguments below -26 (winp < 1e-148) or table misses only?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from footprint_tools_amd import _lib
from footprint_tools_amd.scan import DeviceArray, FootprintScanner

table, DM = bench.load_models()
ctx = _lib.Context(0)
n_iv, L, scales, pm = 20000, 1000, (3, 5, 10, 20, 40), 20
S = len(scales)
sc = FootprintScanner(table, DM, bench.HW, bench.SHW, bench.CLIP, scales, nb_mode="memo")
l = sc.padded_len(L)
total = n_iv * L
t8 = total * 8
d_cp, d_cm, d_sq = DeviceArray(ctx, n_iv * l * 8), DeviceArray(ctx, n_iv * l * 8), DeviceArray(ctx, n_iv * (l + 6))
d_out = DeviceArray(ctx, (3 + S) * t8)
sc.synth_dev(1, n_iv, L, d_cp.ptr, d_cm.ptr, d_sq.ptr)
sc.synth_hotspots_dev(1, n_iv, L, d_cp.ptr, d_cm.ptr, pm)
sc.scan_dev(n_iv, d_cp.ptr, d_cm.ptr, d_sq.ptr, d_out.ptr, d_out.ptr + t8, d_out.ptr + 2 * t8, d_out.ptr + 3 * t8,
            interval_len=L)
print("scan_stats", ctx.scan_stats())
got = d_out.download(np.float64, (3 + S) * total).reshape(3 + S, n_iv, L)
e, o, p, wp = got[0], got[1], got[2], got[3:]
hot = (o.max(axis=1) > 60)
print("hot intervals", hot.sum())
low = (wp.min(axis=(0, 2)) < 1e-148)
print("of them with a window p below 1e-148:", (hot & low).sum(), " (not hot but low:", (~hot & low).sum(), ")")
print("min winp over hot tiles: quantiles", np.quantile(wp.min(axis=(0, 2))[hot], [0, .1, .5, .9, 1]))
print("max exp / obs in hot:", e[hot].max(), o[hot].max(), " min p:", p[hot].min())
