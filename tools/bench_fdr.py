"""Timing of the full `detect` statistics (scan + empirical FDR with 100 null draws per base) on
BASELINE config-2-shaped input; diagnostic, prints one line."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import _lib  # noqa: E402
from footprint_tools_amd.scan import DeviceArray, FootprintScanner  # noqa: E402

n_iv, L, times = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 500, 100
g = np.load("tests/golden/kmer_probs.npz")
lat = np.load("tests/golden/nb_lattice.npz")


class DM(object):
    mu_params, r_params = lat["mu_A"], lat["r_A"]


ctx = _lib.get_ctx()
sc = FootprintScanner(g["table"], DM, 5, 50, 0.01, (3,), nb_mode="memo")
l = sc.padded_len(L)
tot = n_iv * L
d_cp, d_cm, d_sq = DeviceArray(ctx, n_iv * l * 8), DeviceArray(ctx, n_iv * l * 8), DeviceArray(ctx, n_iv * (l + 6))
d_out = DeviceArray(ctx, 5 * tot * 8)
sc.synth_dev(1, n_iv, L, d_cp.ptr, d_cm.ptr, d_sq.ptr)
t8 = tot * 8


def step():
    sc.scan_dev(n_iv, d_cp.ptr, d_cm.ptr, d_sq.ptr, d_out.ptr, d_out.ptr + t8, d_out.ptr + 2 * t8,
                d_out.ptr + 3 * t8, interval_len=L)
    sc.fdr_dev(n_iv, d_out.ptr, d_out.ptr + 3 * t8, d_out.ptr + 4 * t8, times=times, seed=1, interval_len=L)


step()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
ctx.synchronize()
dt = (time.perf_counter() - t0) / 3
ef = d_out.download(np.float64, 2000, 4 * t8)
print("scan+fdr(times=%d): %d bases in %.2f ms -> %.3g bases/s (%.3g null draws/s); efdr[:5]=%s"
      % (times, tot, dt * 1e3, tot / dt, tot * times / dt, np.round(ef[:5], 4)))
