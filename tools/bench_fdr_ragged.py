"""Timing of the empirical-FDR pass alone on whole-genome-shaped input (ragged, lognormal lengths,
mean ~162 bases) for several numbers of null draws per base: the intercept is what an interval costs
before its first draw (sort of the observed values, thresholds, rank guide), the slope is a draw.
Diagnostic; one line per `times`."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import _lib  # noqa: E402
from footprint_tools_amd.scan import DeviceArray, FootprintScanner  # noqa: E402

n_iv = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
NO_OBS = len(sys.argv) > 3 and sys.argv[3] == "noobs"  # the observed windows taken from the p-value track
NO_HOST_OFF = len(sys.argv) > 3 and sys.argv[3] == "nohostoff"  # the offsets fetched back from the device by the call
g = np.load("tests/golden/kmer_probs.npz")
lat = np.load("tests/golden/nb_lattice.npz")


class DM(object):
    mu_params, r_params = lat["mu_A"], lat["r_A"]


rs = np.random.RandomState(4)
lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 2000).astype(np.int64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
total = int(off[-1])
ctx = _lib.get_ctx()
sc = FootprintScanner(g["table"], DM, 5, 50, 0.01, (3,), nb_mode="memo")
n_c, n_s = sc.input_sizes(n_iv, total)
d_cp, d_cm, d_sq = DeviceArray(ctx, n_c * 8), DeviceArray(ctx, n_c * 8), DeviceArray(ctx, n_s)
_lib.check(ctx.L.fpt_synth_dev(ctx.h, 1, 0, n_c, d_cp.ptr, d_cm.ptr, 0, n_s, d_sq.ptr))
d_off = DeviceArray(ctx, off.nbytes).upload(off)
d_out = DeviceArray(ctx, 5 * total * 8)
t8 = total * 8
sc.scan_dev(n_iv, d_cp.ptr, d_cm.ptr, d_sq.ptr, d_out.ptr, d_out.ptr + t8, d_out.ptr + 2 * t8, d_out.ptr + 3 * t8,
            interval_off_dev=d_off.ptr, interval_off_host=off)
ctx.synchronize()
for times in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (4, 52, 100)):
    def step():
        sc.fdr_dev(n_iv, d_out.ptr, d_out.ptr + 3 * t8, d_out.ptr + 4 * t8, times=times, seed=1, interval_off_dev=d_off.ptr,
                   obs=None if NO_OBS else d_out.ptr + t8, interval_off_host=None if NO_HOST_OFF else off)
    step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("fdr ragged: %d intervals, %d bases, times=%d: %.2f ms -> %.3g bases/s, %.3g draws/s"
          % (n_iv, total, times, dt * 1e3, total / dt, total * times / dt))
