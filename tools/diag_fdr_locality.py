"""How much of the empirical-FDR draws' time is the locality of the sampler's table gathers: the 100-draw ragged
call (tools/bench_fdr_ragged.py's shape) with the expected counts (a) as the scan made them, (b) all set to their
median (one table row: every gather of a wavefront in a few lines), (c) spread uniformly over 0 .. 40 (every lane
another row).  Diagnostic."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import _lib  # noqa: E402
from footprint_tools_amd.scan import DeviceArray, FootprintScanner  # noqa: E402

n_iv = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
times = int(sys.argv[2]) if len(sys.argv) > 2 else 100
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
exp = d_out.download(np.float64, total)
vals, cnt = np.unique(exp, return_counts=True)
print("expected counts of the scan: %d distinct values, min %g max %g median %g; the ten most frequent: %s"
      % (vals.size, vals.min(), vals.max(), np.median(exp), ", ".join("%g (%.1f%%)" % (v, 100.0 * c / total) for v, c in
                                                                         sorted(zip(vals, cnt), key=lambda x: -x[1])[:10])))
per_iv = np.array([np.unique(exp[off[i]:off[i + 1]]).size for i in range(0, n_iv, 97)])
print("distinct expected counts per interval: mean %.1f max %d" % (per_iv.mean(), per_iv.max()))
d_e = DeviceArray(ctx, t8)
for name, e in (("as scanned", exp), ("all the median", np.full(total, np.median(exp))),
                ("uniform over 0..40", rs.randint(0, 41, total).astype(np.float64)),
                ("gamma(2, 3) rounded", np.round(rs.gamma(2.0, 3.0, total)))):
    d_e.upload(np.ascontiguousarray(e))

    def step():
        sc.fdr_dev(n_iv, d_e.ptr, d_out.ptr + 3 * t8, d_out.ptr + 4 * t8, times=times, seed=1, interval_off_dev=d_off.ptr,
                   interval_off_host=off)
    step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-22s times=%d: %.2f ms" % (name, times, dt * 1e3))
