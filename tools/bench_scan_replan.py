"""What planning a ragged scan costs (diagnostic): the whole-genome shape scanned with the same offsets
again and again (the tile table of the call before is reused) and with two alternating offset arrays
(every call plans: what a job whose batches differ pays)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import _lib  # noqa: E402
from footprint_tools_amd.scan import DeviceArray, FootprintScanner  # noqa: E402

n_iv = int(sys.argv[1]) if len(sys.argv) > 1 else 437500
g, lat = np.load("tests/golden/kmer_probs.npz"), np.load("tests/golden/nb_lattice.npz")


class DM(object):
    mu_params, r_params = lat["mu_A"], lat["r_A"]


rs = np.random.RandomState(4)
lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 2000).astype(np.int64)
lens2 = lens.copy()
lens2[0], lens2[1] = lens[1] + 1, lens[0] - 1  # same total, another table
offs = [np.concatenate([[0], np.cumsum(x)]).astype(np.int64) for x in (lens, lens2)]
total = int(offs[0][-1])
ctx = _lib.get_ctx()
sc = FootprintScanner(g["table"], DM, 5, 50, 0.01, (3,), nb_mode="memo")
n_c, n_s = sc.input_sizes(n_iv, total)
d_cp, d_cm, d_sq = DeviceArray(ctx, n_c * 8), DeviceArray(ctx, n_c * 8), DeviceArray(ctx, n_s)
_lib.check(ctx.L.fpt_synth_dev(ctx.h, 1, 0, n_c, d_cp.ptr, d_cm.ptr, 0, n_s, d_sq.ptr))
d_offs = [DeviceArray(ctx, o.nbytes).upload(o) for o in offs]
d_out = DeviceArray(ctx, 4 * total * 8)
t8 = total * 8


def step(k):
    sc.scan_dev(n_iv, d_cp.ptr, d_cm.ptr, d_sq.ptr, d_out.ptr, d_out.ptr + t8, d_out.ptr + 2 * t8, d_out.ptr + 3 * t8,
                interval_off_dev=d_offs[k].ptr, interval_off_host=offs[k])


for name, seq in (("same offsets", [0] * 10), ("alternating offsets", [0, 1] * 5)):
    step(seq[-1] ^ 1 if name.startswith("alt") else 0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in seq:
        step(k)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / len(seq)
    print("%-20s %d intervals, %d bases: %.2f ms per call -> %.3g bases/s" % (name + ":", n_iv, total, dt * 1e3, total / dt))
