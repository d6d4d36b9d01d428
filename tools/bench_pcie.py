"""PCIe-inclusive rate of the host-array entry point (FootprintScanner.scan: H2D of counts and
sequence, fused scan, D2H of exp/obs/p/winp) on config-2-shaped input; diagnostic, one line."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc  # noqa: E402  (input generator only)
orc.lib()
from footprint_tools_amd.scan import FootprintScanner  # noqa: E402

n_iv, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 500
g = np.load("tests/golden/kmer_probs.npz")
lat = np.load("tests/golden/nb_lattice.npz")


class DM(object):
    mu_params, r_params = lat["mu_A"], lat["r_A"]


sc = FootprintScanner(g["table"], DM, 5, 50, 0.01, (3,))
l = sc.padded_len(L)
cp, cm = orc.synth_counts(1, 0, n_iv * l, 0), orc.synth_counts(1, 0, n_iv * l, 1)
sq = orc.synth_bases(1, 0, n_iv * (l + 6))
sc.scan(cp[:l * 10], cm[:l * 10], sq[:(l + 6) * 10], interval_len=L)
t0 = time.perf_counter()
out = sc.scan(cp, cm, sq, interval_len=L)
dt = time.perf_counter() - t0
nbytes = cp.nbytes + cm.nbytes + sq.nbytes + 4 * n_iv * L * 8
print("host arrays -> scan -> host arrays: %d bases in %.1f ms -> %.3g bases/s (%.1f GB/s over the link, pageable numpy buffers)"
      % (n_iv * L, dt * 1e3, n_iv * L / dt, nbytes / dt / 1e9))
