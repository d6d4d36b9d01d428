#!/usr/bin/env python3
"""Table form of g(t) = Phi(-t) exp(t^2/2) for fptm::ndtr_fast_tab (footprint_tools_amd/csrc/fpt_math.hpp).

g is smooth in x = 1/(t + 5); on [1/31, 1/5] (t in [0, 26]) it is cut into N intervals of equal width
in x, and on each a cubic in w = (position inside the interval, 0 <= w < 1) interpolates g at the four
Chebyshev nodes of the interval (60-digit arithmetic).  Entry k = (c3, c2, c1, c0):
g ~ ((c3 w + c2) w + c1) w + c0.  Three fused multiply-adds, a multiply-add for the slot, a
truncation and a fraction take the place of the 14-step Horner chain of ndtr_fast (32 bytes per
entry, 4 KB in LDS for N = 128).

    python tools/fit_ndtr_gtab.py [N] > footprint_tools_amd/csrc/fpt_ndtr_gtab.hpp
"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K, T = mp.mpf(5), mp.mpf(26)
x_lo, x_hi = 1 / (T + K), 1 / K
# the slot of x: kf = (x - x_lo) * scale, k = trunc(kf), w = kf - k; the upper end is widened by a hair so
# that x = 1/5 (t = 0) still falls into the last interval
scale = N / ((x_hi - x_lo) * (1 + mp.mpf(2) ** -40))
h = 1 / scale


def g(x):
    t = 1 / x - K
    return mp.ncdf(-t) * mp.exp(t * t / 2)


rows = []
for k in range(N):
    a = x_lo + k * h
    nodes = [(mp.mpf(1) + mp.cos(mp.pi * (2 * j + 1) / 8)) / 2 for j in range(4)]  # Chebyshev nodes in w
    A = mp.matrix([[w ** 3, w ** 2, w, 1] for w in nodes])
    b = mp.matrix([g(a + w * h) for w in nodes])
    c = mp.lu_solve(A, b)
    rows.append([float(c[i]) for i in range(4)])

# measured error of the whole evaluation in float64 against 60-digit values
E = [float(x) for x in """1.30576633701114596e-08 1.50659498156726995e-07 1.30333658084232984e-06 1.01780513225220905e-05
6.16642915780934246e-05 3.33552565651518866e-04 1.54035309262363981e-03 5.55041086683500577e-03
1.28113950342307421e-02 1.49539700818229828e-02 5.76474860451242028e-03""".split()] if False else None
rng = np.random.default_rng(5)
ts = np.concatenate([rng.uniform(0, 26, 80000), rng.uniform(0, 4, 40000), np.linspace(0, 26, 4001)])
tab = np.array(rows)
x = 1.0 / (ts + 5.0)
kf = (x - float(x_lo)) * float(scale)
k = np.clip(kf.astype(np.int64), 0, N - 1)
w = kf - np.floor(kf)
gg = ((tab[k, 0] * w + tab[k, 1]) * w + tab[k, 2]) * w + tab[k, 3]
exact_g = np.array([float(g(mp.mpf(1) / (mp.mpf(float(t)) + 5))) for t in ts])
err = np.abs(gg / exact_g - 1).max()

print("// fpt_ndtr_gtab.hpp -- made by tools/fit_ndtr_gtab.py %d: the table of fptm::ndtr_fast_tab." % N)
print("// g(t) = Phi(-t) exp(t^2/2) as a cubic in w per interval of x = 1/(t + 5); measured relative error of g")
print("// on %d points of t in [0, 26], float64 evaluation against 60-digit values: %.2e" % (ts.size, err))
print("#pragma once")
print("#define FPT_NDTR_GTAB_N %d" % N)
print("#define FPT_NDTR_GTAB_XLO %.17e" % float(x_lo))
print("#define FPT_NDTR_GTAB_SCALE %.17e" % float(scale))
print("#define FPT_NDTR_GTAB_LIST \\")
for r in rows:
    print("    " + ", ".join("%.17e" % v for v in r) + ", \\")
print("    0.0")
print("// (entries are c3, c2, c1, c0 of g ~ ((c3 w + c2) w + c1) w + c0; the trailing 0.0 closes the list)")
sys.stderr.write("N=%d max rel err of g: %.3e\n" % (N, err))
