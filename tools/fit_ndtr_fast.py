#!/usr/bin/env python3
"""Coefficients of fptm::ndtr_fast (footprint_tools_amd/csrc/fpt_math.hpp).

    Phi(-t) = exp(-t^2/2) * g(t),   g(t) = Phi(-t) exp(t^2/2),   t in [0, 26]

g is approximated by a polynomial of degree N in v = 1/(t+K) - r0 (r0 = centre of 1/(t+K) over the
range), exp(-t^2/2) as 2^n 2^f with 2^f a polynomial of degree NE on |f| <= 1/2: truncated Chebyshev series of the
functions evaluated with mpmath at 60 digits, converted to the monomial basis in exact arithmetic.
Prints the macro bodies and the measured relative error against 60-digit values.

    python tools/fit_ndtr_fast.py [N NE K]        (defaults 14 8 5: 4.5e-12; 17 10 5: 2.5e-13)
"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
N, NE, K = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (14, 8, 5.0)
T = 26.0


def g(t):
    t = mp.mpf(t)
    return mp.ncdf(-t) * mp.exp(t * t / 2)


def cheb_fit(f, n, m):
    """first n+1 Chebyshev coefficients of f on [-1, 1] from m Gauss-Chebyshev nodes"""
    xs = [mp.cos(mp.pi * (2 * k + 1) / (2 * m)) for k in range(m)]
    fs = [f(x) for x in xs]
    c = [2 * mp.fsum(fs[k] * mp.cos(mp.pi * j * (2 * k + 1) / (2 * m)) for k in range(m)) / m for j in range(n + 1)]
    c[0] /= 2
    return c


def cheb2mono(c):
    n = len(c) - 1
    Tn = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for j in range(2, n + 1):
        a = [mp.mpf(0)] + [2 * x for x in Tn[j - 1]]
        b = Tn[j - 2] + [mp.mpf(0)] * (len(a) - len(Tn[j - 2]))
        Tn.append([x - y for x, y in zip(a, b)])
    out = [mp.mpf(0)] * (n + 1)
    for j in range(n + 1):
        for i, x in enumerate(Tn[j]):
            out[i] += c[j] * x
    return out


wlo, whi = 1 / (mp.mpf(T) + K), 1 / mp.mpf(K)
alpha, beta = 2 / (whi - wlo), -(whi + wlo) / (whi - wlo)  # u = alpha / (t + K) + beta in [-1, 1]
mono_u = cheb2mono(cheb_fit(lambda u: g(alpha / (u - beta) - K), N, 64))
r0 = -beta / alpha
gv = [float(mono_u[j] * alpha ** j) for j in range(N + 1)]  # polynomial in v = 1/(t+K) - r0
# exp(-t^2/2) = 2^q, q = t^2 * (-log2(e)/2) = n + f with n = rint(q): 2^f on |f| <= 1/2 (f = q - n is exact,
# and the rounding of q -- 1e-13 absolute at t = 26 -- is far below the fit's error)
h = mp.mpf(1) / 2
me = cheb2mono(cheb_fit(lambda u: mp.power(2, h * u), NE, 40))
ev = [float(me[i] / h ** i) for i in range(NE + 1)]

print("kNdtrR0 = %.17g   (K = %g)" % (float(r0), K))
print("FPT_NDTR_G_LIST (degree %d, highest power first):\n  " % N + ", ".join("%.17e" % x for x in gv[::-1]))
print("FPT_NDTR_E_LIST (degree %d):\n  " % NE + ", ".join("%.17e" % x for x in ev[::-1]))

rng = np.random.default_rng(3)
ts = np.concatenate([rng.uniform(0, T, 60000), rng.uniform(0, 4, 40000), np.linspace(0, T, 4001)])
exact = np.array([float(mp.ncdf(-mp.mpf(float(t)))) for t in ts])
v = 1.0 / (ts + K) - float(r0)
p = np.zeros_like(v)
for c in gv[::-1]:
    p = p * v + c
qq = (ts * ts) * (-0.5 * 1.4426950408889634)
n = np.rint(qq)
rr = qq - n
q = np.zeros_like(rr)
for c in ev[::-1]:
    q = q * rr + c
print("max relative error on %d points of [0, %g]: %.2e" % (ts.size, T, np.abs(np.ldexp(q * p, n.astype(int)) / exact - 1).max()))
