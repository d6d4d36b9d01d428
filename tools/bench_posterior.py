"""Timing of the posterior caller (fpt_posterior_dev: cli/post.py:98-124 for a whole batch in one
launch) on whole-genome-shaped input: ragged intervals (lognormal, mean ~162 bases), D datasets
with hotspot gaps, integer expected counts, an FDR track with ~5 % calls.  Diagnostic; prints one
line.  Usage: python tools/bench_posterior.py [n_intervals [n_datasets]]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from footprint_tools_amd import _lib  # noqa: E402
from footprint_tools_amd.scan import DeviceArray  # noqa: E402
from footprint_tools_amd.stats import posterior  # noqa: E402

n_iv = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lat = np.load("tests/golden/nb_lattice.npz")
rs = np.random.RandomState(3)
lens = np.clip(rs.lognormal(4.9, 0.62, n_iv), 50, 2000).astype(np.int64)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
total = int(off[-1])
exp = np.round(rs.gamma(2.0, 4.0, (D, total)))
obs = np.floor(exp * rs.uniform(0.0, 1.6, (D, total)))
fdr = rs.uniform(0, 1, (D, total)) ** 1.0
fdr[rs.uniform(0, 1, (D, total)) < 0.05] = 0.01
w = (rs.uniform(0, 1, (D, total)) < 0.9).astype(float)
obs[w == 0], exp[w == 0], fdr[w == 0] = 0.0, 0.0, 1.0
betas = np.array([[2.0 + 0.5 * d, 8.0 - 0.5 * d] for d in range(D)])
ctx = _lib.get_ctx()
packed = posterior.pack_models([(lat["mu_" + "ABC"[d % 3]], lat["r_" + "ABC"[d % 3]]) for d in range(D)])
d_in = DeviceArray(ctx, 4 * D * total * 8).upload(np.concatenate([obs.ravel(), exp.ravel(), fdr.ravel(), w.ravel()]))
d_off = DeviceArray(ctx, off.nbytes).upload(off)
d_out = DeviceArray(ctx, D * total * 8)
n = D * total * 8


def step():
    posterior.posterior_dev(ctx, n_iv, total, D, 0, betas, d_in.ptr, d_in.ptr + n, d_in.ptr + 2 * n, d_in.ptr + 3 * n,
                            d_out.ptr, interval_off_dev=d_off.ptr, max_interval_len=int(lens.max()), models=packed)


step()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
ctx.synchronize()
dt = (time.perf_counter() - t0) / 5
post = d_out.download(np.float64, 8 * D)
print("posterior: %d intervals, %d bases x %d datasets in %.3f ms -> %.3g bases/s, %.3g dataset-bases/s; post[:4]=%s"
      % (n_iv, total, D, dt * 1e3, total / dt, total * D / dt, np.round(post[:4], 4)))
