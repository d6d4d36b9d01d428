"""Diagnostic: distribution of the observed windows' y = ndtri(win p) in tools/bench_fdr_ragged.py's input."""
import sys
import numpy as np
from scipy.special import ndtri
sys.path.insert(0, ".")
sys.argv = [sys.argv[0], "20000", "4"]
exec(open("tools/bench_fdr_ragged.py").read().split("for times in")[0])
wp = d_out.download(np.float64, total, 3 * t8)
p = d_out.download(np.float64, total, 2 * t8)
y = ndtri(wp[np.isfinite(wp) & (wp < 1)])
print("win p quantiles 1,10,50,90,99 %:", np.quantile(wp, [.01, .1, .5, .9, .99]))
print("y quantiles:", np.quantile(y, [.01, .1, .5, .9, .99]), "share y > 0: %.3f, y > 3: %.4f, |y| < 0.01: %.4f" % ((y > 0).mean(), (y > 3).mean(), (np.abs(y) < 0.01).mean()))
print("p quantiles:", np.quantile(p, [.01, .1, .5, .9, .99]))
