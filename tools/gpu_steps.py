"""Step-by-step GPU bring-up (diagnostic, not a pytest file): each step prints before and
after so a hang or fault is attributable.  Run:  python tools/gpu_steps.py [step ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402  (checker first: no fork/exec after GPU init)

oracle.lib()
from footprint_tools_amd import _lib  # noqa: E402
from footprint_tools_amd.scan import FootprintScanner  # noqa: E402


def say(*a):
    print("[%7.2f]" % (time.time() - T0), *a, flush=True)


T0 = time.time()
steps = sys.argv[1:] or ["ctx", "special", "kmer", "predict", "nb", "window", "scan"]
g = np.load("tests/golden/kmer_probs.npz")
lat = np.load("tests/golden/nb_lattice.npz")
say("start", steps)
ctx = _lib.get_ctx()
say("ctx ok, device", ctx.device)
if "special" in steps:
    x = np.linspace(0.1, 30, 1000)
    out = np.empty_like(x)
    _lib.check(ctx.L.fpt_special(ctx.h, 1, _lib.ptr(x), None, None, x.size, _lib.ptr(out)))
    say("lgam err", np.max(np.abs(out - oracle.map1("lgam", x))))
    a, b, xx = np.full(1000, 5.0), np.linspace(1, 40, 1000), np.full(1000, 0.3)
    _lib.check(ctx.L.fpt_special(ctx.h, 7, _lib.ptr(a), _lib.ptr(b), _lib.ptr(xx), 1000, _lib.ptr(out)))
    say("incbet err", np.max(np.abs(out - oracle.incbet(a, b, xx))))
if "kmer" in steps:
    ctx.set_bias_table(g["table"], 1e-6)
    s = np.ascontiguousarray(g["seq0"])
    f, r = np.empty(s.size - 6), np.empty(s.size - 6)
    _lib.check(ctx.L.fpt_kmer_probs(ctx.h, _lib.ptr(s), s.size, _lib.ptr(f), _lib.ptr(r)))
    say("kmer exact", np.array_equal(f, g["fwd0"]), np.array_equal(r, g["rev0"]))
if "predict" in steps:
    from footprint_tools_amd.modeling import predict
    rs = np.random.RandomState(0)
    obs, pr = rs.randint(0, 20, 611).astype(float), rs.uniform(1e-3, .2, 611)
    e, w = predict.predict(obs, pr, 5, 50, 0.01)
    e0, w0 = oracle.fast_predict(obs, pr, 5, 50, 0.01)
    say("predict exp exact", np.array_equal(e, e0), "win err", np.max(np.abs(w - w0)))
if "nb" in steps:
    from footprint_tools_amd.modeling import dispersion
    dm = dispersion.dispersion_model()
    dm.mu_params, dm.r_params = lat["mu_A"], lat["r_A"]
    p = dm.p_values(lat["lat_exp"], lat["lat_obs"])
    say("p_values rel err", np.nanmax(np.abs(p - lat["cdf_A"]) / np.maximum(lat["cdf_A"], 1e-300)))
if "window" in steps:
    from footprint_tools_amd.stats import windowing
    x = np.random.RandomState(1).uniform(0, 1, 500)
    w = windowing.stouffers_z(x, 3)
    say("stouffer err", np.max(np.abs(w - oracle.window("stouffers_z", x, 3))))
if "scan" in steps:
    class DM(object):
        mu_params, r_params = lat["mu_A"], lat["r_A"]
    for (n_iv, L, scales) in [(4, 200, (3,)), (64, 500, (3, 5, 10, 20, 40)), (16, 1000, (3, 40))]:
        sc = FootprintScanner(g["table"], DM, 5, 50, 0.01, scales)
        l = sc.padded_len(L)
        cp, cm = oracle.synth_counts(1, 0, n_iv * l, 0), oracle.synth_counts(1, 0, n_iv * l, 1)
        sq = oracle.synth_bases(1, 0, n_iv * (l + 6))
        say("scan launch", n_iv, L, scales)
        out = sc.scan(cp, cm, sq, interval_len=L)
        e, o, p, wp = oracle.detect_batch(cp, cm, sq, n_iv, L, 5, 50, 0.01, g["table"], DM.mu_params,
                                          DM.r_params, scales)
        say("scan exp exact", np.array_equal(out["exp"], e), "obs", np.array_equal(out["obs"], o),
            "p err", np.nanmax(np.abs(out["pval"] - p) / np.maximum(p, 1e-300)),
            "winp nanmask", np.array_equal(np.isnan(out["winp"]), np.isnan(wp)),
            "winp err", np.nanmax(np.abs(out["winp"] - wp) / np.maximum(wp, 1e-300)),
            "ms", sc.last_kernel_ms())
say("done")
