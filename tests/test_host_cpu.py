"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/fpt.h
declares, fails loudly without a device, and the host logic of the Python mirror matches the
golden vectors."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from .conftest import ROOT, golden, has_gpu


@pytest.fixture(scope="module")
def lib():
    from footprint_tools_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "footprint_tools_amd", "csrc")])
    return _lib


def test_library_exports_header_symbols(lib):
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "fpt.h")).read()
    declared = sorted(set(re.findall(r"\b(fpt_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(L, sym), "libfpt_hip.so does not export %s" % sym
    assert sorted(lib.EXPORTS) == declared
    assert L.fpt_version() >= 100


def test_scan_desc_layout_matches_header(lib, tmp_path):
    """ctypes mirror of struct fpt_scan_desc has the C compiler's size and offsets."""
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fpt.h"\nint main(){printf("%zu %zu %zu %zu\\n",'
                   'sizeof(fpt_scan_desc), offsetof(fpt_scan_desc, scales), offsetof(fpt_scan_desc, counts_plus),'
                   'offsetof(fpt_scan_desc, status_out));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, o_scales, o_cp, o_st = map(int, subprocess.check_output([str(exe)]).split())
    D = lib.ScanDesc
    assert (C.sizeof(D), D.scales.offset, D.counts_plus.offset, D.status_out.offset) == (size, o_scales, o_cp, o_st)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fpt.h"\nint main(){printf("%zu %zu %zu %zu\\n",'
                   'sizeof(fpt_fdr_desc), offsetof(fpt_fdr_desc, seed), offsetof(fpt_fdr_desc, exp),'
                   'offsetof(fpt_fdr_desc, null_uniform));return 0;}\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, o_seed, o_exp, o_nu = map(int, subprocess.check_output([str(exe)]).split())
    F = lib.FdrDesc
    assert (C.sizeof(F), F.seed.offset, F.exp.offset, F.null_uniform.offset) == (size, o_seed, o_exp, o_nu)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fpt.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   'offsetof(fpt_fdr_desc, null_winp_out), sizeof(fpt_segment_desc), offsetof(fpt_segment_desc, track),'
                   'offsetof(fpt_segment_desc, threshold), offsetof(fpt_segment_desc, decreasing));return 0;}\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    o_nw, size, o_tr, o_th, o_dec = map(int, subprocess.check_output([str(exe)]).split())
    S = lib.SegmentDesc
    assert F.null_winp_out.offset == o_nw
    assert (C.sizeof(S), S.track.offset, S.threshold.offset, S.decreasing.offset) == (size, o_tr, o_th, o_dec)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fpt.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(fpt_posterior_desc), offsetof(fpt_posterior_desc, total_bases), offsetof(fpt_posterior_desc, fdr_cutoff),'
                   'offsetof(fpt_posterior_desc, betas), offsetof(fpt_posterior_desc, post_out), offsetof(fpt_posterior_desc, status_out),'
                   'sizeof(fpt_cutcount_desc), offsetof(fpt_cutcount_desc, flip));return 0;}\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, o_tb, o_fc, o_b, o_po, o_st, csize, o_flip = map(int, subprocess.check_output([str(exe)]).split())
    P = lib.PosteriorDesc
    assert (C.sizeof(P), P.total_bases.offset, P.fdr_cutoff.offset, P.betas.offset, P.post_out.offset,
            P.status_out.offset) == (size, o_tb, o_fc, o_b, o_po, o_st)
    from footprint_tools_amd.cutcounts import CutCountDesc
    assert (C.sizeof(CutCountDesc), CutCountDesc.flip.offset) == (csize, o_flip)


@pytest.mark.skipif(has_gpu(), reason="checks the no-device failure path")
def test_no_device_fails_loudly(lib):
    L = lib.load()
    h = C.c_void_p()
    rc = L.fpt_ctx_create(0, C.byref(h))
    assert rc == lib.FPT_ERR_NODEVICE and not h.value
    assert b"no CPU fallback" in L.fpt_last_error()
    with pytest.raises(lib.FptError):
        lib.Context(0)
    from footprint_tools_amd.stats import windowing
    with pytest.raises(lib.FptError):
        windowing.stouffers_z(np.ones(20) * 0.5, 3)


def test_product_never_imports_oracle():
    """the product package must not reach into oracle/ (or any CPU restatement)."""
    pkg = os.path.join(ROOT, "footprint_tools_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), "%s/%s mentions the oracle" % (dirpath, f)


def test_host_fits_match_golden():
    from footprint_tools_amd.modeling import dispersion
    g = golden("nb_lattice.npz")
    for key in "ABCD":
        dm = dispersion.dispersion_model()
        dm.mu_params, dm.r_params = g["mu_" + key], g["r_" + key]
        assert np.array_equal([dm.fit_mu(x) for x in g["fit_x"]], g["fit_mu_" + key])
        for x, zd, want in zip(g["fit_x"], g["fit_r_zerodiv_" + key], g["fit_r_" + key]):
            if zd:
                with pytest.raises(ZeroDivisionError):
                    dm.fit_r(x)
            else:
                assert dm.fit_r(x) == want
    import pickle
    dm2 = pickle.loads(pickle.dumps(dm))
    assert np.array_equal(dm2.r_params, dm.r_params)


def test_fdr_utils_match_golden():
    from footprint_tools_amd.stats import fdr, utils
    g = golden("fdr.npz")
    assert np.array_equal(utils.bisect(g["bis_a"], g["bis_b"]), g["bis_out"])
    assert np.array_equal(fdr.emperical_fdr(g["null"], g["pvals"]), g["efdr"])
    for k, (thr, w, dec) in enumerate(g["seg_params"]):
        assert utils.segment(g["seg_x"], float(thr), int(w), bool(dec)) == g["seg%d" % k].tolist()
    # unsorted / NaN-containing `a` takes the literal two-pointer path
    a = np.array([0.1, np.nan, 0.05, 0.7])
    b = np.array([0.06, 0.2, np.nan])
    assert np.array_equal(utils.bisect(a, b), [0, 3, 4])


def test_posterior_host_parts_match_golden():
    from footprint_tools_amd.stats import posterior
    g = golden("posterior.npz")
    assert np.array_equal(posterior.compute_prior_weighted(g["fdr"], g["w"]), g["prior"])
    d = posterior.compute_delta_prior(g["obs"], g["exp"], g["fdr"], g["betas"])
    assert np.allclose(d, g["delta"], rtol=1e-12, equal_nan=True)
    post = posterior.posterior(g["prior"], g["ll_on"], g["ll_off"])
    assert np.allclose(post, g["post"], rtol=1e-12, equal_nan=True)


def test_bias_table_and_kmer_index():
    from footprint_tools_amd.modeling import bias
    assert bias.kmer_index("AAAAAC") == 1 and bias.kmer_index("TTTTTT") == 4095
    assert bias.kmer_index("ACGTN") is None and bias.kmer_index("ACGTNA") is None
    bm = bias.bias_model()
    bm["ACGTAC"] = 0.5
    t = bm.table()
    assert t[bias.kmer_index("ACGTAC")] == 0.5 and (t == 1e-6).sum() == 4095
    assert bm["NNNNNN"] == 1e-6 and bm.offset() == 3


def test_shard_intervals_balanced():
    from footprint_tools_amd.scan import shard_intervals
    for ws in (1, 2, 4, 8):
        sh = shard_intervals((100000, 500), ws, 55)
        assert sh[0][0] == 0 and sh[-1][1] == 100000
        assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        sizes = [b - a for a, b in sh]
        assert max(sizes) - min(sizes) <= 1
    rs = np.random.RandomState(1)
    lens = np.clip(rs.lognormal(5, .7, 5000).astype(int), 50, 2000)
    sh = shard_intervals(lens, 8, 55)
    cost = lens + 111
    per = [cost[a:b].sum() for a, b in sh]
    assert sum(b - a for a, b in sh) == 5000 and max(per) / (cost.sum() / 8) < 1.02


def test_output_writers_text_format():
    """bedGraph / BED text of cli/utils.py:86-210 against what the reference's own writer functions
    wrote for the same inputs (tests/golden/writers.npz, made by make_golden.py g9): "%.4f" scores,
    nan / inf spelling, filters, other delimiters and formats, merged segments both ways."""
    import io
    from footprint_tools_amd import detect
    g = np.load(os.path.join(ROOT, "tests", "golden", "writers.npz"))

    class IV(object):
        chrom, start, end = str(g["interval"][0]), int(g["interval"][1]), int(g["interval"][2])

    def text(fn, *a, **kw):
        buf = io.StringIO()
        fn(*a, file=buf, **kw)
        return buf.getvalue()

    stats = g["stats"]
    assert text(detect.write_stats_to_output, IV, stats) == str(g["stats_all"])
    assert text(detect.write_stats_to_output, IV, stats, filter_fn=lambda x: x[:, 1] >= 5) == str(g["stats_filtered"])
    assert text(detect.write_stats_to_output, IV, stats, delim=",", fmt_string="0.6e") == str(g["stats_fmt6e_comma"])
    for k in "abcde":
        col = g["fdr_" + k]
        assert text(detect.write_segments_to_output, IV, col, 0.01, decreasing=True) == str(g["seg_dec_" + k]), k
        assert text(detect.write_segments_to_output, IV, col, 0.5, name="fp", score_fn=np.max) == str(g["seg_inc_" + k]), k
    # the first header line names the package and its version: everything after it is the reference's
    hdr = text(detect.write_output_header, ["exp", "obs", "-log(pval)", "-log(winpval)", "fdr"], extra=["a", "b"])
    assert hdr.startswith("# generated by ") and hdr.splitlines()[1:] == str(g["header_full"]).splitlines()[1:]
    hdr = text(detect.write_output_header, ["score"], include_name=False, extra="x=1")
    assert hdr.splitlines()[1:] == str(g["header_noname"]).splitlines()[1:] and hdr.endswith("\n")
    hdr = text(detect.write_output_header, ["score"], delim=" ")
    assert hdr.splitlines()[1:] == str(g["header_plain"]).splitlines()[1:]


def test_native_stats_formatter_equals_python():
    """fpt_format_stats against Python's own "{:0.Nf}" (what cli/utils.py:155-161 prints) on random
    values over 40 decades and on the cases where the rounding could go either way."""
    import io
    from footprint_tools_amd import detect
    rs = np.random.RandomState(12)
    vals = [0.0, -0.0, 0.5, 1.5, 2.5, 0.125, 0.375, 0.00005, 0.00015, 0.99995, 0.999949999, -1e-9, 1e-320, 5e-324,
            4503599627370495.5, 4503599627370496.0, 9007199254740993.0, 1e17, 1e22, 1.7976931348623157e308,
            np.nan, -np.nan, np.inf, -np.inf, 12345.678951, 1e-5, 0.30000000000000004, 2.675, 1.005, 8.5e-5]
    vals += list(rs.randn(3000) * 10.0 ** rs.randint(-12, 21, 3000))
    vals += list(np.round(rs.rand(2000) * 100, 4) + rs.choice([0.0, 5e-5, -5e-5], 2000))  # near the 4-decimal ties
    vals += list((rs.randint(0, 2 ** 20, 1000) + 0.5) / 2.0 ** rs.randint(0, 12, 1000))   # exact binary ties
    vals = np.array(vals, dtype=np.float64)
    vals = np.concatenate([vals, -vals])
    k = 5
    m = np.resize(vals, (vals.size // k + 1, k))

    class IV(object):
        chrom, start = "chrUn_gl000220", 16000000

    for prec in (0, 1, 2, 4, 6, 9):
        fmt = "0.%df" % prec
        buf = io.StringIO()
        detect.write_stats_to_output(IV, m, file=buf, fmt_string=fmt)
        want = "".join("\t".join([IV.chrom, str(IV.start + i), str(IV.start + i + 1)]
                                 + [("{0:" + fmt + "}").format(v) for v in m[i]]) + "\n" for i in range(m.shape[0]))
        got = buf.getvalue()
        if got != want:
            for a, b in zip(got.splitlines(), want.splitlines()):
                assert a == b, (prec, a, b)
        assert got == want
    # row selection, another delimiter, an empty selection
    buf = io.StringIO()
    detect.write_stats_to_output(IV, m, file=buf, delim=",", filter_fn=lambda x: x[:, 0] > 1.0)
    rows = np.nonzero(m[:, 0] > 1.0)[0]
    assert buf.getvalue() == "".join(",".join([IV.chrom, str(IV.start + i), str(IV.start + i + 1)]
                                              + ["{0:0.4f}".format(v) for v in m[i]]) + "\n" for i in rows)
    buf = io.StringIO()
    detect.write_stats_to_output(IV, m, file=buf, filter_fn=lambda x: x[:, 0] > np.inf)
    assert buf.getvalue() == ""


# ---------------------------------------------------------------- learn_dm: NB fit, piecewise fit, model
def test_nbinom_fit_golden():
    """nbinom.mle / nbinom.fit against the reference's (stats/distributions/nbinom.pyx:25-80)."""
    from footprint_tools_amd.stats.distributions import nbinom
    g = golden("nbfit.npz")
    for c in range(int(g["n_case"])):
        x, guess = g["data%d" % c], g["guess%d" % c]
        assert np.allclose(nbinom.mle(guess, x, np.sum(x) / len(x)), g["mle%d" % c], rtol=1e-12, atol=1e-9)
        assert np.allclose(nbinom.fit(x, p=guess[0], r=guess[1]), g["fit%d" % c], rtol=1e-9)
        assert np.allclose(nbinom.fit(x), g["fit_noguess%d" % c], rtol=1e-9)


def test_piecewise_fit():
    from footprint_tools_amd.modeling.piecewise import PiecewiseLinFit
    rs = np.random.RandomState(5)
    breaks = np.array([0.0, 10.0, 25.0, 60.0])
    slopes = np.array([2.0, -0.5, 0.25])
    x = np.sort(rs.uniform(0, 60, 400))
    x[0], x[-1] = 0.0, 60.0
    knots = np.concatenate([[1.0], 1.0 + np.cumsum(slopes * np.diff(breaks))])
    y = np.interp(x, breaks, knots)
    f = PiecewiseLinFit(x, y)
    f.fit_with_breaks(breaks)
    assert np.allclose(f.slopes, slopes) and np.allclose(f.intercepts, knots[:-1] - slopes * breaks[:-1])
    assert f.ssr < 1e-18 and np.allclose(f.predict(x), y)
    assert f.fit_with_breaks_opt(breaks[1:-1]) < 1e-18 < f.fit_with_breaks_opt([12.0, 30.0])
    # the design matrix is the hinge basis: same solution as a hand-built least squares
    noisy = y + rs.normal(0, 0.3, x.size)
    f = PiecewiseLinFit(x, noisy)
    f.fit_with_breaks(breaks)
    A = np.column_stack([np.ones_like(x), x - breaks[0], np.maximum(x - breaks[1], 0), np.maximum(x - breaks[2], 0)])
    beta = np.linalg.lstsq(A, noisy, rcond=None)[0]
    assert np.allclose(f.beta, beta)
    # forced point: the curve passes through it and is the best such curve
    f.fit_with_breaks_force_points(breaks, [5.0], [4.0])
    assert abs(f.predict([5.0])[0] - 4.0) < 1e-9
    base = f.ssr
    for _ in range(20):
        b = f.beta + rs.normal(0, 1e-3, f.beta.size)
        b[0] += 4.0 - PiecewiseLinFit.predict(f, [5.0], beta=b, breaks=breaks)[0]  # restore the constraint
        resid = f.assemble_regression_matrix(breaks, f.x_data).dot(b) - f.y_data
        assert resid.dot(resid) >= base - 1e-9
    # unsorted input is sorted; extrapolation beyond the last breakpoint continues the last segment
    g = PiecewiseLinFit(x[::-1], y[::-1])
    g.fit_with_breaks(np.array([0.0, 10.0, 25.0, 45.0]))
    assert g.x_data[0] == 0.0 and g.n_segments == 3


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_piecewise_fit_properties_on_noisy_data(seed):
    """pwlf cannot be had here (absent from /root/reference and from the image), so the fitter is pinned by what
    the published method guarantees, on noisy data and through the calls the reference makes
    (dispersion.pyx:446-458): `optimize.minimize(fit.fit_with_breaks_opt, guess)` recovers the interior
    breakpoints of a continuous piecewise-linear truth within a tolerance the noise sets; the fitted curve is
    CONTINUOUS at every breakpoint; its residual sum of squares is not above that of the true curve; a forced
    point is met; the objective does not depend on the order of the interior breakpoints."""
    from scipy import optimize
    from footprint_tools_amd.modeling.piecewise import PiecewiseLinFit
    rs = np.random.RandomState(seed)
    breaks = np.array([0.0, 3.0 + rs.uniform(0, 1), 8.0 + rs.uniform(0, 2), 16.0 + rs.uniform(0, 3), 26.0 + rs.uniform(0, 3), 60.0])
    slopes = np.array([0.06, 0.02, 0.008, 0.003, 0.001]) * (1 + 0.2 * rs.uniform(-1, 1, 5))
    knots = np.concatenate([[0.05], 0.05 + np.cumsum(slopes * np.diff(breaks))])
    x = np.arange(0.0, 60.5, 0.5)
    truth = np.interp(x, breaks, knots)
    sigma = 2e-3
    y = truth + rs.normal(0, sigma, x.size)
    f = PiecewiseLinFit(x, y)
    res = optimize.minimize(f.fit_with_breaks_opt, [3.0, 7.0, 15.0, 25.0])  # the reference's call and its guess
    got = np.sort(res.x)
    # a kink of slope change d is located to about sigma / d x (a few points): the sharp ones well, the faint ones loosely
    kink = np.abs(np.diff(slopes))
    tol = np.maximum(1.0, 12.0 * sigma / kink)
    assert np.all(np.abs(got - breaks[1:-1]) < tol), (got, breaks[1:-1], tol)
    full = np.concatenate([[x[0]], got, [x[-1]]])
    f.fit_with_breaks(full)
    true_ssr = float(((truth - y) ** 2).sum())
    assert f.ssr <= true_ssr * (1 + 1e-9)            # least squares over a family that holds the truth (up to its breaks)
    assert f.ssr > 0.5 * true_ssr                    # ... and no over-fit: ten parameters on 121 points
    for c in got:                                    # continuity: both one-sided limits agree at every breakpoint
        lo, hi = f.predict(np.array([c - 1e-9]))[0], f.predict(np.array([c + 1e-9]))[0]
        assert abs(lo - hi) < 1e-7
    seg_val = f.intercepts + f.slopes * full[:-1]    # ... and the per-segment lines the model stores meet there too
    assert np.allclose(seg_val[1:], f.intercepts[:-1] + f.slopes[:-1] * full[1:-1], atol=1e-9)
    assert np.allclose(f.slopes, slopes, atol=8 * sigma)
    assert abs(f.fit_with_breaks_opt(got) - f.fit_with_breaks_opt(got[::-1])) < 1e-15
    f.fit_with_breaks_force_points(full, [1.0], [y[2]])   # dispersion.pyx:458 forces the point at x = 1
    assert abs(f.predict(np.array([1.0]))[0] - y[2]) < 1e-12 and f.ssr >= true_ssr * 0.5


def test_learn_dispersion_model_recovers_truth():
    """histogram simulated from a known mu(x), r(x) -> learn_dispersion_model -> the model
    reproduces them (dispersion.pyx:357-469 flow with the pwlf-free fitter)."""
    from footprint_tools_amd.modeling import dispersion
    rs = np.random.RandomState(3)
    rows, cols = 80, 400
    h = np.zeros((rows, cols), dtype=np.int64)
    mu_true = lambda e: 0.5 + 0.9 * e
    invr_true = lambda e: 0.05 + 0.004 * e
    for e in range(rows):
        n = int(40000 / (1 + 0.08 * e))
        r = 1.0 / invr_true(e)
        k = rs.negative_binomial(r, r / (r + mu_true(e)), n)
        np.add.at(h[e], k[k < cols], 1)
    np.random.seed(0)
    dm = dispersion.learn_dispersion_model(h)  # default 2.5 % trimming: biases r upwards, as in the reference
    assert dm.mu_params.shape == (9,) and dm.r_params.shape == (15,)
    assert np.sum(np.isfinite(dm.p)) == rows and dm.r.max() <= 200.0
    for e in (1, 5, 20, 40):
        assert abs(dm.fit_mu(e) - mu_true(e)) / mu_true(e) < 0.08, e
    dm = dispersion.learn_dispersion_model(h, trim=(0, 100))  # untrimmed: the fit is consistent
    for e in (1, 5, 20, 40):
        assert abs(dm.fit_mu(e) - mu_true(e)) / mu_true(e) < 0.05, e
        assert abs(1.0 / dm.fit_r(e) - invr_true(e)) / invr_true(e) < 0.12, e
    # a histogram with too little data
    with pytest.raises(ValueError):
        dispersion.learn_dispersion_model(np.ones((5, 10), dtype=np.int64))
    # round trip through the JSON writer / loader
    txt = dispersion.write_dispersion_model(dm)
    assert "mu_params" in txt


def test_learn_dm_row_with_variance_equal_to_mean():
    """The reference replaces the moment estimate of r only when it is <= 0 (dispersion.pyx:417-419):
    a row whose trimmed sample has variance == mean starts the root finder at (NaN, inf), gets that
    start back, and drops out of the curve fits through its non-finite mean.  (Checked against the
    reference's nbinom.fit in the build container: (nan, inf) for this very sample.)"""
    from footprint_tools_amd.modeling import dispersion
    h = np.zeros((3, 3), dtype=np.int64)
    h[0] = [500, 0, 500]      # mean 1, variance 1
    h[1] = [100, 300, 100]    # under-dispersed: negative estimate -> start r = 10
    h[2] = [5, 0, 0]          # fewer than `cutoff` observations
    p, r = dispersion._nb_fit_rows(h, cutoff=250, trim=(0, 100))
    assert np.isnan(p[0]) and np.isinf(r[0])
    assert np.isfinite(p[1]) and np.isfinite(r[1])
    assert np.isnan(p[2]) and np.isnan(r[2])


def test_dispersion_model_json_roundtrip(tmp_path):
    """the reference's JSON schema (dispersion.pyx:471-549): [dtype, base64, shape] per array"""
    import json
    from footprint_tools_amd.modeling import dispersion
    dm = dispersion.dispersion_model()
    dm.mu_params = [25, 50, 75, 0, 0.5, 1.0, 1.0, 0.98, 0.97]
    dm.r_params = np.linspace(0.01, 0.15, 15)
    dm.h = np.arange(12, dtype=np.int64).reshape(3, 4)
    text = dispersion.write_dispersion_model(dm, extra="unit test")
    fields = json.loads(text)
    assert fields["h"][0] == "int64" and fields["h"][2] == [3, 4] and fields["metadata"] == "unit test"
    path = tmp_path / "dm.json"
    path.write_text(text)
    back = dispersion.load_dispersion_model(str(path))
    assert np.array_equal(back.mu_params, np.asarray(dm.mu_params, dtype=float))
    assert np.array_equal(back.r_params, dm.r_params) and np.array_equal(back.h, dm.h)
    enc = dispersion.base64encode(np.array([[1.5, 2.5]]))
    assert enc[0] == "float64" and tuple(enc[2]) == (1, 2) and np.array_equal(dispersion.base64decode(enc), [[1.5, 2.5]])


def test_interval_file_and_duck_type(tmp_path):
    """cli/detect.py:50,118: a headerless tab-separated interval file -> genomic_interval objects with
    the attributes the path uses (chrom, start, end, widen -> copy, len)."""
    import gzip
    from footprint_tools_amd.intervals import genomic_interval, read_intervals
    text = "# a comment\nchr1\t100\t250\nchr1\t900\t1000\tpeak7\t55\t-\n\nchrX\t5\t6\textra\n"
    p = tmp_path / "iv.bed"
    p.write_text(text)
    ivs = read_intervals(str(p))
    assert [(i.chrom, i.start, i.end, i.strand) for i in ivs] == [("chr1", 100, 250, None), ("chr1", 900, 1000, "-"),
                                                                 ("chrX", 5, 6, None)]
    assert len(ivs[0]) == 150 and ivs[1].name == "peak7" and str(ivs[2]) == "chrX\t5\t6"
    w = ivs[0].widen(56)
    assert (w.start, w.end) == (44, 306) and (ivs[0].start, ivs[0].end) == (100, 250) and w is not ivs[0]
    assert genomic_interval("chr1", 100, 250) == ivs[0] and len({ivs[0], genomic_interval("chr1", 100, 250)}) == 1
    with gzip.open(str(tmp_path / "iv.bed.gz"), "wt") as f:
        f.write(text)
    assert read_intervals(str(tmp_path / "iv.bed.gz")) == ivs
    for bad in ("chr1\t5\n", "chr1\tx\t9\n", "chr1\t9\t5\n"):
        q = tmp_path / "bad.bed"
        q.write_text(bad)
        with pytest.raises(ValueError):
            read_intervals(str(q))


def test_qvalues_golden():
    """Storey's pi0 and q-values (stats/fdr/__init__.py:39-95) against the reference's on six p-value sets
    (uniform, enriched near zero, rounded to ties, none below 0.2); bh_qvalue -- whose reference body cannot run
    under Python 3 (the fixture records that) -- against scipy's Benjamini-Hochberg adjustment."""
    from scipy.stats import false_discovery_control
    from footprint_tools_amd.stats import fdr
    g = golden("qvalues.npz")
    assert int(g["bh_runs"]) == 0
    for k in range(6):
        p = g["p%d" % k]
        assert np.allclose(np.ravel(fdr.pi0est(p)), g["pi0_%d" % k], rtol=1e-13, atol=0)
        q = fdr.qvalue(p)
        assert np.allclose(q, g["q%d" % k], rtol=1e-12, atol=0), k
        bh = fdr.bh_qvalue(p)
        assert np.allclose(bh, false_discovery_control(p, method="bh"), rtol=1e-12, atol=0), k
    assert np.allclose(np.ravel(fdr.pi0est(g["p0"], g["lamb"])), g["pi0_lamb"], rtol=1e-13, atol=0)
    with pytest.raises(ValueError):
        fdr.bh_qvalue(np.array([0.1, 1.2]))
    assert fdr.bh_qvalue(np.zeros(0)).size == 0


def test_host_prefault_touches_without_changing_length():
    """fpt_host_prefault (no GPU, no context): a team of threads writes one zero byte per page of a fresh array so that
    the first copy into it pays no page faults.  Bounds: nothing outside [host, host + bytes) is written; small and
    empty arrays are left alone; bad arguments are refused."""
    from footprint_tools_amd import _lib
    L = _lib.load()
    n = (1 << 23) + 12345                      # 8 MiB and a ragged tail
    buf = np.full(n + 8192, 7, dtype=np.uint8)
    view = buf[4096 + 3:4096 + 3 + n]          # unaligned start, guard bytes either side
    assert L.fpt_host_prefault(view.ctypes.data, view.size) == 0
    assert np.all(buf[:4096 + 3] == 7) and np.all(buf[4096 + 3 + n:] == 7)      # the guards are untouched
    touched = np.flatnonzero(view == 0)
    assert touched.size >= n // 4096 - 2 and np.all(np.diff(touched) == 4096)     # one byte per page, nothing else
    small = np.full(1000, 7, dtype=np.uint8)
    assert L.fpt_host_prefault(small.ctypes.data, small.size) == 0 and np.all(small == 7)
    assert L.fpt_host_prefault(None, 0) == 0
    assert L.fpt_host_prefault(None, 10) == _lib.FPT_ERR_INVALID and L.fpt_host_prefault(small.ctypes.data, -1) == _lib.FPT_ERR_INVALID
