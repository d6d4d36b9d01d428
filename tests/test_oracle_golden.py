"""The CPU checker (oracle/fpt_oracle.c) against the golden vectors produced by the genuine
reference (tests/golden/make_golden.py) and, when present, against the reference's own native
code compiled into oracle/_ref/libfpt_ref.so.  Same libm, same operation order: the bar is
bit-exact unless a test says otherwise."""
import hashlib

import numpy as np
import pytest

from .conftest import golden, rel_err


def same(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


# ---------------------------------------------------------------- A1
def test_kmer_probs_golden(orc):
    g = golden("kmer_probs.npz")
    assert g["table"].shape == (4096,)
    for i in range(int(g["n_seq"])):
        fwd, rev, fi, ri = orc.kmer_probs(g["seq%d" % i], g["table"])
        assert same(fwd, g["fwd%d" % i]) and same(rev, g["rev%d" % i])
        ok = fi >= 0
        assert np.array_equal(fwd[ok], g["table"][fi[ok]])
        assert np.all(fwd[~ok] == 1e-6)
    # index definition: first base most significant, A=0 C=1 G=2 T=3
    _, _, fi, ri = orc.kmer_probs("AAAAACGTTTTT", g["table"])
    assert fi[0] == 1 and fi[1] == 6 and ri[5] == 1  # revcomp(GTTTTT)=AAAAAC


# ---------------------------------------------------------------- A2-A4
def test_fast_predict_golden(orc):
    g = golden("predict.npz")
    meta = g["meta"]
    for c, (hw, shw, clip, l) in enumerate(meta):
        e, w = orc.fast_predict(g["obs%d" % c], g["probs%d" % c], int(hw), int(shw), float(clip))
        assert same(e, g["exp%d" % c]), "exp case %d %s" % (c, meta[c])
        assert same(w, g["win%d" % c]), "win case %d %s" % (c, meta[c])


def test_fast_predict_rounding_ties_golden(orc):
    """predict_ties.npz: rows where the rounding noise of the reference's own trimmed sum decides
    round() on dozens of positions (meta[:, 4] counts them) -- the oracle follows the reference's
    order of operations, so it is bit-identical there too."""
    g = golden("predict_ties.npz")
    meta = g["meta"]
    assert meta[:, 4].sum() > 50
    for c, (hw, shw, clip, l, n_flip) in enumerate(meta):
        e, w = orc.fast_predict(g["obs%d" % c], g["probs%d" % c], int(hw), int(shw), float(clip))
        assert same(e, g["exp%d" % c]), "exp case %d %s" % (c, meta[c])
        assert same(w, g["win%d" % c]), "win case %d %s" % (c, meta[c])


def test_trimmed_mean_quirks(orc):
    """OS1 == OS2 windows."""
    p = np.full(301, 0.1)
    e, w = orc.fast_predict(np.full(301, 5.0), p, 0, 50, 0.01)  # hw=0: window sums are 0
    assert np.all(w == 0)
    obs = np.full(400, 0.5)  # window sum (2*hw=10 wide) == 5.0 everywhere inside
    e, w = orc.fast_predict(obs, p[:1].repeat(400), 5, 50, 0.01)
    assert abs(w[200] - 5.0 * 100 / 99) < 1e-12


@pytest.mark.skipif("__import__('oracle.oracle').oracle.ref_lib() is None")
def test_fast_predict_vs_reference_c(orc):
    rs = np.random.RandomState(5)
    for (hw, shw, clip) in [(5, 50, .01), (4, 20, .1), (5, 50, .3), (6, 0, .01), (2, 3, .2)]:
        for l in (7, 64, 333, 1111):
            for kind in range(3):
                obs = [rs.randint(0, 20, l), rs.poisson(.2, l), rs.gamma(2., 2., l)][kind].astype(float)
                probs = rs.uniform(1e-4, .2, l)
                e, w = orc.fast_predict(obs, probs, hw, shw, clip)
                e2, w2 = orc.fast_predict(obs, probs, hw, shw, clip, ref=True)
                assert same(e, e2) and same(w, w2)


# ---------------------------------------------------------------- A5-A7
def test_fit_and_nb_lattice_golden(orc):
    g = golden("nb_lattice.npz")
    for key in "ABCD":
        mu, r = g["mu_" + key], g["r_" + key]
        xs = g["fit_x"]
        assert same([orc.fit_mu(mu, x) for x in xs], g["fit_mu_" + key])
        got = []
        for x, zd in zip(xs, g["fit_r_zerodiv_" + key]):
            if zd:
                with pytest.raises(ZeroDivisionError):
                    orc.fit_r(r, x)
                got.append(np.nan)
            else:
                got.append(orc.fit_r(r, x))
        assert same(got, g["fit_r_" + key])
        for what in ("cdf", "logpmf", "pmf"):
            v = orc.nb_values(what, mu, r, g["lat_exp"], g["lat_obs"])
            assert same(v, g["%s_%s" % (what, key)]), (key, what)
    for key in "ABC":
        for what in ("cdf", "logpmf"):
            v = orc.nb_values(what, g["mu_" + key], g["r_" + key], g["tail_exp"], g["tail_obs"])
            assert same(v, g["tail_%s_%s" % (what, key)])
    L = orc.lib()
    for fn, key in ((L.orc_nb_cdf, "sc_cdf"), (L.orc_nb_logpmf, "sc_logpmf"), (L.orc_nb_pmf, "sc_pmf")):
        v = [fn(int(k), float(p), float(r)) for k, p, r in zip(g["sc_k"], g["sc_p"], g["sc_r"])]
        assert same(v, g[key]), key


def test_zero_division_in_p_values(orc):
    g = golden("nb_lattice.npz")
    with pytest.raises(ZeroDivisionError):
        orc.nb_values("cdf", g["mu_D"], g["r_D"], np.array([1.0, 2.5, 3.0]), np.array([1.0, 1.0, 1.0]))


def test_hcephes_grids_golden(orc):
    g = golden("nb_lattice.npz")
    assert same(orc.incbet(g["ib_a"], g["ib_b"], g["ib_x"]), g["ib_val"])
    assert same(orc.map1("gamma", g["g_x"]), g["g_gamma"])
    assert same(orc.map1("lgam", g["g_x"]), g["g_lgam"])
    assert same(orc.map1("log1p", g["l1p_x"]), g["l1p_val"])
    w = golden("window.npz")
    assert same(orc.map1("ndtri", w["ndtri_y"]), w["ndtri_val"])
    assert same(orc.map1("ndtr", w["ndtr_a"]), w["ndtr_val"])
    assert same(orc.map1("erf", w["ndtr_a"]), w["erf_val"])
    assert same(orc.map1("erfc", w["ndtr_a"]), w["erfc_val"])
    assert same(orc.chdtrc(w["ch_df"], w["ch_x"]), w["ch_val"])


def test_incbet_sanity_vs_scipy(orc):
    """independent sanity check of the fixture itself (not a parity bar)."""
    from scipy.special import betainc
    rs = np.random.RandomState(3)
    a, b, x = rs.uniform(.1, 60, 2000), rs.uniform(.1, 60, 2000), rs.uniform(0.001, .999, 2000)
    got, want = orc.incbet(a, b, x), betainc(a, b, x)
    big = want > 1e-290
    assert rel_err(got[big], want[big]) < 1e-11


# ---------------------------------------------------------------- A8-A9
def test_windows_golden(orc):
    g = golden("window.npz")
    for nm in g["names"]:
        x, w = g["x_" + nm], g["w_" + nm]
        for hw in g["hws"]:
            for fn in ("sum", "product", "fishers_combined", "stouffers_z"):
                assert same(orc.window(fn, x, int(hw)), g["%s_%s_%d" % (fn, nm, hw)]), (fn, nm, hw)
            assert same(orc.window("weighted_stouffers_z", x, int(hw), w),
                        g["weighted_stouffers_z_%s_%d" % (nm, hw)]), (nm, hw)


def test_window_nan_semantics(orc):
    """edges are 1.0; p<2^-54, p==0, p>=1, NaN poison the whole window."""
    p = np.full(40, 0.3)
    p[10], p[20], p[30] = 1e-18, 1.0, np.nan
    out = orc.window("stouffers_z", p, 3)
    assert np.all(out[:3] == 1.0) and np.all(out[-3:] == 1.0)
    for c in (10, 20, 30):
        assert np.all(np.isnan(out[c - 3:c + 4]))
    assert np.all(np.isfinite(out[14:17]))
    assert np.all(orc.window("sum", p[:5], 3) == 1.0)  # n <= 2*hw -> all ones


# ---------------------------------------------------------------- A10
def test_fdr_helpers_golden(orc):
    g = golden("fdr.npz")
    assert same(orc.bisect(g["bis_a"], g["bis_b"]), g["bis_out"])
    assert same(orc.emperical_fdr(g["null"], g["pvals"]), g["efdr"])
    for k, (thr, w, dec) in enumerate(g["seg_params"]):
        assert orc.segment(g["seg_x"], float(thr), int(w), bool(dec)) == g["seg%d" % k].tolist()


# ---------------------------------------------------------------- A11
def test_log_likelihood_golden(orc):
    from oracle.oracle import DM_SYNTH_A  # noqa: F401
    g = golden("posterior.npz")
    lat = golden("nb_lattice.npz")
    for i, key in enumerate(g["dm_keys"]):
        mu, r = lat["mu_" + str(key)], lat["r_" + str(key)]
        on = orc.log_likelihood_row(mu, r, g["obs"][i], g["exp"][i], g["delta"], 3)
        off = orc.log_likelihood_row(mu, r, g["obs"][i], g["exp"][i], 1.0, 3)
        assert same(on, g["ll_on"][i]) and same(off, g["ll_off"][i])


def test_posterior_sequence_golden(orc):
    """the checker's restatement of cli/post.py:109-122 (priors, likelihoods, log-sum-exp, clamp)
    against what the reference's functions returned (posterior.npz) and what its driver class
    returned for whole track files (post_driver.npz)."""
    g = golden("posterior.npz")
    lat = golden("nb_lattice.npz")
    models = [(lat["mu_" + str(k)], lat["r_" + str(k)]) for k in g["dm_keys"]]
    stats, pc = orc.posterior_stats(g["obs"], g["exp"], g["fdr"], g["w"], g["betas"], models, cutoff=0.05)
    assert same(pc["prior"], g["prior"])
    assert np.allclose(pc["delta"], g["delta"], rtol=1e-13, atol=0)
    assert np.allclose(pc["ll_on"], g["ll_on"], rtol=1e-12, atol=0) and same(pc["ll_off"], g["ll_off"])
    want = -g["post"]
    want[want <= 0] = 0.0
    assert np.allclose(stats, want.T, rtol=1e-9, atol=1e-12, equal_nan=True)


# ---------------------------------------------------------------- whole path, BASELINE config 1
def test_e2e_config1_golden(orc):
    g = golden("e2e_cfg1.npz")
    lat = golden("nb_lattice.npz")
    table = golden("kmer_probs.npz")["table"]
    n_iv, L, hw, shw = int(g["n_iv"]), int(g["L"]), int(g["hw"]), int(g["shw"])
    clip, scales, seed = float(g["clip"]), g["scales"].astype(np.int32), int(g["seed"])
    l = L + 2 * (hw + shw) + 1
    cp = orc.synth_counts(seed, 0, n_iv * l, 0)
    cm = orc.synth_counts(seed, 0, n_iv * l, 1)
    sq = orc.synth_bases(seed, 0, n_iv * (l + 6))
    e, o, p, wp = orc.detect_batch(cp, cm, sq, n_iv, L, hw, shw, clip, table, lat["mu_A"], lat["r_A"],
                                   scales, n_threads=4)
    nf = g["exp"].shape[0]
    assert same(e.reshape(n_iv, L)[:nf], g["exp"]) and same(o.reshape(n_iv, L)[:nf], g["obs"])
    assert same(p.reshape(n_iv, L)[:nf], g["p"])
    assert same(wp.reshape(len(scales), n_iv, L)[:, :nf].transpose(1, 0, 2), g["winp"])
    sha, sha_p = hashlib.sha256(), hashlib.sha256()
    W = wp.reshape(len(scales), n_iv, L)
    for i in range(n_iv):
        sha.update(e[i * L:(i + 1) * L].tobytes())
        sha.update(o[i * L:(i + 1) * L].tobytes())
        sha_p.update(p[i * L:(i + 1) * L].tobytes())
        for s in range(len(scales)):
            sha_p.update(np.ascontiguousarray(W[s, i]).tobytes())
    assert sha.hexdigest() == str(g["sha256_exp_obs"])
    assert sha_p.hexdigest() == str(g["sha256_p_winp"])


def test_synth_generator_matches_numpy_definition(orc):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "mk", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    src = open(spec.origin).read()
    ns = {"np": np}
    # only the two pure-numpy generator functions are needed (the module itself needs the reference)
    start, end = src.index("def splitmix64"), src.index("def table_2bit")
    exec(src[start:end], ns)
    for stream in (0, 1):
        assert np.array_equal(orc.synth_counts(7, 12345, 999, stream), ns["synth"](7, 12345, 999, stream))
    assert np.array_equal(orc.synth_bases(7, 5, 777), ns["synth"](7, 5, 777, 2))


def test_philox_known_answers(orc):
    """Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors)."""
    import ctypes as C
    L = orc.lib()
    L.orc_philox_raw.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
    for ctr, key, want in [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]:
        c = (C.c_uint32 * 4)(*ctr)
        L.orc_philox_raw(c, key[0], key[1])
        assert tuple(c) == want
    u = [L.orc_philox_uniform(5, b, s) for b in range(50) for s in range(40)]
    assert 0.0 <= min(u) and max(u) < 1.0 and abs(np.mean(u) - 0.5) < 0.03


def test_exp_obs_histogram_is_the_reference_loop(orc):
    """cli/learn_dm.py:276-287, statement for statement (Python's int(), numpy's indexing, `except IndexError`),
    on pairs that include fractions, values beyond either end and negative ones (an index from the end)"""
    rs = np.random.RandomState(12)
    e = np.concatenate([rs.randint(0, 260, 5000).astype(float), [2.9, -0.5, -1.0, -200.0, -201.0, 199.999, 200.0, 3.0]])
    o = np.concatenate([rs.randint(0, 1300, 5000).astype(float), [7.7, 4.0, -0.9, -1000.0, 5.0, -1001.0, 999.0, 1000.0]])
    cnts = np.column_stack([e, o])
    hist = np.zeros((200, 1000), dtype=int)
    for i in range(cnts.shape[0]):
        try:
            hist[int(cnts[i, 0]), int(cnts[i, 1])] += 1
        except IndexError:
            pass
    got = orc.exp_obs_histogram(e, o)
    assert np.array_equal(got, hist) and got[199, 0] >= 1 and got[0, 4] >= 1 and got.sum() == hist.sum() < e.size


def test_fdr_null_oracle_matches_reference_statistically(orc):
    """the reproducible inverse-CDF null sampler gives the same empirical FDR as the reference's
    numpy sampler up to Monte-Carlo noise (one interval, 400 draws per base)."""
    lat = golden("nb_lattice.npz")
    g = golden("e2e_cfg1.npz")
    exp, winp = g["exp"][0], g["winp"][0][0]
    L, times = exp.size, 400
    ef = orc.fdr_null(lat["mu_A"], lat["r_A"], exp, winp, 3, times, seed=3)
    rs = np.random.RandomState(0)
    r = np.array([orc.fit_r(lat["r_A"], x) for x in exp])
    mu = np.array([orc.fit_mu(lat["mu_A"], x) for x in exp])
    pn = np.empty((L, times))
    for i in range(L):  # dispersion.pyx:349-353
        k = rs.negative_binomial(r[i], r[i] / (r[i] + mu[i]), times)
        pn[i] = orc.nb_values("cdf", lat["mu_A"], lat["r_A"], np.full(times, exp[i]), k.astype(float))
    wn = np.stack([orc.window("stouffers_z", pn[:, s].copy(), 3) for s in range(times)], axis=1)
    ref = orc.emperical_fdr(wn, winp)
    assert np.mean(np.abs(ef - ref)) < 0.004 and np.max(np.abs(ef - ref)) < 0.03
