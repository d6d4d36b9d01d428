"""The device math header (footprint_tools_amd/csrc/fpt_math.hpp) compiled for the HOST and
checked against the oracle / golden vectors.  This is a CPU-side logic check of the source the
HIP kernels inline; the GPU build of the same source is checked in tests/test_gpu_*.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from .conftest import ROOT, golden, rel_err

f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hm") / "libhostmath.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-I", os.path.join(ROOT, "footprint_tools_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_math_shim.cpp"), "-o", so])
    L = C.CDLL(so)
    L.hm_map1.argtypes = [C.c_int, f64p, C.c_long, f64p]
    L.hm_incbet.argtypes = [f64p, f64p, f64p, C.c_long, f64p]
    L.hm_chdtrc.argtypes = [f64p, f64p, C.c_long, f64p]
    L.hm_nb_values.argtypes = [C.c_int, f64p, f64p, f64p, f64p, C.c_long, f64p]
    return L


def _map1(hm, op, x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    hm.hm_map1(op, x, x.size, out)
    return out


TOL = 1e-12  # host build, same libm: essentially exact; GPU tests use the 1e-6 contract


def test_special_functions_host(hm):
    g, w = golden("nb_lattice.npz"), golden("window.npz")
    assert rel_err(_map1(hm, 0, g["g_x"]), g["g_gamma"]) < TOL
    assert rel_err(_map1(hm, 1, g["g_x"]), g["g_lgam"]) < TOL
    assert rel_err(_map1(hm, 4, g["l1p_x"]), g["l1p_val"]) < TOL
    assert rel_err(_map1(hm, 3, w["ndtri_y"]), w["ndtri_val"]) < TOL
    assert rel_err(_map1(hm, 2, w["ndtr_a"]), w["ndtr_val"]) < TOL
    assert rel_err(_map1(hm, 5, w["ndtr_a"]), w["erf_val"]) < TOL
    assert rel_err(_map1(hm, 6, w["ndtr_a"]), w["erfc_val"]) < TOL
    out = np.empty_like(g["ib_val"])
    hm.hm_incbet(g["ib_a"], g["ib_b"], g["ib_x"], out.size, out)
    assert rel_err(out, g["ib_val"]) < TOL
    out = np.empty_like(w["ch_val"])
    hm.hm_chdtrc(w["ch_df"], w["ch_x"], out.size, out)
    assert rel_err(out, w["ch_val"]) < TOL


def test_log_fast_host(hm):
    """the posterior kernel's short logarithm (host build of the same source; the device refines a hardware
    reciprocal where this divides): within 2 ulp of numpy's log / log1p"""
    rs = np.random.RandomState(5)
    x = np.concatenate([np.exp(rs.uniform(-340, 340, 100000)), rs.uniform(0.5, 2.0, 100000), 1.0 + rs.uniform(-1e-6, 1e-6, 10000),
                        [1.0, 0.5, 2.0, 0.70710678118654746, 0.70710678118654757, 1e-150, 1e150]])
    got, want = _map1(hm, 8, x), np.log(x)
    assert (np.abs(got - want) / np.maximum(np.abs(want), 0.5)).max() < 4.5e-16
    u = np.concatenate([rs.uniform(0, 1, 100000), np.exp(rs.uniform(-745, 0, 50000)), [0.0, 1.0, 5e-324, 1e-17, 2.0 ** -53]])
    got, want = _map1(hm, 9, u), np.log1p(u)
    assert (np.abs(got - want) / np.maximum(want, 1e-300)).max() < 4.5e-16


def test_ndtr_window_host(hm):
    """The one-formula normal cdf of the fused scan's Stouffer windows against the reference's
    own ndtr values (golden grid: branch boundaries, |a| up to 37.5) and, densely, against the
    restated ndtr.c: <= 2e-11 relative for |a| < 26 (fit: 4.5e-12), identical beyond and for inf / NaN."""
    w = golden("window.npz")
    assert rel_err(_map1(hm, 7, w["ndtr_a"]), w["ndtr_val"]) < 2e-11
    rs = np.random.RandomState(7)
    a = np.concatenate([rs.uniform(-26, 26, 400000), rs.normal(0, 1.5, 400000), np.linspace(-26, 26, 20001),
                        [0.0, -0.0, 1e-300, -1e-300, 25.999999999, -25.999999999, 26.0, -26.0]])
    fast, ref = _map1(hm, 7, a), _map1(hm, 2, a)
    assert np.all((fast > 0) & (fast <= 1))
    assert rel_err(fast, ref) < 2e-11
    far = np.array([26.5, -26.5, 27.2, -27.2, 30.0, -30.0, 38.0, -38.0, 75.0, -75.0, np.inf, -np.inf, np.nan])
    f2, r2 = _map1(hm, 7, far), _map1(hm, 2, far)
    assert np.array_equal(f2, r2, equal_nan=True)
    # monotone to within rounding on a fine grid around the centre and into the tail
    x = np.linspace(-12, 12, 200001)
    y = _map1(hm, 7, x)
    assert np.all(np.diff(y) >= -2e-11 * y[1:])


def test_nb_lattice_host(hm):
    g = golden("nb_lattice.npz")
    for key in "ABCD":
        for what, name in enumerate(("cdf", "logpmf", "pmf")):
            out = np.empty_like(g["lat_exp"])
            rc = hm.hm_nb_values(what, g["mu_" + key], g["r_" + key], g["lat_exp"], g["lat_obs"], out.size, out)
            assert rc == 0
            assert rel_err(out, g["%s_%s" % (name, key)]) < TOL, (key, name)
    out = np.empty(3)
    assert hm.hm_nb_values(0, g["mu_D"], g["r_D"], np.array([1., 2.5, 3.]), np.ones(3), 3, out) == 1


def test_uniform_word_of_null_sampler(hm):
    """The word of a caller-supplied double: floor(u 2^32), out-of-range / NaN inside the table, and the
    library's own uniforms (w + 1/2) 2^-32 give back w."""
    hm.hm_uniform_word.argtypes = [C.c_double]
    hm.hm_uniform_word.restype = C.c_uint32
    assert [hm.hm_uniform_word(x) for x in (0.0, 0.5, 1.0 - 2.0 ** -53, -1.0, 7.0, float("nan"), 2.0 ** -33)] == \
        [0, 2 ** 31, 2 ** 32 - 1, 0, 2 ** 32 - 1, 0, 0]
    rs = np.random.RandomState(4)
    uu = rs.random_sample(2000)
    ww = np.array([hm.hm_uniform_word(float(x)) for x in uu], np.float64)
    assert np.all(ww / 2.0 ** 32 <= uu) and np.all(uu < (ww + 1) / 2.0 ** 32)
    w = np.concatenate([rs.randint(0, 2 ** 32, 2000, dtype=np.uint64), [0, 1, 2 ** 21 - 1, 2 ** 21, 2 ** 32 - 1]])
    assert [hm.hm_uniform_word(float((x + 0.5) / 2.0 ** 32)) for x in w] == [int(x) for x in w]


def _alias_implied_pmf(lg, ent):
    """The distribution an alias table draws from, exactly: slot s gives itself for threshold / 2^(32-lg) of
    its 1 / n and its alias for the rest."""
    n = 1 << lg
    acc = (ent >> lg).astype(np.float64) / 2.0 ** (32 - lg)
    p = np.zeros(n)
    np.add.at(p, np.arange(n), acc / n)
    np.add.at(p, (ent & (n - 1)).astype(int), (1.0 - acc) / n)
    return p


def test_null_sampler_alias_tables(orc):
    """The alias tables of the null sampler (include/fpt.h, fpt_fdr_dev): the width of a row is the smallest
    power of two whose rest is <= 2^-32 (or the cap), every entry's alias is an outcome of the row, and the
    distribution the table draws from is the row's pmf to a few 2^-32 per outcome."""
    lat = golden("nb_lattice.npz")
    for key in "ABC":
        for ex in (0, 1, 2, 3, 7, 10, 40, 120, 255):
            lg, ent, cdf = orc.null_alias_row(lat["mu_" + key], lat["r_" + key], ex)
            n = 1 << lg
            assert 1 <= lg <= 11 and ent.size == n
            want = orc.nb_values("cdf", lat["mu_" + key], lat["r_" + key], np.full(n, float(ex)), np.arange(n, dtype=np.float64))
            assert np.array_equal(cdf, want)
            rest = 1.0 - cdf[n - 2]
            assert rest <= 2.0 ** -32 or lg == 11
            if lg > 1 and n // 2 - 2 >= 0:
                assert 1.0 - cdf[n // 2 - 2] > 2.0 ** -32  # no narrower table would do
            pm = np.diff(np.concatenate([[0.0], cdf[:n - 1], [1.0]]))
            assert np.abs(_alias_implied_pmf(lg, ent) - pm).max() < 4e-9, (key, ex)
            assert np.all(pm[(ent & (n - 1)).astype(int)][(ent >> lg) < (2 ** (32 - lg) - 1)] > 0)  # an alias has mass
    # a narrow memo caps the width: 2^lg <= table_k, the rest of the row goes to "n-1 or more"
    lg, ent, cdf = orc.null_alias_row(lat["mu_A"], lat["r_A"], 40, table_k=16)
    assert lg == 4
    pm = np.diff(np.concatenate([[0.0], cdf[:15], [1.0]]))
    assert pm[-1] > 0.1 and np.abs(_alias_implied_pmf(lg, ent) - pm).max() < 1e-9
    lg, ent, cdf = orc.null_alias_row(lat["mu_A"], lat["r_A"], 3, table_k=1)
    assert lg == 1 and ent.size == 2


def test_null_sampler_draws_follow_the_pmf(orc):
    """Draws of the null sampler from uniform words: outcome counts against the row's pmf (chi-square over the
    outcomes with an expected count of 20 or more), the rest of a capped row through the inverse cdf, and a
    non-integer expected value by the inverse cdf on u."""
    lat = golden("nb_lattice.npz")
    rs = np.random.RandomState(11)
    n_draw = 400000
    for key, ex, table in (("A", 3, (256, 2048)), ("B", 40, (256, 2048)), ("C", 10, (256, 2048)), ("A", 40, (256, 16))):
        w = rs.randint(0, 2 ** 32, n_draw, dtype=np.uint64)
        u = (w.astype(np.float64) + 0.5) / 2.0 ** 32
        k, p = orc.null_draws(lat["mu_" + key], lat["r_" + key], ex, u, table=table)
        ks = np.arange(4096, dtype=np.float64)
        cdf = orc.nb_values("cdf", lat["mu_" + key], lat["r_" + key], np.full(ks.size, float(ex)), ks)
        # every draw returns the cdf of an outcome; the outcome itself is k, or found by the search (k = -1)
        kk = np.where(k >= 0, k, np.searchsorted(cdf, p, side="left"))
        assert np.array_equal(cdf[kk], p)
        lg = orc.null_alias_row(lat["mu_" + key], lat["r_" + key], ex, table_k=table[1])[0]
        assert np.all(k[k >= 0] < (1 << lg) - 1) and np.all(kk[k < 0] >= (1 << lg) - 1)
        if table[1] == 16:
            assert (k < 0).mean() > 0.1  # the capped row: a good share of the draws is beyond the table
        pm = np.diff(np.concatenate([[0.0], cdf]))
        cnt = np.bincount(kk, minlength=pm.size)[:pm.size].astype(np.float64)
        big = pm * n_draw >= 20
        chi2 = ((cnt[big] - pm[big] * n_draw) ** 2 / (pm[big] * n_draw)).sum() + \
            (cnt[~big].sum() - pm[~big].sum() * n_draw) ** 2 / max(pm[~big].sum() * n_draw, 1.0)
        dof = int(big.sum())
        assert chi2 < dof + 5.0 * np.sqrt(2.0 * dof), (key, ex, table, chi2, dof)
    # non-integer expected value: the smallest k with cdf(k) >= u
    u = rs.random_sample(2000)
    k, p = orc.null_draws(lat["mu_A"], lat["r_A"], 2.5, u)
    ks = np.arange(512, dtype=np.float64)
    cdf = orc.nb_values("cdf", lat["mu_A"], lat["r_A"], np.full(ks.size, 2.5), ks)
    assert np.all(k == -1) and np.array_equal(p, cdf[np.searchsorted(cdf, u, side="left")])

