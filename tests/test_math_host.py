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


def test_guide_index_of_null_sampler(hm, orc):
    """guide_slot is monotone, guide_edge(slot) never exceeds a u of that slot, and a walk that
    starts at the guide entry is short for every expected value of the NB lattice."""
    hm.hm_guide_slot.argtypes = [C.c_double]
    hm.hm_guide_edge.argtypes = [C.c_int]
    hm.hm_guide_edge.restype = C.c_double
    n = hm.hm_guide_slots()
    rs = np.random.RandomState(4)
    u = np.concatenate([rs.random_sample(20000), 2.0 ** -rs.uniform(1, 40, 5000),
                        1.0 - 2.0 ** -rs.uniform(1, 40, 5000), [0.0, 0.5, 1.0 - 2.0 ** -53, 2.0 ** -53]])
    u.sort()
    slot = np.array([hm.hm_guide_slot(float(x)) for x in u])
    assert slot.min() == 0 and slot.max() == n - 1
    assert np.all(np.diff(slot) >= 0)
    edge = np.array([hm.hm_guide_edge(int(s)) for s in range(n)])
    assert np.all(np.diff(edge) >= 0) and edge[0] == 0.0
    assert np.all(edge[slot] <= u)
    hm.hm_guide_locate.argtypes = [C.c_double, C.POINTER(C.c_double)]
    fr = C.c_double()
    loc = np.array([(hm.hm_guide_locate(float(x), C.byref(fr)), fr.value) for x in u])
    assert np.array_equal(loc[:, 0], slot) and loc[:, 1].min() >= 0.0 and loc[:, 1].max() <= 1.0
    same = np.diff(slot) == 0
    assert np.all(np.diff(loc[:, 1])[same] >= 0)  # position inside a slot grows with u
    # out-of-range / NaN draws (caller-supplied uniforms) still index inside the table
    for bad in (-1.0, 1.0, 7.0, float("nan")):
        assert 0 <= hm.hm_guide_slot(bad) < n
    # walk length from the guide entry: number of k with edge <= cdf(k) < u
    lat = golden("nb_lattice.npz")
    ks = np.arange(256, dtype=np.float64)
    worst = 0
    for ex in (0.0, 1.0, 5.0, 19.0, 60.0):
        cdf = orc.nb_values("cdf", lat["mu_A"], lat["r_A"], np.full(256, ex), ks)
        start = np.searchsorted(cdf, edge[slot], side="left")
        stop = np.searchsorted(cdf, u, side="left")
        inside = stop < 256
        worst = max(worst, int((stop - start)[inside & (u < 1 - 2.0 ** -20) & (u > 2.0 ** -20)].max()))
    assert worst <= 4
