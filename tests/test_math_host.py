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
    """The slot of a draw's 32-bit word: guide_edge(idx) <= u < guide_edge(idx + 1), the index and the
    position inside a slot grow with the word, and a walk that starts at the guide entry is short for
    every expected value of the NB lattice."""
    hm.hm_guide_edge.argtypes = [C.c_int]
    hm.hm_guide_edge.restype = C.c_double
    hm.hm_guide_word.argtypes = [C.c_double]
    hm.hm_guide_word.restype = C.c_uint32
    hm.hm_guide_index.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
    n = hm.hm_guide_entries()
    assert n == 3 * 257
    rs = np.random.RandomState(4)
    w = np.concatenate([rs.randint(0, 2 ** 32, 20000, dtype=np.uint64), rs.randint(0, 2 ** 24, 5000, dtype=np.uint64),
                        2 ** 32 - 1 - rs.randint(0, 2 ** 24, 5000, dtype=np.uint64),
                        [0, 1, 2 ** 16 - 1, 2 ** 16, 2 ** 24 - 1, 2 ** 24, 255 * 2 ** 24 - 1, 255 * 2 ** 24,
                         2 ** 32 - 2 ** 16 - 1, 2 ** 32 - 2 ** 16, 2 ** 32 - 1]]).astype(np.uint64)
    w.sort()
    u = (w.astype(np.float64) + 0.5) / 2.0 ** 32
    fr = C.c_float()
    loc = np.array([(hm.hm_guide_index(int(x), C.byref(fr)), fr.value) for x in w])
    idx, frac = loc[:, 0].astype(int), loc[:, 1]
    edge = np.array([hm.hm_guide_edge(int(s)) for s in range(n)])
    assert idx[0] == 257 and idx[-1] == 769 and idx.min() == 1 and 0 <= frac.min() and frac.max() < 1.0
    assert not np.any((idx == 0) | ((idx >= 255) & (idx <= 256)) | (idx == 513) | (idx == 770))  # closing entries only
    assert np.all(edge[idx] <= u) and np.all(u < edge[idx + 1])
    for lo, hi in ((0, 256), (257, 513), (514, 770)):  # every level: edges and indices in order
        assert np.all(np.diff(edge[lo:hi + 1]) > 0)
        lvl = (idx >= lo) & (idx < hi)
        assert np.all(np.diff(idx[lvl]) >= 0)
    same = np.diff(idx) == 0
    assert np.all(np.diff(frac)[same] >= 0)  # position inside a slot grows with the word
    assert edge[256] == 1.0 and edge[513] == edge[1] and edge[770] == 1.0 and edge[514] == edge[255]
    # the word of a caller-supplied double: floor(u 2^32), out-of-range / NaN inside the table
    assert [hm.hm_guide_word(x) for x in (0.0, 0.5, 1.0 - 2.0 ** -53, -1.0, 7.0, float("nan"), 2.0 ** -33)] == \
        [0, 2 ** 31, 2 ** 32 - 1, 0, 2 ** 32 - 1, 0, 0]
    uu = rs.random_sample(2000)
    ww = np.array([hm.hm_guide_word(float(x)) for x in uu], np.float64)
    assert np.all(ww / 2.0 ** 32 <= uu) and np.all(uu < (ww + 1) / 2.0 ** 32)
    # probes of the walk from the interpolated start (nb_draw_z2): up while cdf < u, down while the
    # entry below still qualifies
    lat = golden("nb_lattice.npz")
    ks = np.arange(256, dtype=np.float64)
    worst, mean = 0, 0.0
    for ex in (0.0, 1.0, 5.0, 19.0, 60.0):
        cdf = orc.nb_values("cdf", lat["mu_A"], lat["r_A"], np.full(256, ex), ks)
        g = np.minimum(np.searchsorted(cdf, edge, side="left"), 255)
        a, b = g[idx], g[idx + 1]
        k0 = a + np.where(b - a > 64, 0, (frac.astype(np.float32) * (b - a).astype(np.float32)).astype(int))
        stop = np.minimum(np.searchsorted(cdf, u, side="left"), 255)
        steps = np.where(stop > k0, stop - k0, np.where(k0 > a, k0 - stop + 1, 0))
        assert np.all((a <= stop) & (stop <= b))  # the bracket holds the answer
        inner = (u < 1 - 2.0 ** -16) & (u > 2.0 ** -16)
        worst = max(worst, int(steps[inner].max()))
        mean = max(mean, float(steps.mean()))  # (two thirds of the words are uniform, the rest from the tails)
    assert worst <= 8 and mean < 0.5, (worst, mean)
