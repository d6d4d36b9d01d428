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
