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
    """bedGraph / BED text of cli/utils.py:86-210: "%.4f" scores, merged segments."""
    import io
    from footprint_tools_amd import detect

    class IV(object):
        chrom, start, end = "chr7", 1000, 1004

    stats = np.array([[3.0, 2.0, 0.123456, np.nan, 1.0], [0.0, 10.0, 33.3, 0.00004, 0.5]])
    buf = io.StringIO()
    detect.write_stats_to_output(IV, stats, file=buf)
    assert buf.getvalue() == ("chr7\t1000\t1001\t3.0000\t2.0000\t0.1235\tnan\t1.0000\n"
                              "chr7\t1001\t1002\t0.0000\t10.0000\t33.3000\t0.0000\t0.5000\n")
    buf = io.StringIO()
    detect.write_stats_to_output(IV, stats, file=buf, filter_fn=lambda x: x[:, 1] >= 5)
    assert buf.getvalue().count("\n") == 1 and buf.getvalue().startswith("chr7\t1001\t1002")
    fdr_col = np.array([1, 1, .001, .001, .002, 1, 1, 1, .0005, 1.0])
    buf = io.StringIO()
    detect.write_segments_to_output(IV, fdr_col, 0.01, file=buf, decreasing=True)
    assert buf.getvalue() == "chr7\t1000\t1011\t.\t0.0005\n"
    buf = io.StringIO()
    detect.write_output_header(["exp", "obs"], file=buf, include_name=False, extra=["a", "b"])
    assert buf.getvalue().splitlines()[1:] == ["# a", "# b", "# chrom\tstart\tend\texp\tobs"]
