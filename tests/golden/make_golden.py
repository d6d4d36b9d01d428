#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz from the GENUINE reference.

Runs only in the dev container (needs /root/reference):
    python oracle/pyref/build_pyref.py      # cython+gcc on the reference's own sources -> /tmp
    make -C oracle                          # also builds oracle/_ref/libfpt_ref.so
    python tests/golden/make_golden.py

Expected outputs come from
  * the reference package itself (footprint_tools.modeling / .stats, imported through
    oracle/pyref/ref_import.py), and
  * the reference's native C (hcephes, predict.h, windowing.h) compiled where it lies into
    oracle/_ref/libfpt_ref.so, for functions that have no Python entry point
    (`cdef predict`, raw hcephes calls).
No code of this repo's restatement or HIP path is used to produce an expected value.
The reference has no tests or golden vectors of its own, so these files are
what pins parity.
"""
import ctypes as C
import hashlib
import os
import platform
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "pyref"))
import ref_import  # noqa: E402

ref_import.load()
from footprint_tools.modeling import bias, dispersion, predict  # noqa: E402
from footprint_tools.stats import fdr, posterior, utils, windowing  # noqa: E402
from footprint_tools.stats.distributions import nbinom  # noqa: E402

f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
REFC = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfpt_ref.so"))
REFC.ref_fast_predict.argtypes = [f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_double, f64p, f64p]
REFC.ref_map1.argtypes = [C.c_int, f64p, C.c_long, f64p]
REFC.ref_incbet_v.argtypes = [f64p, f64p, f64p, C.c_long, f64p]
REFC.ref_chdtrc_v.argtypes = [f64p, f64p, C.c_long, f64p]

PROVENANCE = np.array([
    "reference=vierstralab/footprint-tools v1.3.7 (+hcephes 0.4.1) built from /root/reference",
    "gcc -O2 -fwrapv; cython 3.x; " + platform.platform(),
    "numpy " + np.__version__ + "; python " + platform.python_version(),
])

DMS = {
    # SURVEY.md App. B canonical synthetic model
    "A": ([25, 50, 75, 0, 0.5, 1.0, 1.0, 0.98, 0.97],
          [3, 7, 15, 25, 75, 0.05, 0.08, 0.115, 0.16, 0.185, 0.02, 0.01, 0.005, 0.002, 0.001]),
    # Poisson-like (r in the hundreds: a+b > MAXGAM -> lgam epilogue of incbet)
    "B": ([10, 40, 80, 0.2, 0.1, 0.5, 0.95, 0.96, 0.955],
          [5, 10, 20, 40, 80, 0.002, 0.002, 0.003, 0.004, 0.006,
           0.0002, 0.0002, 0.0001, 0.00005, 0.00002]),
    # heavy dispersion (r ~ 1..2.5), mu clamps to 0.1 near 0
    "C": ([20, 60, 100, -0.5, 0.3, 1.5, 1.1, 1.05, 1.0],
          [2, 6, 12, 30, 60, 0.4, 0.45, 0.5, 0.6, 0.7, 0.03, 0.02, 0.015, 0.01, 0.005]),
    # first 1/r segment negative below x=2.5 (r clamps to 1e-6)
    "D": ([25, 50, 75, 0, 0.5, 1.0, 1.0, 0.98, 0.97],
          [3, 7, 15, 25, 75, -0.05, 0.08, 0.115, 0.16, 0.185, 0.02, 0.01, 0.005, 0.002, 0.001]),
}


def make_dm(key):
    dm = dispersion.dispersion_model()
    dm.mu_params, dm.r_params = DMS[key]
    return dm


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth(seed, pos0, n, stream):
    """SURVEY.md 8(d) counter-hash generator (bench-defined): stream 0/1 counts, 2 bases."""
    with np.errstate(over="ignore"):
        key = splitmix64(np.array([seed + stream], dtype=np.uint64))[0]
        h = splitmix64(key + np.arange(pos0, pos0 + n, dtype=np.uint64))
    if stream < 2:
        return ((h >> np.uint64(33)) % np.uint64(20)).astype(np.float64)
    return np.frombuffer(b"ACGT", np.uint8)[((h >> np.uint64(13)) & np.uint64(3)).astype(np.int64)]


def table_2bit(bm):
    """4096-entry table in 2-bit order (A=0,C=1,G=2,T=3; first base most significant)."""
    import itertools
    return np.array([bm["".join(k)] for k in itertools.product("ACGT", repeat=6)])


def ref_predict_c(obs, probs, hw, shw, clip):
    obs = np.ascontiguousarray(obs, np.float64)
    probs = np.ascontiguousarray(probs, np.float64)
    e, w = np.empty(obs.size), np.empty(obs.size)
    REFC.ref_fast_predict(obs, probs, obs.size, hw, shw, clip, e, w)
    return e, w


def ref_map1(op, x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    REFC.ref_map1(op, x, x.size, out)
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, provenance=PROVENANCE, **arrs)
    print("%-20s %8.1f KB" % (name, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------- G1: 6-mer lookup (A1)
def g1():
    bm = bias.kmer_model(os.path.join(ref_import.REF, "data", "vierstra_et_al.6mer-model.txt"))
    rs = np.random.RandomState(11)
    out = {"table": table_2bit(bm)}
    seqs = []
    s0 = "".join(rs.choice(list("ACGT"), 3000))
    seqs.append(s0)
    s1 = list("".join(rs.choice(list("ACGT"), 2000)))
    for i in rs.choice(2000, 20, replace=False):
        s1[i] = "N"
    s1[0:8] = list("NNNNNNNN")
    s1[500:503] = list("RYK")  # other IUPAC codes are unknown k-mers too
    seqs.append("".join(s1))
    s2 = "".join(rs.choice(list("ACGTacgtn"), 1500))
    seqs.append(s2)
    seqs.append("ACGTAC")  # len 6 -> empty output
    seqs.append("ACGTACG")  # one output
    for i, s in enumerate(seqs):
        up = s.upper()  # predict.pyx:140
        fwd = bm.probs(up)
        rev = bm.probs(predict.reverse_complement(up))[::-1]
        out["seq%d" % i] = np.frombuffer(s.encode(), np.uint8)
        out["fwd%d" % i] = np.asarray(fwd, np.float64)
        out["rev%d" % i] = np.asarray(rev, np.float64)
    out["n_seq"] = np.array(len(seqs))
    save("kmer_probs.npz", **out)
    return bm, out["table"]


# ---------------------------------------------------------------- G2: fast_predict (A2-A4)
def g2():
    rs = np.random.RandomState(22)
    params = [(5, 0, .01), (5, 50, .01), (3, 10, .01), (5, 50, .05), (5, 25, 0.0), (1, 1, .4),
              (5, 50, .2), (2, 7, .1)]
    kinds = ["u20", "pois", "zero", "five", "two", "float", "ramp"]
    lens = [11, 111, 611, 1111]
    out, meta = {}, []
    c = 0
    for (hw, shw, clip) in params:
        for kind in kinds:
            for l in lens:
                if l == 1111 and kind not in ("u20", "pois"):
                    continue
                if kind == "u20":
                    obs = rs.randint(0, 20, l).astype(float)
                elif kind == "pois":
                    obs = rs.poisson(0.3, l).astype(float)
                elif kind == "zero":
                    obs = np.zeros(l)
                elif kind == "five":
                    obs = np.full(l, 5.0)
                elif kind == "two":
                    obs = rs.choice([0.0, 7.0], l, p=[.9, .1])
                elif kind == "float":
                    obs = rs.gamma(2.0, 1.7, l)
                else:
                    obs = np.arange(l, dtype=float) % 37
                probs = rs.uniform(3e-4, 0.22, l)
                e, w = ref_predict_c(obs, probs, hw, shw, clip)
                out["obs%d" % c], out["probs%d" % c] = obs, probs
                out["exp%d" % c], out["win%d" % c] = e, w
                meta.append((hw, shw, clip, l))
                c += 1
    # degenerate lengths: l <= 2*hw, l <= 2*shw, l == 0/1
    for (hw, shw, clip, l) in [(5, 50, .01, 10), (5, 50, .01, 100), (5, 50, .01, 101),
                               (5, 50, .01, 102), (5, 0, .01, 1), (5, 3, .3, 8), (0, 2, .25, 9)]:
        obs = rs.randint(0, 20, l).astype(float)
        probs = rs.uniform(3e-4, 0.22, l)
        e, w = ref_predict_c(obs, probs, hw, shw, clip)
        out["obs%d" % c], out["probs%d" % c] = obs, probs
        out["exp%d" % c], out["win%d" % c] = e, w
        meta.append((hw, shw, clip, l))
        c += 1
    out["meta"] = np.array(meta, dtype=np.float64)
    save("predict.npz", **out)


# ---------------------------------------------------------------- G3: NB lattice (A5-A7) + hcephes grids
def g3():
    out = {}
    ne, no = 120, 80
    E, O = np.meshgrid(np.arange(ne, dtype=float), np.arange(no, dtype=float), indexing="ij")
    e, o = E.ravel().copy(), O.ravel().copy()
    out["lat_exp"], out["lat_obs"] = e, o
    xs = np.concatenate([np.arange(0, 260, dtype=float), np.array([0.5, 2.49, 2.5, 24.999, 1e3, 1e6])])
    out["fit_x"] = xs
    for key in DMS:
        dm = make_dm(key)
        out["mu_%s" % key], out["r_%s" % key] = np.asarray(dm.mu_params, float), np.asarray(dm.r_params, float)
        out["fit_mu_%s" % key] = np.array([dm.fit_mu(x) for x in xs])
        fr, zd = [], []
        for x in xs:  # fit_r raises ZeroDivisionError when the piecewise value is exactly 0
            try:
                fr.append(dm.fit_r(x))
                zd.append(0)
            except ZeroDivisionError:
                fr.append(np.nan)
                zd.append(1)
        out["fit_r_%s" % key], out["fit_r_zerodiv_%s" % key] = np.array(fr), np.array(zd)
        out["cdf_%s" % key] = np.asarray(dm.p_values(e, o)).copy()
        out["logpmf_%s" % key] = np.asarray(dm.log_pmf_values(e, o)).copy()
        out["pmf_%s" % key] = np.asarray(dm.pmf_values(e, o)).copy()
    # heavier tails / non-integer obs (C truncation), large counts
    rs = np.random.RandomState(33)
    e2 = np.round(rs.gamma(2.0, 30.0, 4000))
    o2 = np.floor(rs.gamma(1.5, 40.0, 4000)) + rs.choice([0.0, 0.25, 0.999], 4000)
    out["tail_exp"], out["tail_obs"] = e2, o2
    for key in ("A", "B", "C"):
        dm = make_dm(key)
        out["tail_cdf_%s" % key] = np.asarray(dm.p_values(e2, o2)).copy()
        out["tail_logpmf_%s" % key] = np.asarray(dm.log_pmf_values(e2, o2)).copy()
    # nbinom scalars (nbinom.pyx:82-138)
    ks = np.array([0, 1, 2, 5, 17, 40, 300, -1, -5], dtype=np.int32)
    ps = np.array([0.01, 0.3, 0.5, 0.9, 0.999])
    rr = np.array([1e-6, 0.5, 3.0, 20.0, 180.0, 2000.0])
    K, P, R = [a.ravel() for a in np.meshgrid(ks, ps, rr, indexing="ij")]
    out["sc_k"], out["sc_p"], out["sc_r"] = K.astype(np.int32), P, R
    out["sc_cdf"] = np.array([nbinom.cdf(int(k), p, r) for k, p, r in zip(K, P, R)])
    out["sc_logpmf"] = np.array([nbinom.logpmf(int(k), p, r) for k, p, r in zip(K, P, R)])
    out["sc_pmf"] = np.array([nbinom.pmf(int(k), p, r) for k, p, r in zip(K, P, R)])

    # raw hcephes grids incl. branch boundaries (incbet.c, gamma.c, unity.c)
    a = np.concatenate([np.logspace(-6, 3, 28), [1.0, 2.0, 33.0, 85.8, 170.0, 171.0, 172.0]])
    b = np.concatenate([np.logspace(-2, 3, 18), [1.0, 2.0, 11.0, 85.82, 171.7]])
    x = np.concatenate([np.logspace(-8, -0.01, 24), [0.5, 0.9, 0.95, 0.9500001, 0.99, 1 - 1e-9,
                                                     0.0, 1.0, -0.1, 1.5]])
    A, B, X = [v.ravel().copy() for v in np.meshgrid(a, b, x, indexing="ij")]
    ib = np.empty_like(A)
    REFC.ref_incbet_v(A, B, X, A.size, ib)
    out["ib_a"], out["ib_b"], out["ib_x"], out["ib_val"] = A, B, X, ib
    gx = np.concatenate([np.linspace(-40.5, 40.5, 163), np.logspace(-12, 2.3, 60),
                         [33.0, 33.0000001, 143.01608, 143.02, 171.6, 171.7, 172.0, 1e-9, -1e-10,
                          0.0, 2.0, 3.0, 1.0, -1.0, -2.5, 12.999, 13.0, 999.9, 1000.0, 1e8, 1.1e8,
                          1e300, 3e305, -34.5, -35.0, -100.25, np.inf, -np.inf, np.nan]])
    out["g_x"] = gx
    out["g_gamma"], out["g_lgam"] = ref_map1(0, gx), ref_map1(1, gx)
    lx = np.concatenate([np.linspace(-0.999, 1.5, 120), np.logspace(-18, -1, 40), -np.logspace(-18, -1, 40),
                         [0.0, -1.0, -1.5, 0.41421356237309503, 0.41421356237309515, -0.2928932188134524,
                          -0.29289321881345254]])
    out["l1p_x"], out["l1p_val"] = lx, ref_map1(4, lx)
    save("nb_lattice.npz", **out)


# ---------------------------------------------------------------- G4: windows (A8-A9) + ndtr/ndtri grids
def g4():
    rs = np.random.RandomState(44)
    out = {}
    arrays = {}
    arrays["unif"] = rs.uniform(0, 1, 300)
    sm = rs.uniform(0, 1, 200) ** 6
    arrays["small"] = sm
    pl = rs.uniform(0, 1, 260)
    pl[[7, 50, 51, 120, 200, 201, 255]] = [1e-18, 1e-15, 0.0, 1.0, np.nan, 5e-17, 6e-17]
    pl[[30, 31]] = [1.0 - 1e-16, 2.0]
    arrays["planted"] = pl
    arrays["short5"] = rs.uniform(0, 1, 5)
    arrays["one"] = np.array([0.3])
    arrays["empty"] = np.zeros(0)
    hws = [0, 1, 2, 3, 5, 10, 20, 40]
    fns = {"sum": windowing.sum, "product": windowing.product,
           "fishers_combined": windowing.fishers_combined, "stouffers_z": windowing.stouffers_z}
    names = []
    for nm, x in arrays.items():
        out["x_" + nm] = x
        names.append(nm)
        wts = rs.uniform(0.1, 3.0, x.size)
        out["w_" + nm] = wts
        for hw in hws:
            for fn, f in fns.items():
                out["%s_%s_%d" % (fn, nm, hw)] = np.asarray(f(np.ascontiguousarray(x), hw)) if x.size else np.ones(0)
            out["weighted_stouffers_z_%s_%d" % (nm, hw)] = (
                np.asarray(windowing.weighted_stouffers_z(np.ascontiguousarray(x), wts, hw)) if x.size else np.ones(0))
    out["names"] = np.array(names)
    out["hws"] = np.array(hws)
    # ndtri / ndtr / erf / erfc grids incl. boundaries (ndtri.c:48-88, ndtr.c:34-132)
    e2 = 0.13533528323661269189
    y = np.concatenate([np.linspace(0, 1, 401), np.logspace(-320, -1, 200), 1 - np.logspace(-16.5, -1, 100),
                        [e2, np.nextafter(e2, 1), np.nextafter(e2, 0), 1 - e2, np.nextafter(1 - e2, 1),
                         np.exp(-32.0), 1.2664165549e-14, 1.27e-14, 1.0, 0.0, -0.5, 1.5, 1 - 2 ** -53,
                         np.nan]])
    out["ndtri_y"], out["ndtri_val"] = y, ref_map1(3, y)
    a = np.concatenate([np.linspace(-40, 40, 801), [1.4142135623730951, -1.4142135623730951, 11.313708498984761,
                                                    -11.3137085, 37.5, -37.5, 37.6, -37.6, -38.5, 1e-300, np.inf,
                                                    -np.inf, np.nan], rs.normal(0, 3, 300)])
    out["ndtr_a"], out["ndtr_val"] = a, ref_map1(2, a)
    out["erf_val"], out["erfc_val"] = ref_map1(5, a), ref_map1(6, a)
    df = np.array([1.0, 2.0, 6.0, 14.0, 22.0, 42.0, 82.0, 162.0, 0.5])
    xx = np.concatenate([np.logspace(-3, 3.2, 60), [0.0, -1.0, 1.0]])
    D, XX = [v.ravel().copy() for v in np.meshgrid(df, xx, indexing="ij")]
    ch = np.empty_like(D)
    REFC.ref_chdtrc_v(D, XX, D.size, ch)
    out["ch_df"], out["ch_x"], out["ch_val"] = D, XX, ch
    save("window.npz", **out)


# ---------------------------------------------------------------- G5: config 1 end to end
class _Reads(object):
    def __init__(self, plus, minus):
        self.p, self.m = plus, minus

    def __getitem__(self, iv):
        return {"+": self.p, "-": self.m}


class _Fasta(object):
    def __init__(self, seq):
        self.seq = seq

    def fetch(self, chrom, start, end):
        assert end - start == len(self.seq)
        return self.seq


def g5(bm):
    n_iv, L, hw, shw, clip = 1000, 500, 5, 50, 0.01
    scales = (3, 5, 10, 20, 40)
    n_full = 16
    pad = hw + shw
    l = L + 2 * pad + 1
    dm = make_dm("A")
    full = {k: [] for k in ("exp", "obs", "p")}
    full_w = []
    sums = np.zeros((n_iv, 3 + len(scales)))
    nans = np.zeros((n_iv, 1 + len(scales)), np.int32)
    sha = hashlib.sha256()
    sha_p = hashlib.sha256()
    for i in range(n_iv):
        cp = synth(1, i * l, l, 0)
        cm = synth(1, i * l, l, 1)
        sq = synth(1, i * (l + 6), l + 6, 2).tobytes().decode()
        pr = predict.prediction(_Reads(cp, cm), _Fasta(sq), bm, half_win_width=hw,
                                smoothing_half_win_width=shw, smoothing_clip=clip)
        iv = ref_import.genomic_interval("chr1", 1000, 1000 + L)
        obs, exp, _ = pr.compute(iv)
        obs = obs["+"][1:] + obs["-"][:-1]  # cli/detect.py:121-122
        exp = exp["+"][1:] + exp["-"][:-1]
        p = np.asarray(dm.p_values(exp, obs)).copy()
        wps = [windowing.stouffers_z(np.ascontiguousarray(p), s) for s in scales]
        sha.update(np.ascontiguousarray(exp).tobytes())
        sha.update(np.ascontiguousarray(obs).tobytes())
        sha_p.update(p.tobytes())
        for wp in wps:
            sha_p.update(np.ascontiguousarray(wp).tobytes())
        sums[i, 0], sums[i, 1], sums[i, 2] = exp.sum(), obs.sum(), np.nansum(p)
        nans[i, 0] = np.isnan(p).sum()
        for s, wp in enumerate(wps):
            sums[i, 3 + s] = np.nansum(wp)
            nans[i, 1 + s] = np.isnan(wp).sum()
        if i < n_full:
            full["exp"].append(exp)
            full["obs"].append(obs)
            full["p"].append(p)
            full_w.append(np.stack(wps))
    save("e2e_cfg1.npz", n_iv=np.array(n_iv), L=np.array(L), hw=np.array(hw), shw=np.array(shw),
         clip=np.array(clip), scales=np.array(scales), seed=np.array(1), dm_key=np.array("A"),
         exp=np.stack(full["exp"]), obs=np.stack(full["obs"]), p=np.stack(full["p"]),
         winp=np.stack(full_w), sums=sums, nan_counts=nans,
         sha256_exp_obs=np.array(sha.hexdigest()), sha256_p_winp=np.array(sha_p.hexdigest()))


# ---------------------------------------------------------------- G6: FDR helpers (A10)
def g6():
    rs = np.random.RandomState(66)
    out = {}
    nul = rs.uniform(0, 1, (500, 100)) ** 2
    pv = rs.uniform(0, 1, 500) ** 3
    pv[[3, 77]] = np.nan
    pv[[10, 11]] = pv[12]
    nul[5, 5] = np.nan
    out["null"], out["pvals"], out["efdr"] = nul, pv, fdr.emperical_fdr(nul, pv)
    a = np.sort(rs.uniform(0, 1, 300))
    b = np.sort(rs.uniform(-0.1, 1.1, 120))
    out["bis_a"], out["bis_b"], out["bis_out"] = a, b, utils.bisect(a, b)
    x = rs.uniform(0, 1, 400)
    x[100:120] = 0.001
    x[121:125] = 0.002
    x[395:] = 0.0001
    x[0:2] = 0.0
    segs = []
    for k, (thr, w, dec) in enumerate([(0.01, 3, True), (0.5, 1, False), (0.01, 1, True), (0.9, 5, False)]):
        s = utils.segment(x, thr, w, dec)
        out["seg%d" % k] = np.array(s, dtype=np.int64).reshape(-1, 2)
        segs.append((thr, w, int(dec)))
    out["seg_x"], out["seg_params"] = x, np.array(segs)
    # dispersion_model.sample (dispersion.pyx:318-355): numpy's global legacy RNG, seeded
    dm = make_dm("A")
    sx = np.round(rs.gamma(2.0, 12.0, 60))
    np.random.seed(12345)
    vals, pv = dm.sample(sx, 9)
    out["sample_x"], out["sample_vals"], out["sample_pvals"] = sx, np.asarray(vals), np.asarray(pv)
    out["sample_seed"] = np.array(12345)
    save("fdr.npz", **out)


# ---------------------------------------------------------------- G7: posterior (A11)
def g7():
    rs = np.random.RandomState(77)
    nd, n = 4, 200
    exp = np.round(rs.gamma(2.0, 10.0, (nd, n)))
    obs = rs.poisson(np.maximum(exp * rs.uniform(0.2, 1.2, (nd, n)), 0.05)).astype(float)
    fdrv = rs.uniform(0, 1, (nd, n)) ** 3
    w = (rs.uniform(0, 1, (nd, n)) > 0.2).astype(float)
    betas = rs.uniform(0.5, 4.0, (nd, 2))
    keys = ["A", "B", "C", "A"]
    dms = [make_dm(k) for k in keys]
    prior = posterior.compute_prior_weighted(fdrv, w, cutoff=0.05)
    delta = posterior.compute_delta_prior(obs, exp, fdrv, betas, cutoff=0.05)
    ll_on = posterior.log_likelihood(obs, exp, dms, delta=delta, w=3)
    ll_off = posterior.log_likelihood(obs, exp, dms, w=3)
    post = posterior.posterior(prior, ll_on, ll_off)
    save("posterior.npz", obs=obs, exp=exp, fdr=fdrv, w=w, betas=betas, dm_keys=np.array(keys),
         prior=prior, delta=delta, ll_on=ll_on, ll_off=ll_off, post=post)


# ---------------------------------------------------------------- G8: NB maximum-likelihood fit (learn_dm)
def g8():
    """nbinom.mle / nbinom.fit (stats/distributions/nbinom.pyx:25-80), the per-row fit of
    learn_dispersion_model (dispersion.pyx:391-430).  The piecewise-linear half of that function
    needs pwlf, which is not installed here, so it has no golden vector."""
    rs = np.random.RandomState(88)
    out = {}
    n_case = 0
    for (mu, r, n) in [(3.0, 2.0, 400), (12.0, 5.0, 3000), (30.0, 1.2, 5000), (0.7, 8.0, 800), (55.0, 20.0, 2500)]:
        data = np.sort(rs.negative_binomial(r, r / (r + mu), n).astype(np.float64))
        lower = int(np.floor(data.shape[0] * (2.5 / 100.0)))
        upper = int(np.ceil(data.shape[0] * (97.5 / 100.0)))
        x = data[lower:upper]
        m, v = np.mean(x), np.var(x)
        est_r = (m * m) / (v - m)
        if est_r <= 0.0:
            est_r = 10.0
        est_p = est_r / (est_r + m)
        out["data%d" % n_case] = x
        out["guess%d" % n_case] = np.array([est_p, est_r])
        out["fit%d" % n_case] = np.array(nbinom.fit(x, p=est_p, r=est_r))
        out["fit_noguess%d" % n_case] = np.array(nbinom.fit(x))
        out["mle%d" % n_case] = np.asarray(nbinom.mle(np.array([est_p, est_r]), x, np.sum(x) / len(x)))
        n_case += 1
    out["n_case"] = np.array(n_case)
    save("nbfit.npz", **out)


# ---------------------------------------------------------------- G9: output writers
def g9():
    """Text written by the reference's own writer functions (cli/utils.py:86-210) for fixed inputs.
    The module is executed from its file with a stand-in for pysam, which it imports at module
    scope and uses only in the file checks."""
    import importlib.util
    import io
    import types
    sys.modules.setdefault("pysam", types.ModuleType("pysam"))
    spec = importlib.util.spec_from_file_location(
        "footprint_tools_cli_utils", os.path.join(ref_import.REF, "footprint_tools", "cli", "utils.py"))
    cu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cu)
    iv = ref_import.genomic_interval("chr7", 1000, 1012)
    rs = np.random.RandomState(5)
    stats = np.array([[3.0, 2.0, 0.123456, np.nan, 1.0],
                      [0.0, 10.0, 33.3, 0.00004, 0.5],
                      [-1.5, 1e-300, 0.99995, 0.00005, np.inf],
                      [12345.678951, 0.5, 0.25, 1e-5, -np.inf],
                      [7.0, 7.0, 0.00005000001, 0.99994999, 2.5e-5]])
    stats = np.vstack([stats, rs.rand(7, 5) * np.array([30, 30, 1, 1, 1])])
    fdr_cols = {
        "a": np.array([1, 1, .001, .001, .002, 1, 1, 1, .0005, 1.0, 1.0, 1.0]),
        "b": np.array([.001, .001, 1, 1, 1, 1, 1, 1, 1, .002, .003, .004]),
        "c": np.array([np.nan, .001, .001, np.nan, 1, 1, .001, 1, 1, 1, 1, .001]),
        "d": np.ones(12),
        "e": np.array([.2, .3, .001, .5, .001, .6, .7, .8, .9, .001, .001, .95]),
    }
    out = {"stats": stats, "interval": np.array(["chr7", "1000", "1012"])}

    def text(fn, *a, **kw):
        buf = io.StringIO()
        fn(*a, file=buf, **kw)
        return buf.getvalue()

    out["stats_all"] = np.array(text(cu.write_stats_to_output, iv, stats))
    out["stats_filtered"] = np.array(text(cu.write_stats_to_output, iv, stats, filter_fn=lambda x: x[:, 1] >= 5))
    out["stats_fmt6e_comma"] = np.array(text(cu.write_stats_to_output, iv, stats, delim=",", fmt_string="0.6e"))
    for k, col in fdr_cols.items():
        out["fdr_" + k] = col
        out["seg_dec_" + k] = np.array(text(cu.write_segments_to_output, iv, col, 0.01, decreasing=True))
        out["seg_inc_" + k] = np.array(text(cu.write_segments_to_output, iv, col, 0.5, name="fp", score_fn=np.max))
    out["header_full"] = np.array(text(cu.write_output_header, ["exp", "obs", "-log(pval)", "-log(winpval)", "fdr"],
                                       extra=["a", "b"]))
    out["header_noname"] = np.array(text(cu.write_output_header, ["score"], include_name=False, extra="x=1"))
    out["header_plain"] = np.array(text(cu.write_output_header, ["score"], delim=" "))
    save("writers.npz", **out)


# ---------------------------------------------------------------- G10: cut counts
def g10():
    """What the reference's own cutcounts.bamfile.lookup (cutcounts.py:119-313) returns for a fixed
    list of alignments.  pysam is not installed here, so the module is executed from its file
    over a stand-in that serves the alignments the way AlignmentFile.fetch does (mapped reads
    overlapping the region, in coordinate order); the counting, pairing, filter and offset rules
    that run are the reference's."""
    import importlib.util
    import types

    def span(cigar):
        n, num = 0, ""
        for ch in cigar:
            if ch.isdigit():
                num += ch
            else:
                n += int(num) if ch in "MDN=X" else 0
                num = ""
        return n

    class Read(object):
        def __init__(self, chrom, r):
            f = r["flag"]
            self.reference_name, self.query_name = chrom, r["name"]
            self.reference_start, self.reference_end = r["pos"], r["pos"] + span(r["cigar"])
            self.mapping_quality, self.template_length = r["mapq"], r.get("tlen", 0)
            self.is_paired, self.is_proper_pair = bool(f & 1), bool(f & 2)
            self.is_reverse, self.is_read1, self.is_read2 = bool(f & 16), bool(f & 64), bool(f & 128)
            self.is_secondary, self.is_qcfail = bool(f & 256), bool(f & 512)
            self.is_duplicate, self.is_supplementary = bool(f & 1024), bool(f & 2048)

    files = {}

    class AlignmentFile(object):
        def __init__(self, filepath, mode="rb", **kw):
            self.refs, self.reads = files[filepath]

        def fetch(self, chrom, start, end):
            for r in self.reads:
                if self.refs[r["ref"]][0] == chrom:
                    x = Read(chrom, r)
                    if x.reference_start < end and x.reference_end > start:
                        yield x

        def close(self):
            pass

    fake = types.ModuleType("pysam")
    fake.AlignmentFile = AlignmentFile
    fake.VariantRecord = type("VariantRecord", (), {})
    saved = sys.modules.get("pysam")
    sys.modules["pysam"] = fake
    try:
        spec = importlib.util.spec_from_file_location(
            "footprint_tools_cutcounts", os.path.join(ref_import.REF, "footprint_tools", "cutcounts.py"))
        cc = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(cc)
    finally:
        if saved is not None:
            sys.modules["pysam"] = saved
    rs = np.random.RandomState(21)
    refs = [("chr1", 6000), ("chr2", 4000)]
    reads = []
    for k in range(6000):
        cig = str(rs.choice(["36M", "20M2D16M", "5S31M", "10M100N26M", "30M1I5M", "18=2X16=", "36M4H", "50M", "1M"]))
        flag = int(rs.choice([0, 16, 99, 147, 83, 163, 1024, 1024 + 16, 512, 256, 256 + 16, 2048 + 16, 1 + 16,
                              1 + 2 + 64 + 256, 1 + 2 + 128 + 2048, 1 + 64, 1 + 2 + 16 + 128 + 1024]))
        r = dict(ref=int(rs.randint(0, 2)), pos=int(rs.randint(0, 3900)), cigar=cig, flag=flag,
                 mapq=int(rs.choice([0, 1, 2, 30, 60, 255])), name="q%d" % k)
        reads.append(r)
        if flag in (99, 83) and rs.rand() < 0.7:  # its mate (read 2, other strand), same name, near by
            reads.append(dict(r, flag=147 if flag == 99 else 163, pos=min(3900, r["pos"] + int(rs.randint(0, 300))),
                              mapq=int(rs.choice([0, 30, 60]))))
    reads.sort(key=lambda r: (r["ref"], r["pos"]))
    files["x.bam"] = (refs, reads)
    ivs = [("chr1", 300, 800, "+"), ("chr2", 10, 60, "+"), ("chr1", 350, 420, "-"), ("chr1", 780, 2000, "+"),
           ("chr1", 3800, 3990, "+"), ("chr2", 1000, 1500, "-"), ("chr1", 0, 90, "+"), ("chr2", 3900, 3999, "+"),
           ("chr1", 5, 6, "+")]
    params = [dict(), dict(min_qual=0), dict(min_qual=31, remove_dups=True), dict(remove_qcfail=False, offset=(4, -5)),
              dict(min_qual=2, remove_dups=True, remove_qcfail=False, offset=(1, 0))]
    out = {"refs_name": np.array([r[0] for r in refs]), "refs_len": np.array([r[1] for r in refs]),
           "read_ref": np.array([r["ref"] for r in reads]), "read_pos": np.array([r["pos"] for r in reads]),
           "read_flag": np.array([r["flag"] for r in reads]), "read_mapq": np.array([r["mapq"] for r in reads]),
           "read_cigar": np.array([r["cigar"] for r in reads]), "read_name": np.array([r["name"] for r in reads]),
           "iv_chrom": np.array([v[0] for v in ivs]), "iv_start": np.array([v[1] for v in ivs]),
           "iv_end": np.array([v[2] for v in ivs]), "iv_strand": np.array([v[3] for v in ivs]),
           "params": np.array([[p.get("min_qual", 1), int(p.get("remove_dups", False)), int(p.get("remove_qcfail", True)),
                                p.get("offset", (0, -1))[0], p.get("offset", (0, -1))[1]] for p in params])}
    total = 0
    for j, p in enumerate(params):
        bf = cc.bamfile("x.bam", **p)
        plus, minus = [], []
        for c, a, b, st in ivs:
            got = bf.lookup(ref_import.genomic_interval(c, a, b, strand=st))
            plus.append(np.asarray(got["+"], np.float64))
            minus.append(np.asarray(got["-"], np.float64))
        out["plus_%d" % j], out["minus_%d" % j] = np.concatenate(plus), np.concatenate(minus)
        total += out["plus_%d" % j].sum() + out["minus_%d" % j].sum()
    assert total > 3000
    save("cutcounts.npz", **out)


# ---------------------------------------------------------------- G11: the detect driver (cli/detect.py:43-145)
def g11():
    """Per-interval statistics from the reference's own driver class, cli/detect.py `deviation_stats`
    (`__getitem__`: cutcounts.bamfile -> prediction.compute -> strand merge -> p_values ->
    stouffers_z(3) -> sample / emperical_fdr -> column_stack), on the alignments of cutcounts.npz
    and a random genome.  The module is executed from its file; what is absent from this image is
    stood in for: pysam (alignments and sequence served from memory), genome_tools' `dataset`
    base class (empty), click_option_group's decorators (identity)."""
    import importlib.util
    import tempfile
    import types
    g = np.load(os.path.join(HERE, "cutcounts.npz"))
    refs = [(str(n), int(l)) for n, l in zip(g["refs_name"], g["refs_len"])]
    reads = [dict(ref=int(a), pos=int(b), cigar=str(c), flag=int(d), mapq=int(e), name=str(f))
             for a, b, c, d, e, f in zip(g["read_ref"], g["read_pos"], g["read_cigar"], g["read_flag"],
                                         g["read_mapq"], g["read_name"])]
    rs = np.random.RandomState(33)
    genome = {name: "".join(rs.choice(list("ACGT"), n)) for name, n in refs}
    genome["chr1"] = genome["chr1"][:2000] + genome["chr1"][2000:2100].lower() + "NNNN" + genome["chr1"][2104:]

    def span(cigar):
        n, num = 0, ""
        for ch in cigar:
            if ch.isdigit():
                num += ch
            else:
                n += int(num) if ch in "MDN=X" else 0
                num = ""
        return n

    class Read(object):
        def __init__(self, chrom, r):
            f = r["flag"]
            self.reference_name, self.query_name = chrom, r["name"]
            self.reference_start, self.reference_end = r["pos"], r["pos"] + span(r["cigar"])
            self.mapping_quality, self.template_length = r["mapq"], 0
            self.is_paired, self.is_proper_pair = bool(f & 1), bool(f & 2)
            self.is_reverse, self.is_read1, self.is_read2 = bool(f & 16), bool(f & 64), bool(f & 128)
            self.is_secondary, self.is_qcfail = bool(f & 256), bool(f & 512)
            self.is_duplicate, self.is_supplementary = bool(f & 1024), bool(f & 2048)

    class AlignmentFile(object):
        def __init__(self, filepath, mode="rb", **kw):
            pass

        def fetch(self, chrom, start, end):
            for r in reads:
                if refs[r["ref"]][0] == chrom:
                    x = Read(chrom, r)
                    if x.reference_start < end and x.reference_end > start:
                        yield x

        def close(self):
            pass

    class FastaFile(object):
        def __init__(self, filepath, **kw):
            pass

        def fetch(self, chrom, start, end):
            return genome[chrom][start:end]

    fake = types.ModuleType("pysam")
    fake.AlignmentFile, fake.FastaFile = AlignmentFile, FastaFile
    fake.VariantRecord = type("VariantRecord", (), {})
    fake.set_verbosity = lambda v: None
    cog = types.ModuleType("click_option_group")
    ident = lambda *a, **k: (lambda f: f)
    cog.optgroup = types.SimpleNamespace(group=ident, option=ident)
    gtd = types.ModuleType("genome_tools.data")
    gtds = types.ModuleType("genome_tools.data.dataset")
    gtds.dataset = type("dataset", (), {})
    saved = {k: sys.modules.get(k) for k in ("pysam", "click_option_group", "genome_tools.data", "genome_tools.data.dataset")}
    sys.modules.update({"pysam": fake, "click_option_group": cog, "genome_tools.data": gtd,
                        "genome_tools.data.dataset": gtds})
    cli = types.ModuleType("footprint_tools.cli")
    cli.__path__ = [os.path.join(ref_import.REF, "footprint_tools", "cli")]
    sys.modules["footprint_tools.cli"] = cli
    gtdu = types.ModuleType("genome_tools.data.utils")
    gtdu.numpy_collate_concat = lambda x: np.concatenate(x)
    sys.modules["genome_tools.data.utils"] = gtdu
    try:
        spec = importlib.util.spec_from_file_location(
            "footprint_tools.cli.detect", os.path.join(ref_import.REF, "footprint_tools", "cli", "detect.py"))
        det = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(det)
        spec = importlib.util.spec_from_file_location(
            "footprint_tools.cli.learn_dm", os.path.join(ref_import.REF, "footprint_tools", "cli", "learn_dm.py"))
        ldm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ldm)
    finally:
        sys.modules.pop("genome_tools.data.utils", None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    ivs = [("chr1", 300, 800), ("chr1", 1900, 2300), ("chr2", 100, 237), ("chr1", 2500, 3750), ("chr2", 1000, 1064)]
    bm = bias.kmer_model(os.path.join(ref_import.REF, "data", "vierstra_et_al.6mer-model.txt"))
    dm = make_dm("A")
    kw = dict(min_qual=1, remove_dups=True, remove_qcfail=True, offset=(0, -1), half_win_width=5, is_cram=False,
              fasta_reference_filepath="g.fa", smoothing_half_win_width=50, smoothing_clip=0.01, fdr_shuffle_n=50)
    with tempfile.NamedTemporaryFile("w", suffix=".bed", delete=False) as f:
        for c, a, b in ivs:
            f.write("%s\t%d\t%d\n" % (c, a, b))
        bed = f.name
    try:
        np.random.seed(7)  # cli/detect.py:348-352
        ds = det.deviation_stats(bed, "x.bam", "g.fa", bm, dm, **kw)
        out = {"genome_" + n: np.frombuffer(sq.encode(), np.uint8) for n, sq in genome.items()}
        out["iv_chrom"] = np.array([v[0] for v in ivs])
        out["iv_start"] = np.array([v[1] for v in ivs])
        out["iv_end"] = np.array([v[2] for v in ivs])
        assert len(ds) == len(ivs)
        for i in range(len(ivs)):
            rec = ds[i]
            assert (rec["interval"].chrom, rec["interval"].start, rec["interval"].end) == ivs[i]
            out["stats_%d" % i] = np.asarray(rec["stats"], np.float64)
        ds_nodm = det.deviation_stats(bed, "x.bam", "g.fa", bm, None, **kw)  # no dispersion model: two columns
        out["stats_nodm_0"] = np.asarray(ds_nodm[0]["stats"], np.float64)
        # cli/learn_dm.py:30-107: the same chain with the predictor's class defaults (no smoothing)
        ec = ldm.expected_counts(bed, "x.bam", "g.fa", bm, min_qual=1, remove_dups=True, remove_qcfail=True,
                                 offset=(0, -1), half_win_width=5)
        for i in range(len(ivs)):
            out["learn_cnts_%d" % i] = np.asarray(ec[i], np.float64)
    finally:
        os.remove(bed)
    assert out["stats_0"].shape == (500, 5) and out["stats_0"][:, 1].sum() > 20
    save("detect_driver.npz", **out)


# ---------------------------------------------------------------- G12: the posterior driver (cli/post.py:40-124)
def g12():
    """Records of the reference's own `posterior_stats.__getitem__` (tracks -> _load_data -> priors ->
    windowed log-likelihoods -> posterior) for three datasets, and the JSON the reference's
    write_dispersion_model wrote for their models.  Stand-ins as in g11; pysam.TabixFile serves the
    track rows from memory the way fetch(..., parser=asTuple()) does."""
    import importlib.util
    import tempfile
    import types
    import pandas as pd
    rs = np.random.RandomState(44)
    keys = ["A", "C", "B"]
    tracks, out = {}, {}
    for d, key in enumerate(keys):
        rows = []
        for x in range(2000, 2400):
            if rs.rand() < (0.0, 0.1, 0.3)[d]:
                continue  # this dataset has no row here (w = 0)
            e = float(rs.randint(0, 40))
            o = float(max(0, int(e * rs.uniform(0.1, 1.6))))
            f = float(rs.rand() ** 4)
            rows.append(("chr7", x, "%.4f" % e, "%.4f" % o, "%.4f" % f))
        tracks["t%d.gz" % d] = rows
        out["track%d_pos" % d] = np.array([r[1] for r in rows])
        out["track%d_exp" % d] = np.array([float(r[2]) for r in rows])
        out["track%d_obs" % d] = np.array([float(r[3]) for r in rows])
        out["track%d_fdr" % d] = np.array([float(r[4]) for r in rows])

    class TabixFile(object):
        def __init__(self, fn):
            self.rows = tracks[os.path.basename(fn)]

        def fetch(self, chrom, start, end, parser=None):
            for c, x, e, o, f in self.rows:
                if c == chrom and x < end and x + 1 > start:
                    yield (c, str(x), str(x + 1), e, o, "0.5000", "0.5000", f)

        def close(self):
            pass

    fake = types.ModuleType("pysam")
    fake.TabixFile, fake.asTuple = TabixFile, (lambda: None)
    fake.set_verbosity = lambda v: None
    cog = types.ModuleType("click_option_group")
    ident = lambda *a, **k: (lambda f: f)
    cog.optgroup = types.SimpleNamespace(group=ident, option=ident)
    gtd = types.ModuleType("genome_tools.data")
    gtds = types.ModuleType("genome_tools.data.dataset")
    gtds.dataset = type("dataset", (), {})
    names = ("pysam", "click_option_group", "genome_tools.data", "genome_tools.data.dataset", "footprint_tools.cli")
    saved = {k: sys.modules.get(k) for k in names}
    cli = types.ModuleType("footprint_tools.cli")
    cli.__path__ = [os.path.join(ref_import.REF, "footprint_tools", "cli")]
    sys.modules.update({"pysam": fake, "click_option_group": cog, "genome_tools.data": gtd,
                        "genome_tools.data.dataset": gtds, "footprint_tools.cli": cli})
    try:
        spec = importlib.util.spec_from_file_location(
            "footprint_tools.cli.post", os.path.join(ref_import.REF, "footprint_tools", "cli", "post.py"))
        post = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(post)
    finally:
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)
    tmp = tempfile.mkdtemp()
    dm_files = []
    # simplejson (absent here) writes bytes values as utf-8 text, which is what makes the
    # reference's base64 fields serialisable; the stdlib module standing in for it is told to
    import json as _json
    sj = types.ModuleType("simplejson")
    sj.load, sj.loads = _json.load, _json.loads
    sj.dumps = lambda o, **kw: _json.dumps(o, default=lambda v: v.decode("utf-8"), **kw)
    sys.modules["simplejson"] = sj
    import datetime as _dt
    real_datetime = _dt.datetime

    class _frozen(real_datetime):  # the writer stamps the file with datetime.now(): keep the fixture reproducible
        @classmethod
        def now(cls, tz=None):
            return cls(2024, 1, 1, 0, 0, 0)

    _dt.datetime = _frozen
    for d, key in enumerate(keys):
        model = make_dm(key)  # a learned model also carries its histogram and the per-row fits
        model.h = rs.randint(0, 50, (6, 9)).astype(np.float64)
        model.p, model.r = rs.rand(6), rs.rand(6) * 10
        text = dispersion.write_dispersion_model(model, extra="dataset %d" % d)
        out["dm_json_%d" % d] = np.array(text)
        out["dm_h_%d" % d], out["dm_p_%d" % d], out["dm_r_%d" % d] = np.array(model.h), np.array(model.p), np.array(model.r)
        fn = os.path.join(tmp, "dm%d.json" % d)
        with open(fn, "w") as f:
            f.write(text)
        dm_files.append(fn)
    _dt.datetime = real_datetime
    betas = [(2.0, 5.0), (2.0, 6.5), (1.5, 4.0)]
    samples = pd.DataFrame({"id": ["s%d" % d for d in range(3)], "tabix_file": [os.path.join(tmp, "t%d.gz" % d) for d in range(3)],
                            "dm_file": dm_files, "beta_a": [b[0] for b in betas], "beta_b": [b[1] for b in betas]})
    ivs = [("chr7", 1990, 2310), ("chr7", 2350, 2420), ("chr7", 2100, 2101), ("chr8", 10, 40)]
    bed = os.path.join(tmp, "iv.bed")
    with open(bed, "w") as f:
        f.write("# comment line\n")
        for c, a, b in ivs:
            f.write("%s\t%d\t%d\n" % (c, a, b))
    ps = post.posterior_stats(bed, samples, 0.05)
    assert len(ps) == len(ivs)
    out["betas"], out["fdr_cutoff"] = np.array(betas), np.array(0.05)
    out["iv_chrom"] = np.array([v[0] for v in ivs])
    out["iv_start"], out["iv_end"] = np.array([v[1] for v in ivs]), np.array([v[2] for v in ivs])
    for i in range(len(ivs)):
        rec = ps[i]
        out["stats_%d" % i] = np.asarray(rec["stats"], np.float64)
    assert out["stats_0"].shape == (320, 3) and (out["stats_0"] > 0).any()
    save("post_driver.npz", **out)


# ---------------------------------------------------------------- G13: rounding ties of fast_predict
def g13():
    """Rows on which round() in predict.h:62 is decided by the rounding noise of the reference's own
    trimmed sum (smoothing.h:72-99: inexact Beliakov weights such as 2/3, elements added in the
    order the two selections left them in): equal propensities (a homopolymer: P/Q = 1/10),
    alternating 1/0 counts (window sums of 5, so T/99 * 1/10 sits on 0.5) with a few bumps that tie
    the window extrema three and more times.  An evaluation from the exact trimmed sum gives a
    different integer on dozens of positions per row (counted in `n_flip`)."""
    rs = np.random.RandomState(1313)
    bm_p = 0.00437  # any value: every propensity of the row is the same
    out, meta = {}, []
    c = 0

    def exact_exp(obs, probs, hw, shw, k):
        l = obs.size
        w = 2 * shw + 1
        W, Q = np.zeros(l), np.zeros(l)
        for i in range(hw, l - hw):
            for j in range(-hw, hw):
                W[i] += obs[i + j]
                Q[i] += probs[i + j]
        sm = np.zeros(l)
        for i in range(shw, l - shw):
            x = np.sort(W[i - shw:i + shw + 1])
            sm[i] = x[k:w - k].sum() / (w - 2 * k)
        e = np.zeros(l)
        for i in range(hw, l - hw):
            v = probs[i] / Q[i] * sm[i]
            e[i] = np.floor(v + 0.5)
        return e

    for (hw, shw, clip, l, kind) in [(5, 50, .01, 1111, "alt"), (5, 50, .01, 1111, "alt"), (5, 50, .01, 611, "alt"),
                                     (5, 50, .01, 611, "alt3"), (5, 50, .01, 611, "five"), (5, 50, .05, 611, "alt"),
                                     (3, 20, .05, 400, "alt"), (5, 50, .01, 1111, "alt_rand_p")]:
        base = np.tile([1.0, 0.0], l)[:l]
        obs = base.copy()
        if kind == "five":
            obs = np.full(l, 5.0)
            obs[rs.randint(0, l, 6)] += 1.0
        else:
            idx = rs.randint(0, l, 30 if l > 700 else 16)
            obs[idx] += rs.choice([-1.0, 1.0, 1.0, 2.0], idx.size) * (base[idx] > 0)
            obs = np.maximum(obs, 0.0)
            if kind == "alt3":
                obs *= 3.0
        probs = np.full(l, bm_p)
        if kind == "alt_rand_p":  # ordinary propensities around a homopolymer stretch
            probs = rs.uniform(3e-4, 0.22, l)
            probs[200:700] = bm_p
        e, w = ref_predict_c(obs, probs, hw, shw, clip)
        k = int((2 * shw + 1) * clip)
        n_flip = int((exact_exp(obs, probs, hw, shw, k) != e).sum())
        out["obs%d" % c], out["probs%d" % c] = obs, probs
        out["exp%d" % c], out["win%d" % c] = e, w
        meta.append((hw, shw, clip, l, n_flip))
        c += 1
    out["meta"] = np.array(meta, dtype=np.float64)
    assert sum(m[4] for m in meta) > 50, meta
    print("ties: positions where the exact trimmed sum rounds differently:", [m[4] for m in meta])
    save("predict_ties.npz", **out)


# ---------------------------------------------------------------- G14: Storey q-values (stats/fdr/__init__.py:39-95)
def g14():
    """pi0est and qvalue on p-value sets of several shapes.  (bh_qvalue, :98-131, cannot be run: its
    sorted(iterable, None, key) is Python 2 -- TypeError under Python 3 -- so no expected values exist for it.)"""
    rs = np.random.RandomState(140)
    out = {}
    sets = [rs.uniform(0, 1, 400) ** 2, rs.uniform(0, 1, 1000), np.concatenate([rs.uniform(0, 1e-3, 50), rs.uniform(0, 1, 450)]),
            rs.beta(0.5, 2.0, 250), np.round(rs.uniform(0, 1, 300), 2), rs.uniform(0.2, 1, 120)]
    for k, pv in enumerate(sets):
        out["p%d" % k] = pv
        out["pi0_%d" % k] = np.asarray(fdr.pi0est(pv), dtype=np.float64).ravel()
        out["q%d" % k] = np.asarray(fdr.qvalue(pv.copy()), dtype=np.float64)
    lamb = np.arange(0.1, 0.9, 0.1)
    out["lamb"] = lamb
    out["pi0_lamb"] = np.asarray(fdr.pi0est(sets[0], lamb), dtype=np.float64).ravel()
    try:
        fdr.bh_qvalue(np.sort(sets[0]))
        out["bh_runs"] = np.array(1)
    except TypeError:
        out["bh_runs"] = np.array(0)
    save("qvalues.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["1", "2", "3", "4", "5", "6", "7", "8", "9", "10", "11", "12", "13", "14"]
    bm, table = g1() if ("1" in which or "5" in which) else (None, None)
    if "2" in which:
        g2()
    if "3" in which:
        g3()
    if "4" in which:
        g4()
    if "5" in which:
        g5(bm)
    if "6" in which:
        g6()
    if "7" in which:
        g7()
    if "8" in which:
        g8()
    if "9" in which:
        g9()
    if "10" in which:
        g10()
    if "11" in which:
        g11()
    if "12" in which:
        g12()
    if "13" in which:
        g13()
    if "14" in which:
        g14()
