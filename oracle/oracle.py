"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU checker.

`oracle/libfpt_oracle.so` is the C restatement (oracle/fpt_oracle.c) and
`oracle/_ref/libfpt_ref.so` (optional) is the reference's own native code
compiled from /root/reference.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this module; the product package
(footprint_tools_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")

WIN_OPS = {"sum": 0, "product": 1, "fishers_combined": 2, "stouffers_z": 3,
           "weighted_stouffers_z": 4}
MAP1 = {"gamma": 0, "lgam": 1, "ndtr": 2, "ndtri": 3, "log1p": 4, "erf": 5, "erfc": 6}

# canonical synthetic dispersion model (DM-SYNTH-A)
DM_SYNTH_A = dict(
    mu=np.array([25, 50, 75, 0, 0.5, 1.0, 1.0, 0.98, 0.97], dtype=np.float64),
    r=np.array([3, 7, 15, 25, 75, 0.05, 0.08, 0.115, 0.16, 0.185, 0.02, 0.01, 0.005, 0.002,
                0.001], dtype=np.float64))


def build(force=False):
    """Compile the checker(s) with the committed Makefile.

    Only spawns `make` when the library is missing (or force=True): a process that has
    already initialised the GPU must not fork/exec on the GPU pool, so callers load the oracle
    BEFORE touching the device and __graft_entry__.build() prebuilds it."""
    so = os.path.join(HERE, "libfpt_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_map1.argtypes = [C.c_int, f64p, C.c_int64, f64p]
        L.orc_incbet_v.argtypes = [f64p, f64p, f64p, C.c_int64, f64p]
        L.orc_chdtrc_v.argtypes = [f64p, f64p, C.c_int64, f64p]
        L.orc_kmer_probs.argtypes = [u8p, C.c_int64, f64p, C.c_double, f64p, f64p, i32p, i32p]
        L.orc_fast_predict.argtypes = [f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_double, f64p, f64p]
        L.orc_fit_mu.argtypes = [f64p, C.c_double]
        L.orc_fit_mu.restype = C.c_double
        L.orc_fit_r.argtypes = [f64p, C.c_double, C.POINTER(C.c_double)]
        L.orc_nb_values.argtypes = [C.c_int, f64p, f64p, f64p, f64p, C.c_int64, f64p]
        for nm in ("orc_nb_cdf", "orc_nb_logpmf", "orc_nb_pmf"):
            getattr(L, nm).argtypes = [C.c_int32, C.c_double, C.c_double]
            getattr(L, nm).restype = C.c_double
        L.orc_window.argtypes = [C.c_int, f64p, C.c_void_p, C.c_int, C.c_int, f64p]
        L.orc_bisect.argtypes = [f64p, C.c_int, f64p, C.c_int, f64p]
        L.orc_emperical_fdr.argtypes = [f64p, C.c_int64, f64p, C.c_int, f64p]
        L.orc_segment.argtypes = [f64p, C.c_int, C.c_double, C.c_int, C.c_int, i32p, C.c_int]
        L.orc_hist2d.argtypes = [f64p, f64p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.orc_hist2d.restype = C.c_int64
        L.orc_detect_batch.argtypes = [f64p, f64p, u8p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                       C.c_double, f64p, C.c_double, f64p, f64p, i32p, C.c_int,
                                       f64p, f64p, f64p, f64p, C.c_int]
        L.orc_log_likelihood_row.argtypes = [f64p, f64p, f64p, f64p, f64p, C.c_int, C.c_int, f64p]
        L.orc_philox_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_philox_uniform.restype = C.c_double
        L.orc_fdr_null.argtypes = [f64p, f64p, f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64,
                                   C.c_void_p, C.c_int, C.c_int, f64p, C.c_void_p]
        L.orc_synth_fill.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                     C.c_void_p]
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own native sources (None when oracle/_ref was not built)."""
    global _REF
    if _REF is None:
        so = os.path.join(HERE, "_ref", "libfpt_ref.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        R.ref_fast_predict.argtypes = [f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_double, f64p, f64p]
        R.ref_window.argtypes = [C.c_int, f64p, C.c_void_p, C.c_int, C.c_int, f64p]
        R.ref_map1.argtypes = [C.c_int, f64p, C.c_long, f64p]
        R.ref_incbet_v.argtypes = [f64p, f64p, f64p, C.c_long, f64p]
        R.ref_chdtrc_v.argtypes = [f64p, f64p, C.c_long, f64p]
        _REF = R
    return _REF


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def map1(name, x, ref=False):
    x = _f(x).ravel()
    out = np.empty_like(x)
    if ref:
        ref_lib().ref_map1(MAP1[name], x, x.size, out)
    else:
        lib().orc_map1(MAP1[name], x, x.size, out)
    return out


def incbet(a, b, x, ref=False):
    a, b, x = np.broadcast_arrays(_f(a), _f(b), _f(x))
    a, b, x = _f(a).ravel(), _f(b).ravel(), _f(x).ravel()
    out = np.empty_like(a)
    (ref_lib().ref_incbet_v if ref else lib().orc_incbet_v)(a, b, x, a.size, out)
    return out


def chdtrc(df, x, ref=False):
    df, x = np.broadcast_arrays(_f(df), _f(x))
    df, x = _f(df).ravel(), _f(x).ravel()
    out = np.empty_like(x)
    (ref_lib().ref_chdtrc_v if ref else lib().orc_chdtrc_v)(df, x, x.size, out)
    return out


def seq_bytes(seq):
    if isinstance(seq, str):
        seq = seq.encode("ascii")
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8).copy()
    return np.ascontiguousarray(seq, dtype=np.uint8)


def kmer_probs(seq, table, dflt=1e-6):
    s = seq_bytes(seq)
    l = max(s.size - 6, 0)
    fwd, rev = np.empty(l), np.empty(l)
    fi, ri = np.empty(l, np.int32), np.empty(l, np.int32)
    lib().orc_kmer_probs(s, s.size, _f(table), dflt, fwd, rev, fi, ri)
    return fwd, rev, fi, ri


def fast_predict(obs, probs, hw, shw, clip, ref=False):
    obs, probs = _f(obs), _f(probs)
    l = obs.size
    e, w = np.empty(l), np.empty(l)
    (ref_lib().ref_fast_predict if ref else lib().orc_fast_predict)(obs, probs, l, hw, shw, clip, e, w)
    return e, w


def fit_mu(mu_par, x):
    return lib().orc_fit_mu(_f(mu_par), float(x))


def fit_r(r_par, x):
    out = C.c_double()
    if lib().orc_fit_r(_f(r_par), float(x), C.byref(out)):
        raise ZeroDivisionError("float division")
    return out.value


def nb_values(what, mu_par, r_par, exp, obs):
    exp, obs = _f(exp), _f(obs)
    out = np.empty(exp.size)
    code = {"cdf": 0, "logpmf": 1, "pmf": 2}[what]
    if lib().orc_nb_values(code, _f(mu_par), _f(r_par), exp, obs, exp.size, out):
        raise ZeroDivisionError("float division")
    return out


def window(op, x, hw, w=None, ref=False):
    x = _f(x)
    out = np.empty(x.size)
    wp = None
    if w is not None:
        w = _f(w)
        wp = w.ctypes.data
    (ref_lib().ref_window if ref else lib().orc_window)(WIN_OPS[op], x, wp, x.size, hw, out)
    return out


def bisect(a, b):
    a, b = _f(a), _f(b)
    out = np.empty(b.size)
    lib().orc_bisect(a, a.size, b, b.size, out)
    return out


def emperical_fdr(pvals_null, pvals):
    nul, p = _f(pvals_null).ravel(), _f(pvals)
    out = np.empty(p.size)
    lib().orc_emperical_fdr(nul, nul.size, p, p.size, out)
    return out


def segment(x, threshold, w=1, decreasing=False):
    x = _f(x)
    seg = np.empty(2 * (x.size + 1), np.int32)
    n = lib().orc_segment(x, x.size, threshold, w, int(bool(decreasing)), seg, x.size + 1)
    return seg[:2 * n].reshape(n, 2).tolist()


def exp_obs_histogram(exp, obs, dims=(200, 1000)):
    """cli/learn_dm.py:276-287: hist[int(exp), int(obs)] += 1, IndexError ignored"""
    e, o = _f(exp).ravel(), _f(obs).ravel()
    hist = np.zeros(dims, dtype=np.int64)
    lib().orc_hist2d(e, o, e.size, int(dims[0]), int(dims[1]), hist.ctypes.data)
    return hist


def detect_batch(counts_plus, counts_minus, seq, n_iv, L, hw, shw, clip, table, mu_par, r_par,
                 scales, dflt=1e-6, n_threads=1):
    """Equal-length batch; returns exp, obs, p (n_iv*L) and winp (S, n_iv*L)."""
    scales = np.ascontiguousarray(scales, dtype=np.int32)
    tot = n_iv * L
    e, o, p = np.empty(tot), np.empty(tot), np.empty(tot)
    wp = np.empty((scales.size, tot))
    rc = lib().orc_detect_batch(_f(counts_plus).ravel(), _f(counts_minus).ravel(),
                                seq_bytes(seq).ravel(), n_iv, L, hw, shw, clip, _f(table), dflt,
                                _f(mu_par), _f(r_par), scales, scales.size, e, o, p, wp, n_threads)
    if rc:
        raise ZeroDivisionError("float division")
    return e, o, p, wp


def log_likelihood_row(mu_par, r_par, obs, exp, delta, w):
    obs, exp = _f(obs), _f(exp)
    delta = _f(np.broadcast_to(delta, obs.shape))
    out = np.empty(obs.size)
    if lib().orc_log_likelihood_row(_f(mu_par), _f(r_par), obs, exp, delta, obs.size, w, out):
        raise ZeroDivisionError("float division")
    return out


def posterior_stats(obs, exp, fdr, w, betas, models, cutoff=0.05, hw=3, pseudocount=0.5):
    """cli/post.py:109-122 for one interval: the reference's own sequence of numpy operations
    (stats/posterior.py:12-149) with its scipy call written out -- scipy.stats.beta.stats(a, b,
    moments="mv") is a/(a+b), ab/((a+b)^2 (a+b+1)), NaN outside a, b > 0 -- and its
    dm.log_pmf_values / windowing.sum through the C restatement.  models: (mu_par, r_par) per
    dataset.  Returns stats (bases x datasets) and the pieces (prior, delta, ll_on, ll_off)."""
    obs, exp, fdr, w = (np.asarray(a, dtype=np.float64) for a in (obs, exp, fdr, w))
    betas = np.asarray(betas, dtype=np.float64)
    n, m = obs.shape
    with np.errstate(all="ignore"):
        # posterior.py:31-42
        k = np.sum(fdr <= cutoff, axis=0)
        nn = np.sum(w, axis=0)
        a = nn - k + pseudocount
        b = k + pseudocount
        pr = a / (a + b)
        prior = np.ones(fdr.shape)
        prior *= pr[np.newaxis, :]
        prior[w == 0] = 1
        # posterior.py:66-90
        mus, ws = np.ones((n, m)), np.ones((n, m))
        for i in range(n):
            kk = obs[i, :]
            trials = np.max(np.vstack([exp[i, :], obs[i, :]]), axis=0)
            al, be = kk + betas[i][0], trials - kk + betas[i][1]
            s = al + be
            ok = (al > 0) & (be > 0)
            mu = np.where(ok, al / s, np.nan)
            v = np.where(ok, al * be / (s ** 2 * (s + 1)), np.nan)
            mus[i, :] = mu
            ws[i, :] = 1 / np.sqrt(v)
        ws[fdr > cutoff] = 0
        delta = np.sum(ws * mus, axis=0) / np.sum(ws, axis=0)
        delta[np.isnan(delta)] = 1
        # posterior.py:114-121
        ll_on, ll_off = np.ones((n, m)), np.ones((n, m))
        for i in range(n):
            ll_on[i, :] = log_likelihood_row(models[i][0], models[i][1], obs[i], exp[i], delta, hw)
            ll_off[i, :] = log_likelihood_row(models[i][0], models[i][1], obs[i], exp[i], 1.0, hw)
        # posterior.py:140-149, post.py:121-122
        p_off = np.log(prior) + ll_off
        p_on = np.log(1 - prior) + ll_on
        post = -(p_off - np.logaddexp(p_on, p_off))
        post[post <= 0] = 0.0
    return post.T, dict(prior=prior, delta=delta, ll_on=ll_on, ll_off=ll_off)


def null_alias_row(mu_par, r_par, ex, table_k=2048):
    """The alias table of the null sampler at the integer expected value `ex`: (lg, entries[2^lg], cdf[2^lg])."""
    L = lib()
    L.orc_null_alias_row.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_null_alias_row.restype = C.c_int
    ent, cdf = np.zeros(2048, np.uint32), np.zeros(2048)
    mu_par, r_par = _f(mu_par), _f(r_par)
    lg = L.orc_null_alias_row(mu_par.ctypes.data, r_par.ctypes.data, float(ex), int(table_k), ent.ctypes.data, cdf.ctypes.data)
    return lg, ent[:1 << lg].copy(), cdf[:1 << lg].copy()


def null_draws(mu_par, r_par, ex, u, table=(256, 2048)):
    """Draws of the null sampler at one expected value from the uniforms `u`: (outcome or -1, cdf of the outcome)."""
    L = lib()
    L.orc_null_draws.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_null_draws.restype = None
    u = _f(u)
    k, p = np.empty(u.size, np.int32), np.empty(u.size)
    mu_par, r_par = _f(mu_par), _f(r_par)
    L.orc_null_draws(mu_par.ctypes.data, r_par.ctypes.data, float(ex), u.ctypes.data, u.size, table[0], table[1],
                     k.ctypes.data, p.ctypes.data)
    return k, p


def fdr_null(mu_par, r_par, exp, winp, hw, times, seed, base0=0, uniforms=None, table=(256, 2048),
             return_null=False):
    """One interval of the reproducible empirical-FDR pass (see fpt_fdr_dev in include/fpt.h).  `table`: the
    library's table dimensions (the alias tables of the draws depend on the second)."""
    exp, winp = _f(exp), _f(winp)
    L = exp.size
    out = np.empty(L)
    nul = np.empty((L, times)) if return_null else None
    up = None
    if uniforms is not None:
        uniforms = _f(uniforms)
        up = uniforms.ctypes.data
    lib().orc_fdr_null(_f(mu_par), _f(r_par), exp, winp, L, hw, times, seed, base0, up, table[0], table[1],
                       out, nul.ctypes.data if return_null else None)
    return (out, nul) if return_null else out


def synth_counts(seed, pos0, n, stream):
    out = np.empty(n)
    lib().orc_synth_fill(seed, pos0, n, stream, out.ctypes.data, None)
    return out


def synth_hotspots(counts, seed, pos0, stream, padded_len, per_mille):
    """Add the hotspot bursts of the heavy-tailed workload to `counts` (in place; stream 0 / 1)."""
    L = lib()
    L.orc_synth_hotspots.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_synth_hotspots(seed, pos0, counts.size, stream, padded_len, per_mille, counts.ctypes.data)
    return counts


def synth_bases(seed, pos0, n):
    out = np.empty(n, np.uint8)
    lib().orc_synth_fill(seed, pos0, n, 2, None, out.ctypes.data)
    return out
