"""diagnostic: rerun one seed of test_fused_scan_fuzz and print where it differs"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as orc  # noqa: E402  (checker first)
orc.lib()
from footprint_tools_amd.scan import FootprintScanner  # noqa: E402

seed = int(sys.argv[1])
rs = np.random.RandomState(1000 + seed)
lat = np.load("tests/golden/nb_lattice.npz")
table = np.load("tests/golden/kmer_probs.npz")["table"]


class DM(object):
    def __init__(self, mu, r):
        self.mu_params, self.r_params = mu, r


hw = int(rs.choice([1, 2, 3, 5, 5, 8]))
shw = int(rs.choice([0, 1, 7, 31, 32, 50, 50, 64, 65, 120]))
clip = float(rs.choice([0.0, 0.005, 0.01, 0.01, 0.02, 0.05, 0.2]))
if shw and int((2 * shw + 1) * clip) * 2 >= 2 * shw + 1:
    clip = 0.01
n_sc = int(rs.randint(0, 4))
scales = tuple(int(x) for x in rs.choice([0, 1, 3, 3, 5, 10, 33, 70], n_sc, replace=False))
dm = str(rs.choice(["A", "B", "C"]))
mode = str(rs.choice(["direct", "memo", "auto"]))
pad = hw + shw
n_iv = int(rs.randint(3, 14))
lens = rs.choice([1, 2, 5, 17, 63, 64, 65, 200, 500, 700, 1024, 1025, 1500, 2300], n_iv)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
n_c = int(off[-1] + n_iv * (2 * pad + 1))
kind = str(rs.choice(["dense", "sparse", "runs", "float", "huge", "neg"]))
if kind == "dense":
    cp, cm = rs.randint(0, 20, n_c).astype(float), rs.randint(0, 20, n_c).astype(float)
elif kind == "sparse":
    cp, cm = rs.poisson(0.05, n_c).astype(float), rs.poisson(0.3, n_c).astype(float)
elif kind == "runs":
    cp = np.repeat(rs.randint(0, 4, n_c // 37 + 1), 37)[:n_c].astype(float)
    cm = np.repeat(rs.randint(0, 2, n_c // 150 + 1), 150)[:n_c].astype(float)
elif kind == "float":
    cp, cm = rs.gamma(1.5, 2.0, n_c), rs.gamma(0.5, 3.0, n_c)
elif kind == "huge":
    cp, cm = rs.randint(0, 20, n_c).astype(float), rs.randint(0, 5, n_c).astype(float)
    cp[rs.randint(0, n_c, 5)] = 2.0 ** rs.randint(20, 26, 5)
else:
    cp, cm = rs.randint(-2, 6, n_c).astype(float), rs.randint(0, 6, n_c).astype(float)
sq = rs.choice(np.frombuffer(b"ACGTACGTACGTacgtN", np.uint8), int(off[-1] + n_iv * (2 * pad + 7)))
print("hw", hw, "shw", shw, "clip", clip, "scales", scales, dm, mode, kind, lens.tolist())
sc = FootprintScanner(table, DM(lat["mu_" + dm], lat["r_" + dm]), hw, shw, clip, scales, nb_mode=mode)
out = sc.scan(cp, cm, sq, interval_off=off)
for i, L in enumerate(lens):
    a, b = off[i] + i * (2 * pad + 1), off[i + 1] + (i + 1) * (2 * pad + 1)
    sa, sb = off[i] + i * (2 * pad + 7), off[i + 1] + (i + 1) * (2 * pad + 7)
    e, o, p, wp = orc.detect_batch(cp[a:b], cm[a:b], sq[sa:sb], 1, int(L), hw, shw, clip, table,
                                   lat["mu_" + dm], lat["r_" + dm], np.array(scales, np.int32))
    sl = slice(off[i], off[i + 1])
    with np.errstate(all="ignore"):
        rp = np.abs(out["pval"][sl] - p) / np.abs(p)
    bp = np.flatnonzero(rp > 1e-7)
    if bp.size:
        t = bp[0]
        print("interval", i, "p mismatch n", bp.size, "at", bp[:5], "exp", e[bp[:5]], "obs", o[bp[:5]], "got", repr(out["pval"][sl][t]), "want", repr(p[t]), "rel", rp[t])
    for s_i in range(len(scales)):
        with np.errstate(all="ignore"):
            rw = np.abs(out["winp"][s_i, sl] - wp[s_i]) / np.abs(wp[s_i])
        bw = np.flatnonzero(rw > 1e-7)
        if bw.size:
            t = bw[0]
            print("interval", i, "scale", scales[s_i], "winp mismatch n", bw.size, "at", bw[:5], "got", repr(out["winp"][s_i, sl][t]), "want", repr(wp[s_i][t]), "rel", rw[t], "p around", p[max(0,t-3):t+4])
    bad = np.flatnonzero(~((out["exp"][sl] == e) | (np.isnan(out["exp"][sl]) & np.isnan(e))))
    if bad.size:
        print("interval", i, "L", L, "n_bad", bad.size, "first", bad[:10], "got", out["exp"][sl][bad[:10]], "want", e[bad[:10]])
        t = bad[0]
        print("  counts+ around:", cp[a + t + pad + 1 - 8: a + t + pad + 1 + 8])
        print("  max |cp| in interval", np.abs(cp[a:b]).max(), "max cm", np.abs(cm[a:b]).max())

if len(sys.argv) > 3:
    i, t = int(sys.argv[2]), int(sys.argv[3])
    L = int(lens[i])
    a, b = off[i] + i * (2 * pad + 1), off[i + 1] + (i + 1) * (2 * pad + 1)
    sa, sb = off[i] + i * (2 * pad + 7), off[i + 1] + (i + 1) * (2 * pad + 7)
    fwd, rev = orc.kmer_probs(sq[sa:sb], table)[:2]
    l = b - a
    for name, c, pr, v in (("+", cp[a:b], fwd[:l], pad + 1 + t), ("-", cm[a:b], rev[:l], pad + t)):
        e, w = orc.fast_predict(c, pr, hw, shw, clip)
        q = sum(pr[v + j] for j in range(-hw, hw))
        W = np.array([sum(c[u + j] for j in range(-hw, hw)) for u in range(v - shw, v + shw + 1)])
        srt = np.sort(W)
        print(name, "E", e[v], "win", repr(w[v]), "p/q", repr(pr[v] / q), "prod", repr(pr[v] / q * w[v]),
              "S-min-max", repr((W.sum() - srt[0] - srt[-1]) / (len(W) - 2)), "mid", repr(srt[1:-1].sum() / (len(W) - 2)))
