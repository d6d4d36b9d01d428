"""Sequence-bias models.  Mirrors footprint_tools/modeling/bias.py (v1.3.7); the per-base
lookup runs on the GPU from a 4096-entry table staged in LDS."""
import itertools
import random

import numpy as np

from .. import _lib

_ACGT = "ACGT"
_CODE = {c: i for i, c in enumerate(_ACGT)}


def kmer_index(kmer):
    """2-bit index of an ACGT 6-mer (first base most significant); None if not indexable."""
    if len(kmer) != 6:
        return None
    idx = 0
    for ch in kmer:
        c = _CODE.get(ch)
        if c is None:
            return None
        idx = idx * 4 + c
    return idx


class bias_model(object):
    """reference: bias.py:9-58"""

    default = 1e-6  # bias.py:17

    def __init__(self):
        self.model = {}
        self.k = 6
        self.mid = 3

    def __getitem__(self, key):
        return self.model.get(key, self.default)

    def __setitem__(self, key, value):
        self.model[key] = value
        self._table_cache = None

    def invalidate(self):
        """Drop the cached device table (call after mutating `self.model` directly)."""
        self._table_cache = None

    def offset(self):
        return max(self.k - self.mid, self.mid)

    def shuffle(self):
        """A copy of the model with the propensities randomly re-assigned to the k-mers
        (bias.py:25-43; one `random.random()` key per value, like the reference)."""
        values = list(self.model.values())
        keys = [random.random() for _ in values]
        perm = sorted(range(len(values)), key=keys.__getitem__)
        out = bias_model()
        out.model = dict(zip(self.model.keys(), (values[i] for i in perm)))
        out.offset = self.offset
        return out

    def predict(self, probs, n=100):
        """Distribute n tags according to relative propensities (bias.py:45-56)."""
        probs = np.asarray(probs)
        return np.around(probs / probs.sum() * n)

    # ---- device table ------------------------------------------------------------------
    def table(self):
        """4096 propensities in 2-bit order; k-mers absent from the model get the default."""
        fp = (id(self.model), len(self.model), self.default)
        cached = getattr(self, "_table_cache", None)
        if cached is not None and cached[0] == fp:
            return cached[1]
        t = np.full(4096, self.default, dtype=np.float64)
        for kmer, val in self.model.items():
            idx = kmer_index(kmer)
            if idx is not None:
                t[idx] = val
        self._table_cache = (fp, t)
        return t

    def probs_both(self, seq, ctx=None):
        """(forward, reverse) propensities as prediction.compute uses them
        (predict.pyx:150-153): fwd[j] = model[seq[j:j+6]], rev[j] = model[revcomp(seq[j+1:j+7])]."""
        ctx = ctx or _lib.get_ctx()
        ctx.set_bias_table(self.table(), self.default)
        if isinstance(seq, str):
            seq = seq.encode("ascii", "replace")
        s = np.frombuffer(bytes(seq), dtype=np.uint8)
        n = max(s.size - 6, 0)
        fwd, rev = np.empty(n), np.empty(n)
        _lib.check(ctx.L.fpt_kmer_probs(ctx.h, _lib.ptr(s), s.size, _lib.ptr(fwd), _lib.ptr(rev)))
        return fwd, rev

    def probs(self, seq):
        """Cleavage preference array from a DNA sequence (bias.py:88-111): len(seq)-6 values."""
        return self.probs_both(seq)[0]


class kmer_model(bias_model):
    """k-mer model read from a two-column text file or URL (bias.py:58-86)."""

    def __init__(self, filepath):
        bias_model.__init__(self)
        self.read_model(filepath)

    def read_model(self, filepath):
        """`KMER<TAB>propensity` per line; k-mers are upper-cased."""
        try:
            if filepath.startswith("http"):
                import urllib.request
                fh = urllib.request.urlopen(filepath)
            else:
                fh = open(filepath, "r")
            with fh:
                for raw in fh:
                    text = raw.decode() if isinstance(raw, bytes) else raw
                    kmer, value = text.strip().split("\t")
                    self.model[kmer.upper()] = float(value)
        except IOError:
            raise IOError("Cannot open file: %s" % filepath)
        self._table_cache = None


class uniform_model(bias_model):
    """Every 6-mer has propensity 1 (bias.py:114-122).  Like the reference, `probs` returns
    len(seq) ones rather than len(seq)-6; fast_predict only reads the first l of them."""

    def __init__(self):
        bias_model.__init__(self)
        self.model = dict.fromkeys(("".join(k) for k in itertools.product("ATCG", repeat=self.k)), 1.0)

    def probs(self, seq):
        return np.ones(len(seq))

    def probs_both(self, seq, ctx=None):
        return np.ones(len(seq)), np.ones(len(seq))
