"""Expected cleavage counts.  Mirrors footprint_tools/modeling/predict.pyx (v1.3.7); the
window sums, trimmed-mean smoothing and rounding of predict.h / smoothing.h run on the GPU."""
import numpy as np

from .. import _lib


def predict(obs, probs, half_win_width, smoothing_half_win_width, smoothing_clip, ctx=None):
    """`cdef predict` of predict.pyx:23-45: returns (exp, win), each len(obs).

    obs / probs may also be 2-D (rows = independent strands / intervals of equal length)."""
    ctx = ctx or _lib.get_ctx()
    obs, probs = _lib.f64(obs), _lib.f64(probs)
    if obs.shape != probs.shape and probs.size < obs.size:
        raise ValueError("probs shorter than obs")
    shape = obs.shape
    l = shape[-1] if obs.ndim else 0
    rows = obs.size // l if l else 0
    if probs.shape != obs.shape:  # uniform_model returns len(seq) values; only l are read
        probs = _lib.f64(probs.reshape(rows, -1)[:, :l])
    exp, win = np.zeros(shape), np.zeros(shape)
    _lib.check(ctx.L.fpt_predict(ctx.h, _lib.ptr(obs), _lib.ptr(probs), rows, l, int(half_win_width),
                                 int(smoothing_half_win_width), float(smoothing_clip),
                                 _lib.ptr(exp), _lib.ptr(win)))
    return exp, win


_COMPLEMENT = bytearray(b"N" * 256)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMPLEMENT[_a] = _b
_COMPLEMENT = bytes(_COMPLEMENT)


def reverse_complement(seq):
    """Reverse complement; anything but ACGTN (either case) becomes N (predict.pyx:47-61)."""
    return seq.encode("ascii", "replace").translate(_COMPLEMENT)[::-1].decode("ascii")


class prediction(object):
    """predict.pyx:63-163.  read_func[interval] -> {'+','-'} arrays, fasta_func.fetch(chrom,s,e)
    -> str, interval needs chrom/start/end/widen (genome_tools.genomic_interval duck type)."""

    def __init__(self, read_func, fasta_func, bm, half_win_width=5, smoothing_half_win_width=0,
                 smoothing_clip=0.01):
        self.read_func = read_func
        self.fasta_func = fasta_func
        self.bm = bm
        self.half_win_width = half_win_width
        self.smoothing_half_win_width = smoothing_half_win_width
        self.smoothing_clip = smoothing_clip
        self.padding = self.half_win_width + smoothing_half_win_width

    def compute(self, x):
        pad_interval = x.widen(self.padding)
        pad_interval.start -= 1  # predict.pyx:132-133
        raw_counts = self.read_func[pad_interval]
        raw_seq = self.fasta_func.fetch(pad_interval.chrom,
                                        pad_interval.start - self.bm.offset(),
                                        pad_interval.end + self.bm.offset()).upper()
        fwd, rev = self.bm.probs_both(raw_seq)
        cp = np.ascontiguousarray(raw_counts['+'], dtype=np.float64)
        cm = np.ascontiguousarray(raw_counts['-'], dtype=np.float64)
        l = cp.shape[0]
        exp, win = predict(np.stack([cp, cm]), np.stack([fwd[:l], rev[:l]]), self.half_win_width,
                           self.smoothing_half_win_width, self.smoothing_clip)
        obs_counts, exp_counts, win_counts = {}, {}, {}
        for row, strand in enumerate(['+', '-']):
            w = raw_counts[strand].shape[0] - self.padding
            obs_counts[strand] = raw_counts[strand][self.padding:w]
            exp_counts[strand] = exp[row][self.padding:w]
            win_counts[strand] = win[row][self.padding:w]
        return obs_counts, exp_counts, win_counts
