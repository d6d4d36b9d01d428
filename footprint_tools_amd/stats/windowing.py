"""Sliding-window reducers.  Mirrors footprint_tools/stats/windowing.pyx (v1.3.7): out[i] =
f(x[i-hw..i+hw]) for i in [hw, n-hw) and 1.0 elsewhere, for every reducer."""
import numpy as np

from .. import _lib


def _window(op, x, hw, w=None, ctx=None):
    ctx = ctx or _lib.get_ctx()
    x = _lib.f64(x)
    if x.ndim == 0:
        raise ValueError("x must be an array")
    n = x.shape[-1]
    rows = x.size // n if n else 0
    out = np.ones(x.shape, dtype=np.float64)
    wp = None
    if w is not None:
        w = _lib.f64(w)
        if w.shape != x.shape:
            raise ValueError("weights and values differ in shape")
        wp = _lib.ptr(w)
    _lib.check(ctx.L.fpt_window(ctx.h, op, _lib.ptr(x), wp, rows, n, int(hw), _lib.ptr(out)))
    return out


def sum(x, hw):
    """windowing.pyx:60-76"""
    return _window(_lib.WIN_SUM, x, hw)


def product(x, hw):
    """windowing.pyx:78-94"""
    return _window(_lib.WIN_PRODUCT, x, hw)


def fishers_combined(x, hw):
    """windowing.pyx:96-112"""
    return _window(_lib.WIN_FISHER, x, hw)


def stouffers_z(x, hw):
    """windowing.pyx:114-130"""
    return _window(_lib.WIN_STOUFFER, x, hw)


def weighted_stouffers_z(x, w, hw):
    """windowing.pyx:160-178"""
    return _window(_lib.WIN_WSTOUFFER, x, hw, w)
