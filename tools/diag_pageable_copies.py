import sys, time
import numpy as np
sys.path.insert(0, ".")
from footprint_tools_amd import _lib
from footprint_tools_amd.scan import DeviceArray
ctx = _lib.Context(0)
n = 200_000_000  # 1.6 GB
d = DeviceArray(ctx, n * 8)
_lib.check(ctx.L.fpt_dev_zero(ctx.h, d.ptr, n * 8)); ctx.synchronize()
for trial in range(3):
    out = np.empty(n)  # fresh pageable
    t0 = time.perf_counter(); _lib.check(ctx.L.fpt_memcpy_d2h(ctx.h, out.ctypes.data, d.ptr, n * 8)); dt = time.perf_counter() - t0
    print("D2H into a FRESH numpy array: %.1f GB/s (%.1f ms)" % (n * 8 / dt / 1e9, dt * 1e3))
    t0 = time.perf_counter(); _lib.check(ctx.L.fpt_memcpy_d2h(ctx.h, out.ctypes.data, d.ptr, n * 8)); dt = time.perf_counter() - t0
    print("D2H into the same array again: %.1f GB/s (%.1f ms)" % (n * 8 / dt / 1e9, dt * 1e3))
src = np.ones(n)
for trial in range(2):
    t0 = time.perf_counter(); _lib.check(ctx.L.fpt_memcpy_h2d(ctx.h, d.ptr, src.ctypes.data, n * 8)); dt = time.perf_counter() - t0
    print("H2D from a numpy array: %.1f GB/s (%.1f ms)" % (n * 8 / dt / 1e9, dt * 1e3))
