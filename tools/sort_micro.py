#!/usr/bin/env python3
"""Kernel times of the plain (key, value) radix sort on uniform random 64-bit keys, from the library's timers."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000000
rng = np.random.default_rng(1)
keys = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64) * np.uint64(2)
vals = np.arange(n, dtype=np.uint32)
ctx = bwt_amd.Context(0)
L = bwt_amd.lib()
u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
for rep in range(2):
    k, v = keys.copy(), vals.copy()
    ctx.set_timing(2)
    ctx.reset_timers()
    rc = L.bwt_hip_test_sort_u64(ctx._h, k.ctypes.data_as(u64p), v.ctypes.data_as(u32p), n, 24, 64)
    assert rc == 0
    t = ctx.timers()
    ctx.set_timing(0)
for name, (ms, work, calls) in sorted(t.items()):
    print("%-30s %8.3f ms per launch  (%d launches)  %.0f GB/s" % (name, ms / calls, calls, work / (ms * 1e-3) / 1e9))
assert (np.diff(k >> np.uint64(24)) >= 0).all()
