"""Two-leaf builds of 1e9 symbols (and a one-leaf build of 2.5e9) over alphabets of 2 to 255 symbols, letters with an
offset included: L decoded back to the text on the device, L a permutation of T, L[rank] = T[pos-1] at every sample.
Test infrastructure: python tools/soak_alphabets_full_size.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bwt_amd  # noqa: E402

ctx = bwt_amd.Context(0)
ctx.set_timing(2)
for n, sigma, leaves in ((1000000000, 2, 2), (1000000000, 3, 2), (1000000000, 5, 2), (1000000000, 16, 2), (1000000000, 255, 2),
                         (2500000000, 3, 1), (2500000000, 16, 1)):
    ctx.reset_timers()
    ctx.gen_text(n, 100 + sigma, sigma)
    blk = ctx.pmbwt(-(-n // leaves), n // 200)
    routes = sorted(k for k, v in ctx.timers().items() if k.startswith("leaf:bucket_emit_items=") and v[2] > 0)
    assert blk.ibwt_mismatches() == 0, (n, sigma)
    L, T = blk.L, ctx.get_text()
    r, p = blk.samples()
    assert (np.bincount(L, minlength=256) == np.bincount(T, minlength=256)).all(), (n, sigma)
    assert (L[r] == T[p - 1]).all() and p[0] == 0 and p[-1] == n - 1, (n, sigma)
    blk.free()
    del L, T
    ctx.trim()
    print("n=%d sigma=%d leaves=%d ok %s" % (n, sigma, leaves, routes), flush=True)
print("soak ok")
