"""Single leaves and two-leaf trees at text lengths around 2^31 (where positions take all 32 bits of an element) and up to
the largest text the library takes (2^32 - 65536): L decoded back to the text on the device, and the same L from both
trees.  python tools/edge_sizes_2_32.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bwt_amd
from tests.golden_util import h8
ctx = bwt_amd.Context(0)
ctx.set_timing(2)
for n in (2147483648 + 5, 2147483648 - 3, 4000000000, 4294901760):
    ctx.reset_timers()
    ctx.gen_text(n, 3, 4)
    one = ctx.pmbwt(n, 1000000)
    v = sorted(k for k, x in ctx.timers().items() if k.startswith("leaf:bucket_emit_items=") and x[2] > 0)
    mism = one.ibwt_mismatches()
    f1 = h8(one.L); s1 = (one.s_rank, one.e_rank)
    one.free()
    ctx.trim()
    two = ctx.pmbwt(n // 2 + 1, 1000000)
    ok = h8(two.L) == f1
    two.free()
    ctx.trim()
    print(n, v, "ibwt mismatches", mism, "same as two leaves", ok, flush=True)
    assert mism == 0 and ok
print("edge sizes ok")
