"""Build time of random DNA over text lengths, as one leaf and as two leaves with a merge: python tools/size_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

ctx = bwt_amd.Context(0)
for n in (100000, 1000000, 4600000, 10000000, 30000000, 100000000, 200000000, 300000000, 1000000000):
    ctx.gen_text(n, 42, 4)
    out = []
    for leaves in (1, 2):
        best = 1e9
        for k in range(3):
            t0 = time.perf_counter()
            blk = ctx.pmbwt(-(-n // leaves), max(2, n // 80))
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
            blk.free()
        out.append("%d leaf: %8.3f ms = %7.1f Mbase/s" % (leaves, best * 1e3, n / best / 1e6))
    print("n=%-11d %s   %s" % (n, out[0], out[1]), flush=True)
