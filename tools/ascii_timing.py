"""How long a text takes whose symbols are letters far apart in the byte range (A, C, G, T as ASCII: 65, 67, 71, 84) against
the same text with the symbols 1..4.  python tools/ascii_timing.py [n] [leaves]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200000000
leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = bwt_amd.Context(0)
ctx.gen_text(n, 42, 4)
T = ctx.get_text()
lut = np.zeros(256, dtype=np.uint8)
lut[1:5] = [65, 67, 71, 84]
A = lut[T]
for name, text in (("1..4", T), ("ACGT", A)):
    ctx.set_text(text)
    ctx.set_timing(2)
    for k in range(2):
        ctx.reset_timers()
        t0 = time.perf_counter()
        blk = ctx.pmbwt(-(-n // leaves), max(1000, n // 100))
        ctx.sync()
        dt = time.perf_counter() - t0
        if k == 1:
            print("%s n=%d leaves=%d: %.1f ms" % (name, n, leaves, dt * 1e3), flush=True)
            for nm, (ms, _, calls) in sorted(ctx.timers().items(), key=lambda t: -t[1][0])[:6]:
                print("  %-34s %9.2f ms %5d" % (nm, ms, calls))
            if name == "1..4":
                L0 = blk.L
            else:
                print("same L after mapping:", bool((lut[L0] == blk.L).all() or True) and int((lut[L0] != blk.L).sum()))
        blk.free()
