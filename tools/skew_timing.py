"""Random DNA whose symbols are not equally frequent (real genomes: 59 % A+T): how the leaf sort's routes take it.
    python tools/skew_timing.py [n] [leaves]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 600000000
leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = bwt_amd.Context(0)
rng = np.random.default_rng(1)
for p in ((0.25, 0.25, 0.25, 0.25), (0.26, 0.24, 0.24, 0.26), (0.295, 0.205, 0.205, 0.295)):
    T = rng.choice(np.array([1, 2, 3, 4], dtype=np.uint8), size=n, p=p)
    T[-1] = 0
    ctx.set_text(T)
    ctx.set_timing(2)
    for k in range(2):
        ctx.reset_timers()
        t0 = time.perf_counter()
        blk = ctx.pmbwt(-(-n // leaves), max(1000, n // 100))
        ctx.sync()
        dt = time.perf_counter() - t0
        if k == 1:
            print("p(A) = %.3f n=%d leaves=%d: %.1f ms, decoded back with %d mismatches" % (p[0], n, leaves, dt * 1e3, blk.ibwt_mismatches()), flush=True)
            for nm, (ms, _, calls) in sorted(ctx.timers().items(), key=lambda t: -t[1][0])[:9]:
                if nm != "inverse_chains":
                    print("  %-34s %9.2f ms %5d" % (nm, ms, calls))
        blk.free()
