"""All timers of one GPU's share of a leaf cut over G GPUs (see dist_leaf_times.py): python tools/dist_leaf_detail.py [n] [G] [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3100000000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else G // 2
ctx = bwt_amd.Context(0)
ctx.gen_text(n, 42, 4)
ns = max(1000, n // 80)
ctx.dist_leaf(0, n, ns, k, G)
ctx.sync()
for level in (0, 2):
    ctx.set_timing(level)
    ctx.reset_timers()
    t0 = time.perf_counter()
    r = ctx.dist_leaf(0, n, ns, k, G)
    ctx.sync()
    print("timing level %d: %.2f ms" % (level, (time.perf_counter() - t0) * 1e3))
tot = 0.0
for name, (ms, _, calls) in sorted(ctx.timers().items(), key=lambda t: -t[1][0]):
    print("  %-34s %9.3f ms %5d" % (name, ms, calls))
