#!/usr/bin/env python3
"""Phase and kernel times of single leaf builds (a leaf that needs window slack, then one that ends at n)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000000
sigma = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = bwt_amd.Context(0)
ctx.gen_text(n, 42, sigma)
for a, b in [(0, n // 2), (n // 2, n)]:
    for rep in range(2):
        ctx.set_timing(2)
        ctx.reset_timers()
        try:
            ctx.leaf(a, b, 10000).free()
        except bwt_amd.BwtHipError as e:  # an instrumented build may leave the leaf incomplete: the times still count
            print("  (%s)" % e)
        t = ctx.timers()
        ctx.set_timing(0)
    print("leaf [%d, %d)" % (a, b))
    for name, (ms, work, calls) in sorted(t.items(), key=lambda kv: -kv[1][0]):
        print("  %-30s %8.3f ms  (%d launches)  %.0f GB/s" % (name, ms, calls, work / (ms * 1e-3) / 1e9 if ms else 0))
