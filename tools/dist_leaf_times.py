#!/usr/bin/env python3
"""What ONE GPU does when G GPUs share one leaf of an n-symbol text by key range (bwt_hip_dist_leaf), timed on a single
GPU for every G and a few shares: its count over the whole text, the text pass that keeps its stretch of first digits,
the second pass and the bucket kernel on what it kept.  The exchange (two numbers and ns / G sample slots per pair of
GPUs) is not part of it.

    python tools/dist_leaf_times.py [n] [G ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3100000000
Gs = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
ctx = bwt_amd.Context(0)
ctx.gen_text(n, 42, 4)
ns = max(1000, n // 80)
for G in Gs:
    for k in sorted({0, G // 2, G - 1}):
        ctx.dist_leaf(0, n, ns, k, G)  # warm the pool
        ctx.sync()
        ctx.set_timing(2)
        ctx.reset_timers()
        t0 = time.perf_counter()
        r = ctx.dist_leaf(0, n, ns, k, G)
        ctx.sync()
        dt = time.perf_counter() - t0
        ctx.set_timing(0)
        top = sorted(((v[0], kk) for kk, v in ctx.timers().items() if kk.startswith("leaf:k_")), reverse=True)[:5]
        print("G=%d share %d: %.1f ms, %d suffixes from rank %d; %s" % (G, k, dt * 1e3, r["count"], r["first_rank"],
              ", ".join("%s %.1f" % (kk[5:], ms) for ms, kk in top)), flush=True)
    ctx.trim()
