#!/usr/bin/env python3
"""Phase times of one pmbwt build on a synthetic text, for looking at where a workload spends its time.

    python tools/phase_times.py --kind repetitive --copies 100 --genome 46000 --leaves 8
    python tools/phase_times.py --kind random --n 100000000 --sigma 4 --leaves 4
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402


def repetitive(copies, genome, rate, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(1, 5, size=genome, dtype=np.uint8)
    parts = []
    for _ in range(copies):
        g = base.copy()
        k = int(genome * rate)
        if k:
            idx = rng.integers(0, genome, size=k)
            g[idx] = rng.integers(1, 5, size=k, dtype=np.uint8)
        parts.append(g)
    parts.append(np.zeros(1, dtype=np.uint8))
    return np.concatenate(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="repetitive")
    ap.add_argument("--copies", type=int, default=100)
    ap.add_argument("--genome", type=int, default=46000)
    ap.add_argument("--rate", type=float, default=1e-3)
    ap.add_argument("--n", type=int, default=10 ** 8)
    ap.add_argument("--sigma", type=int, default=4)
    ap.add_argument("--leaves", type=int, default=8)
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--seg", type=int, default=0)
    args = ap.parse_args()
    ctx = bwt_amd.Context(0)
    if args.kind == "repetitive":
        T = repetitive(args.copies, args.genome, args.rate, 7)
        ctx.set_text(T)
        n = len(T)
    else:
        n = args.n
        ctx.gen_text(n, 42, args.sigma)
    th = -(-n // args.leaves)
    ns = max(1000, n // 64)
    ctx.pmbwt(th, ns, args.seg).free()
    ctx.sync()
    ctx.set_timing(args.level)
    ctx.reset_timers()
    t0 = time.perf_counter()
    ctx.pmbwt(th, ns, args.seg).free()
    ctx.sync()
    dt = time.perf_counter() - t0
    print("n=%d leaves=%d: %.1f ms = %.1f Mbase/s" % (n, args.leaves, dt * 1e3, (n - 1) / dt / 1e6))
    for k, v in sorted(ctx.timers().items(), key=lambda kv: -kv[1][0]):
        print("  %-36s %10.3f ms  %6d calls" % (k, v[0], v[2]))


if __name__ == "__main__":
    main()
