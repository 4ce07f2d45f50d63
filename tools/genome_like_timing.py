"""A text with the features of an assembled genome that random DNA lacks -- skewed base composition, a family of
diverged repeats (300 symbols, 12 % substitutions, a fifth of the text), poly-A tails, microsatellites, runs of N --
built, decoded back on the device, and timed next to random DNA of the same length.
    python tools/genome_like_timing.py [n] [leaves] [letters]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402


def genome_like(n, seed=3, repeats=0.2, runs=12):
    rng = np.random.default_rng(seed)
    T = rng.choice(np.array([1, 2, 3, 4], dtype=np.uint8), size=n, p=[0.295, 0.205, 0.205, 0.295])
    unit = rng.integers(1, 5, size=300, dtype=np.uint8)
    copies = int(n * repeats / 330)
    starts = rng.integers(0, n - 400, size=copies)
    for s in starts:
        c = unit.copy()
        flip = rng.random(300) < 0.12
        c[flip] = rng.integers(1, 5, size=int(flip.sum()))
        T[s:s + 300] = c
        T[s + 300:s + 300 + int(rng.integers(8, 40))] = 1  # poly-A tail
    for _ in range(n // 15000):  # microsatellites
        u = rng.integers(1, 5, size=int(rng.integers(1, 7)), dtype=np.uint8)
        length = int(rng.integers(20, 200))
        s = int(rng.integers(0, n - 300))
        T[s:s + length] = np.tile(u, length // len(u) + 1)[:length]
    for _ in range(runs):
        length = int(10 ** rng.uniform(4, np.log10(max(20000, n // 60))))
        s = int(rng.integers(0, n - length - 1))
        T[s:s + length] = 5
    T[-1] = 0
    return T


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300000000
    leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    letters = len(sys.argv) > 3 and sys.argv[3] != "0"
    ctx = bwt_amd.Context(0)
    T = genome_like(n)
    if letters:
        lut = np.zeros(256, dtype=np.uint8)
        lut[1:6] = [65, 67, 71, 84, 78]
        T = lut[T]
    for name in ("genome-like", "random"):
        if name == "random":
            ctx.gen_text(n, 42, 4)
        else:
            ctx.set_text(T)
        ctx.set_timing(2)
        for k in range(2):
            ctx.reset_timers()
            t0 = time.perf_counter()
            blk = ctx.pmbwt(-(-n // leaves), max(1000, n // 100))
            ctx.sync()
            dt = time.perf_counter() - t0
            if k == 1:
                print("%s n=%d leaves=%d: %.1f ms, decoded back with %d mismatches" % (name, n, leaves, dt * 1e3, blk.ibwt_mismatches()), flush=True)
                for nm, (ms, _, calls) in sorted(ctx.timers().items(), key=lambda t: -t[1][0])[:18]:
                    if nm != "inverse_chains":
                        print("  %-34s %9.2f ms %5d" % (nm, ms, calls))
            blk.free()


if __name__ == "__main__":
    main()
