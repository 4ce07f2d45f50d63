"""How long texts with very long runs of one symbol take (genome assemblies hold runs of N of 1e7 symbols and more): a
random DNA text with one run planted in it, cut into leaves so that the run crosses a leaf boundary, built, decoded
back on the device, and timed.  Test infrastructure: python tools/runs_timing.py [n] [run] [leaves] [further runs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
    run = int(float(sys.argv[2])) if len(sys.argv) > 2 else 30000000
    leaves = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rng = np.random.default_rng(5)
    sigma = int(os.environ.get("RUNS_SIGMA", "4"))  # symbols of the random part
    sym = int(os.environ.get("RUNS_SYM", str(sigma + 1)))  # the run symbol (default: one more, behind the alphabet)
    T = rng.integers(1, sigma + 1, size=n, dtype=np.uint8)
    at = n // 2 - run // 3  # the run crosses the middle of the text
    T[at:at + run] = sym
    more = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # further runs of the same symbol, 1e4 symbols up to `run`, anywhere
    for _ in range(more):
        length = int(10 ** rng.uniform(4, np.log10(run)))
        where = int(rng.integers(0, n - length - 1))
        T[where:where + length] = sym
    T[-1] = 0
    if os.environ.get("RUNS_ASCII"):  # the same as letters, N between G and T, and some N scattered outside the runs
        lut = np.zeros(256, dtype=np.uint8)
        lut[1:6] = [65, 67, 71, 84, 78]
        T = lut[T]
        T[rng.integers(0, n - 1, size=int(os.environ["RUNS_ASCII"]))] = 78
        T[-1] = 0
    ctx = bwt_amd.Context(0)
    ctx.set_text(T)
    ctx.set_timing(2)
    for k in range(2):
        ctx.reset_timers()
        t0 = time.perf_counter()
        blk = ctx.pmbwt(-(-n // leaves), max(1000, n // 100))
        ctx.sync()
        dt = time.perf_counter() - t0
        print("n=%d run=%d leaves=%d: %.1f ms" % (n, run, leaves, dt * 1e3), flush=True)
        if k == 1:
            bad = blk.ibwt_mismatches()
            print("decoded back: %d mismatches" % bad, flush=True)
            for name, (ms, _, calls) in sorted(ctx.timers().items(), key=lambda t: -t[1][0])[:8]:
                print("  %-34s %9.2f ms %5d" % (name, ms, calls))
        blk.free()


if __name__ == "__main__":
    main()
