"""Run by test_gpu_run_block.py in a process of its own (the switches are read when the context is created): texts with
long runs of one symbol on the lean bucket route, where the suffixes that begin inside such a run are left out of the
sort and ranked in closed form, against the oracle.  Runs of 8191 symbols and more are sure to be on the list that
switches this on."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle, gen_text  # noqa: E402
from tests.bucket_route_worker import same_as_oracle  # noqa: E402


def launches(ctx, name):
    return ctx.timers().get(name, (0, 0, 0))[2]


def with_runs(n, seed, sigma, sym, runs):
    """random text of `sigma` symbols with runs (start, length) of `sym` planted"""
    T = gen_text(n, seed, sigma)
    for at, length in runs:
        T[at:at + length] = sym
    T[-1] = 0
    return T


def main():
    assert os.environ.get("BWT_HIP_BUCKET_MIN_N") == "1" and os.environ.get("BWT_HIP_FORCE_WIDE") == "2"
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)

    def check(T, sigma, th, ns, blocks):
        before = launches(ctx, "leaf_run_block")
        same_as_oracle(ctx, T, sigma, th, ns)
        got = (launches(ctx, "leaf_run_block") - before) // 2  # two stretches of the phase per leaf
        assert blocks is None or got == blocks, (got, blocks)

    n = 300000
    # a fifth symbol in runs (the N of an assembly), keys in base 5: one leaf, two leaves with a run across the middle,
    # four leaves (the longest run covers a whole leaf's end and the next one's start)
    runs = [(20000, 9000), (140000, 30000), (200000, 10000), (250000, 12345), (290000, 9999)]
    T = with_runs(n, 42, 4, 5, runs)
    check(T, 5, n, 3000, 1)
    check(T, 5, n // 2, 3000, 2)
    check(T, 5, n // 4, 777, None)
    # runs of the same length with the same symbols behind them for a while: decided by comparing the text
    T = with_runs(n, 43, 4, 5, [(30000, 10000), (90000, 10000), (150000, 10000), (210000, 10000)])
    for at in (40000, 100000, 160000, 220000):
        T[at:at + 40] = T[40000:40040]
    check(T, 5, n, 1000, 1)
    # DNA, runs of the smallest symbol (zero-based codes: the terminator looks like it), one of them ending the text
    T = with_runs(n, 44, 4, 1, [(0, 9000), (100000, 20000), (n - 15000, 14999)])
    check(T, 4, n, 2000, 1)
    check(T, 4, 100000, 2000, None)
    # DNA, runs of the largest symbol; of a middle one; followed by smaller and by larger symbols
    T = with_runs(n, 45, 4, 4, [(5000, 8500), (50000, 25000), (200000, 9000)])
    check(T, 4, n, 500, 1)
    T = with_runs(n, 46, 4, 2, [(5000, 8500), (50000, 25000), (120000, 8200), (200000, 9000), (260000, 16000)])
    T[13500], T[75000], T[128200], T[209000], T[276000] = 1, 3, 4, 1, 3
    check(T, 4, n, 30000, 1)
    check(T, 4, 75000, 30000, None)
    # the run symbol in the middle of the alphabet and rare outside its runs (no code of its own: the symbols above it
    # are coded one lower), as digits and as letters (A C G N T: the device holds them renumbered 1 .. 5, N = 4)
    T = with_runs(n, 52, 5, 4, [(7000, 9000), (60000, 22000), (150000, 8192), (280000, 15000)])
    body = T[:-1]
    keep4 = np.zeros(len(body), dtype=bool)
    for at, length in [(7000, 9000), (60000, 22000), (150000, 8192), (280000, 15000)]:
        keep4[at:at + length] = True
    scattered = (body == 4) & ~keep4
    body[scattered] = np.where(np.random.default_rng(9).random(int(scattered.sum())) < 0.002, 4, 5).astype(np.uint8)
    check(T, 5, n, 2500, 1)
    check(T, 5, n // 3, 2500, None)
    lut = np.zeros(256, dtype=np.uint8)
    lut[1:6] = [65, 67, 71, 78, 84]
    check(lut[T], 84, n, 2500, 1)
    check(lut[T], 84, 110000, 300, None)
    # no long runs at all, but a symbol that hardly occurs: every suffix that begins with it goes the same way
    T = gen_text(n, 53, 4)
    T[:-1][T[:-1] == 4] = 5
    where = np.random.default_rng(10).integers(0, n - 1, size=150)
    T[where] = 4
    T[where[:40] + 1] = 4  # some of them in pairs
    T[-1] = 0
    check(T, 5, n, 900, 1)
    check(lut[T], 84, n // 2, 900, 2)
    # letters (large byte values), bytes, three symbols
    T = with_runs(n, 47, 4, 3, [(100000, 40000)])
    T[:-1] += ord("a") - 1
    check(T, ord("d"), n, 900, 1)
    T = with_runs(200000, 48, 255, 77, [(10000, 9000), (100000, 50000)])
    check(T, 255, 200000, 900, 1)
    T = with_runs(n, 49, 3, 2, [(1000, 100000), (150000, 8300)])
    check(T, 3, n, 900, 1)
    # a text of one symbol, and one that is nearly so
    T = np.ones(100001, dtype=np.uint8)
    T[-1] = 0
    check(T, 1, 100001, 5000, None)
    T = np.full(120001, 3, dtype=np.uint8)
    T[::7001] = 2
    T[60000] = 4
    T[-1] = 0
    check(T, 4, 120001, 5000, None)
    # runs of two symbols: one is taken out, the other one's suffixes overflow a bucket and the route gives up
    T = with_runs(n, 50, 4, 5, [(20000, 30000)])
    T[100000:130000] = 2
    check(T, 5, n, 800, None)
    # the suffix array alone, of a range inside the text
    T = with_runs(n, 51, 4, 5, runs)
    ctx.set_text(T)
    for a, b in [(0, 150000), (145000, 260000), (150000, n)]:
        assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all(), (a, b)
    # switched off: the same results the long way
    ctx.close()
    os.environ["BWT_HIP_RUN_FILTER"] = "0"
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    same_as_oracle(ctx, T, 5, n // 2, 3000)
    assert launches(ctx, "leaf_run_block") == 0
    ctx.close()
    print("run block ok")


if __name__ == "__main__":
    main()
