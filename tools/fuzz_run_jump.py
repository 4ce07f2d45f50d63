"""Texts with runs of one symbol long enough for the doubling rounds' step over them (30000 symbols and more), in
shapes meant to trip it -- runs of one length with different followers, runs next to runs of another symbol, a run
followed by a repeat that holds its symbol, runs at both ends of the text, runs across leaf boundaries -- against the
oracle.  python tools/fuzz_run_jump.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle  # noqa: E402
from tests.bucket_route_worker import same_as_oracle  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    jumped = 0
    for case in range(cases):
        sigma = int(rng.choice([2, 3, 4, 4, 5, 16, 200]))
        n = int(rng.integers(150000, 330000))
        T = rng.integers(1, sigma + 1, size=n, dtype=np.uint8)
        lengths = [int(rng.integers(25000, 60000)) for _ in range(int(rng.integers(1, 5)))]
        if len(lengths) > 1 and rng.random() < 0.6:
            lengths[1] = lengths[0]  # two runs of one length: their followers decide
        at = int(rng.choice([0, 1, int(rng.integers(2, 20000))]))
        for k, length in enumerate(lengths):
            if at + length + 2 >= n - 1:
                break
            sym = int(rng.integers(1, sigma + 1))
            T[at:at + length] = sym
            shape = rng.random()
            if shape < 0.25 and at + 2 * length + 2 < n - 1:  # a run of another symbol right behind it
                T[at + length:at + 2 * length] = sym % sigma + 1
                at += length
            elif shape < 0.4 and at + length + 6000 < n - 1:  # a repeat of two symbols, one of them the run's
                T[at + length:at + length + 6000] = np.tile(np.array([sym % sigma + 1, sym], dtype=np.uint8), 3000)
                at += 6000
            else:
                T[at + length] = int(rng.integers(1, sigma + 1))
            at += length + int(rng.choice([1, 2, int(rng.integers(3, 30000))]))
        if rng.random() < 0.4:  # a run up to the terminator
            length = int(rng.integers(25000, 50000))
            T[n - 1 - length:n - 1] = int(rng.integers(1, sigma + 1))
        T[-1] = 0
        leaves = int(rng.choice([1, 1, 2, 3, 4]))
        ctx.reset_timers()
        same_as_oracle(ctx, T, sigma, -(-n // leaves), int(rng.integers(2, 3000)))
        jumped += "refine:k_comp_isa_run_lengths" in ctx.timers()
        if rng.random() < 0.3:
            a = int(rng.integers(0, n // 2))
            b = int(rng.integers(a + 1, n + 1))
            assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all(), (case, a, b)
        print("case %d ok (n %d, sigma %d, %d leaves)" % (case, n, sigma, leaves), flush=True)
    assert jumped > 0
    print("fuzz_run_jump ok: %d cases, %d with the step over runs" % (cases, jumped))


if __name__ == "__main__":
    main()
