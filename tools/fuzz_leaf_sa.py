"""Differential run of leaf suffix arrays (positions a..b-1 in full-text order) against the C oracle: random texts
with and without long repeats, random ranges.  Test infrastructure: python tools/fuzz_leaf_sa.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = bwt_amd.Context(0)
    for k in range(cases):
        n = int(rng.choice([rng.integers(2, 200), rng.integers(200, 5000), rng.integers(5000, 40000)]))
        sigma = int(rng.choice([2, 3, 4, 4, 4, 7, 20, 255]))
        kind = rng.choice(["uniform", "copies", "runs"])
        if kind == "uniform":
            body = rng.integers(0, sigma, size=n - 1)
        elif kind == "copies":
            unit = rng.integers(0, sigma, size=int(rng.integers(5, 3000)))
            body = np.tile(unit, n // len(unit) + 1)[: n - 1].copy()
            flips = rng.random(n - 1) < rng.choice([0.0, 0.0005, 0.01])
            body[flips] = rng.integers(0, sigma, size=int(flips.sum()))
        else:
            body = np.repeat(rng.integers(0, sigma, size=n), rng.integers(1, 300, size=n))[: n - 1]
        T = np.concatenate([(body + 1).astype(np.uint8), np.zeros(1, dtype=np.uint8)])
        ctx.set_text(T)
        for _ in range(3):
            a = int(rng.integers(0, n))
            b = int(rng.integers(a + 1, n + 1))
            if rng.random() < 0.3:
                b = n
            if b - a > 12000:  # the oracle's comparison sort is slow on long repeats
                b = a + 12000
            tag = "case %d: n=%d sigma=%d %s [%d,%d)" % (k, n, sigma, kind, a, b)
            if os.environ.get("FUZZ_VERBOSE"):
                print(tag, flush=True)
            assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all(), tag
        if k % 50 == 49:
            print("%d cases ok" % (k + 1), flush=True)
    print("fuzz ok: %d cases" % cases)


if __name__ == "__main__":
    main()
