"""Differential run of texts with long runs of one symbol and long repeats of a short unit: a context that steps over listed runs in the chains' suffix
comparisons against one that reads every symbol (BWT_HIP_RUN_TABLE=0), on random lengths, run lengths around the
list's block size, runs next to each other, at the text's ends and across leaf boundaries.  Test infrastructure:
    python tools/fuzz_runs.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402


def make(rng):
    n = int(rng.choice([rng.integers(9000, 40000), rng.integers(40000, 300000)]))
    sigma = int(rng.choice([2, 3, 4, 4, 5, 20]))
    T = rng.integers(1, sigma + 1, size=n, dtype=np.uint8)
    leaves = int(rng.choice([2, 2, 4, 8]))
    th = -(-n // leaves)
    for _ in range(int(rng.integers(1, 6))):
        length = int(rng.choice([rng.integers(4000, 4200), rng.integers(8100, 8300), rng.integers(4096, 20000), rng.integers(20000, 90000)]))
        length = min(length, n - 2)
        where = rng.random()
        if where < 0.3:  # across a leaf boundary
            at = int(rng.integers(1, leaves)) * th - int(rng.integers(0, length))
        elif where < 0.4:
            at = 0
        elif where < 0.5:
            at = n - 1 - length
        else:
            at = int(rng.integers(0, n - length))
        at = max(0, min(at, n - 1 - length))
        if rng.random() < 0.3:
            at &= ~4095  # block-aligned starts
        if rng.random() < 0.5:
            T[at:at + length] = int(rng.integers(1, sigma + 1))
        else:  # a unit of 2 to 9 symbols repeated (the list knows periods up to 8), sometimes with one symbol changed
            unit = rng.integers(1, sigma + 1, size=int(rng.integers(2, 10)), dtype=np.uint8)
            T[at:at + length] = np.tile(unit, length // len(unit) + 1)[:length]
            if rng.random() < 0.3:
                T[at + int(rng.integers(0, length))] = int(rng.integers(1, sigma + 1))
    T[-1] = 0
    ns = int(rng.choice([2, 50, max(2, n // 50)]))
    seg = int(rng.choice([0, 0, 64, 1000]))
    return T, th, ns, seg


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    fast = bwt_amd.Context(0)
    os.environ["BWT_HIP_RUN_TABLE"] = "0"
    plain = bwt_amd.Context(0)
    del os.environ["BWT_HIP_RUN_TABLE"]
    for k in range(cases):
        T, th, ns, seg = make(rng)
        tag = "case %d: n=%d th=%d ns=%d seg=%d" % (k, len(T), th, ns, seg)
        out = []
        for c in (fast, plain):
            c.set_text(T)
            b = c.pmbwt(th, ns, seg)
            out.append((b.L, b.s_rank, b.e_rank, b.samples()))
            if c is fast:
                assert b.ibwt_mismatches() == 0, tag
            b.free()
        (L0, s0, e0, (r0, p0)), (L1, s1, e1, (r1, p1)) = out
        assert (L0 == L1).all() and (s0, e0) == (s1, e1), tag
        assert (r0 == r1).all() and (p0 == p1).all(), tag
        if k % 10 == 9:
            print("%d cases ok" % (k + 1), flush=True)
    print("fuzz ok: %d cases" % cases)


if __name__ == "__main__":
    main()
