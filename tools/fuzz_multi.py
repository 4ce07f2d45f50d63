"""Differential run of the C++ multi-GPU driver (bwt_hip_pmbwt_multi, ranks that share GPU 0) against the
single-context build of the same tree, on random lengths, alphabets, rank counts, leaf sizes, sample counts and chain
lengths.  Small and mid-size texts (values on the wire, every alphabet class) and texts whose upper merges have left
blocks of 2^21 suffixes and more (g values grouped by gap bucket).  Test infrastructure:
    python tools/fuzz_multi.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402


def make(rng):
    big = rng.random() < 0.25
    if big:
        n = int(rng.integers(2 ** 22, 3 * 2 ** 22))
    else:
        n = int(rng.choice([rng.integers(40, 600), rng.integers(600, 9000), rng.integers(9000, 400000)]))
    sigma = int(rng.choice([2, 3, 4, 4, 4, 4, 5, 16, 200]))
    base = int(rng.integers(1, 256 - sigma + 1)) if rng.random() < 0.4 else 1
    if rng.random() < 0.1:
        base = 256 - sigma  # the alphabet ends at 255: symbol + 3 does not fit a byte
    if rng.random() < 0.2 and not big:
        body = np.where(rng.random(n - 1) < 0.9, 0, rng.integers(0, sigma, size=n - 1))  # skewed: parts of very different sizes
    else:
        body = rng.integers(0, sigma, size=n - 1)
    T = np.concatenate([(body + base).astype(np.uint8), np.zeros(1, dtype=np.uint8)])
    world = int(rng.choice([2, 4, 8]))
    leaves = world * int(rng.choice([1, 1, 2, 4]))
    th = -(-n // leaves)
    if rng.random() < 0.2:
        th = n  # one leaf: the ranks share it by key range (with BWT_HIP_BUCKET_MIN_N=1 also at these sizes), or one rank builds it
    ns = int(rng.choice([2, 17, max(2, n // 100), max(2, n // 7), n]))
    seg = int(rng.choice([0, 7, 64, 1000, 4096]))
    return T, world, th, ns, seg


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = bwt_amd.Context(0)
    for k in range(cases):
        T, world, th, ns, seg = make(rng)
        n = len(T)
        tag = "case %d: n=%d symbols %d..%d world=%d th=%d ns=%d seg=%d" % (k, n, T[:-1].min(), T[:-1].max(), world, th, ns, seg)
        if os.environ.get("FUZZ_VERBOSE"):
            print(tag, flush=True)
        ctx.set_text(T)
        one = ctx.pmbwt(th, ns, seg)
        L1, info1, (r1, p1) = one.L, (one.s_rank, one.e_rank), one.samples()
        one.free()
        L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * world, T, th, ns, seg)
        assert used <= world and (s_rank, e_rank) == info1, tag
        assert (L == L1).all(), tag
        assert len(r) == len(r1) and (r == r1).all() and (p == p1).all(), tag
        if k % 20 == 19:
            print("%d cases ok" % (k + 1), flush=True)
    print("fuzz ok: %d cases" % cases)


if __name__ == "__main__":
    main()
