"""Differential run against the C oracle on many small random configurations (text shape, alphabet, leaf size,
sample count, chain length).  Test infrastructure: python tools/fuzz_small.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle  # noqa: E402


def make(rng):
    n = int(rng.choice([rng.integers(2, 40), rng.integers(40, 600), rng.integers(600, 9000), rng.integers(9000, 60000)]))
    sigma = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8, 16, 17, 64, 255]))
    base = int(rng.integers(1, 256 - sigma + 1)) if rng.random() < 0.3 else 1
    kind = rng.choice(["uniform", "runs", "periodic", "skewed"])
    if kind != "uniform" or sigma < 3:
        n = min(n, 2500)  # the oracle's comparison sort is quadratic on long repeats
    if kind == "uniform":
        body = rng.integers(0, sigma, size=n - 1)
    elif kind == "runs":
        body = np.repeat(rng.integers(0, sigma, size=n), rng.integers(1, 40, size=n))[: n - 1]
    elif kind == "periodic":
        unit = rng.integers(0, sigma, size=int(rng.integers(1, 50)))
        body = np.tile(unit, n // len(unit) + 1)[: n - 1].copy()
        flips = rng.random(n - 1) < 0.002
        body[flips] = rng.integers(0, sigma, size=int(flips.sum()))
    else:
        body = np.where(rng.random(n - 1) < 0.9, 0, rng.integers(0, sigma, size=n - 1))
    T = np.concatenate([(body + base).astype(np.uint8), np.zeros(1, dtype=np.uint8)])
    top = base + sigma - 1
    if sigma <= 64 and rng.random() < 0.25:  # the symbols scattered over the byte range (the device holds such a text renumbered)
        values = np.sort(rng.choice(np.arange(1, 256), size=sigma, replace=False))
        lut = np.zeros(256, dtype=np.uint8)
        lut[base:base + sigma] = values
        T = lut[T]
        top = int(values[-1])
    th = int(rng.choice([n, max(1, n // 2), max(1, n // 3 + 1), max(1, n // 7), int(rng.integers(1, n + 1))]))
    ns = int(rng.choice([2, 3, 17, max(2, n // 10), n]))
    seg = int(rng.choice([0, 1, 5, 64, 1000]))
    if n > 3000:  # the oracle looks every chain head up through the sampled suffix array: keep that cheap
        seg = max(seg, 64) if seg else 0
        ns = max(ns, n // 50)
    return T, top, th, ns, seg, kind


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = bwt_amd.Context(0)
    for k in range(cases):
        T, amax, th, ns, seg, kind = make(rng)
        n = len(T)
        tag = "case %d: n=%d max_sym=%d th=%d ns=%d seg=%d %s" % (k, n, amax, th, ns, seg, kind)
        if os.environ.get("FUZZ_VERBOSE"):
            print(tag, flush=True)
        ctx.set_text(T)
        blk = ctx.pmbwt(th, ns, seg)
        o = Oracle.opt(n, alpha_min=1, alpha_max=amax, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
        ob = Oracle.pmbwt(T, o)
        assert (blk.L == ob.L).all(), tag
        assert (blk.s_rank, blk.e_rank, blk.e_char) == (ob.s_rank, ob.e_rank, ob.e_char), tag
        for x, y in zip(blk.samples(), ob.samples()):
            assert len(x) == len(y) and (x == y).all(), tag
        blk.free()
        if k % 50 == 49:
            print("%d cases ok" % (k + 1), flush=True)
    print("fuzz ok: %d cases" % cases)


if __name__ == "__main__":
    main()
