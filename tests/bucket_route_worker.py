"""Run by test_gpu_bucket_route.py in a process of its own (the bucket route's threshold is read when the context
is created): leaf sorts that take the bucket route on small texts, against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle, gen_repetitive, gen_text  # noqa: E402


def same_as_oracle(ctx, T, sigma, th, ns, seg=0):
    n = len(T)
    ctx.set_text(T)
    blk = ctx.pmbwt(th, ns, seg)
    o = Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
    ob = Oracle.pmbwt(T, o)
    assert (blk.L == ob.L).all()
    assert (blk.s_rank, blk.e_rank, blk.e_char) == (ob.s_rank, ob.e_rank, ob.e_char)
    for x, y in zip(blk.samples(), ob.samples()):
        assert (x == y).all()


def launches(ctx, name):
    return ctx.timers().get(name, (0, 0, 0))[2]


def main():
    assert os.environ.get("BWT_HIP_BUCKET_MIN_N") == "1"
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    # random DNA: every bucket is small; one leaf, leaves with window slack, many leaves
    T = gen_text(300000, 42, 4)
    for th in (len(T), 150000, 40000):
        same_as_oracle(ctx, T, 4, th, 3000)
    assert launches(ctx, "leaf:k_bucket_emit") > 0, sorted(ctx.timers())
    # the text ends in a run of the smallest symbol: equal keys that ran into the terminator keep their order
    T = gen_text(200000, 7, 4)
    T[-3000:-1] = 1
    same_as_oracle(ctx, T, 4, 70000, 500)
    # letters with large byte values, 2 bits each
    T = gen_text(120000, 3, 4)
    T[:-1] += ord("a") - 1
    same_as_oracle(ctx, T, ord("d"), 50000, 700, 200)
    # leaf suffix arrays in the middle of the text
    T = gen_text(250000, 11, 4)
    ctx.set_text(T)
    for a, b in [(0, 90000), (100000, 190000), (160000, 250000)]:
        assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all(), (a, b)
    # a skewed text: one bucket holds far more than a workgroup takes, the passes take over
    rng = np.random.default_rng(5)
    T = np.concatenate([np.where(rng.random(200000) < 0.97, 1, rng.integers(1, 5, 200000)).astype(np.uint8),
                        np.zeros(1, dtype=np.uint8)])
    ctx.reset_timers()
    same_as_oracle(ctx, T, 4, 80000, 900)
    assert launches(ctx, "leaf:k_bucket_emit") == 0
    # repeats: the direct pass gives up after the bucket route has sorted the window
    T = gen_repetitive(3000, 60, 0.001, seed=43, sigma=4)
    ctx.reset_timers()
    same_as_oracle(ctx, T, 4, 70000, 1000)
    assert launches(ctx, "leaf:k_bucket_emit") > 0 and launches(ctx, "leaf:k_bucket_sort") > 0
    # a long skewed text: the probe for long repeats (runs of the frequent symbol) or, without it, a sample of the
    # window already says that the bucket route will not do
    n = 70000000
    T = np.concatenate([np.where(rng.random(n) < 0.9, 1, rng.integers(1, 5, n)).astype(np.uint8),
                        np.zeros(1, dtype=np.uint8)])
    ctx.set_text(T)
    ctx.reset_timers()
    blk = ctx.pmbwt(len(T), 4000)
    assert launches(ctx, "leaf:k_repeat_probe") + launches(ctx, "leaf:k_bucket_histogram") > 0
    assert launches(ctx, "leaf:k_bucket_bounds") == 0 and launches(ctx, "leaf:k_bucket_emit") == 0
    L = blk.L
    assert (np.bincount(L, minlength=5) == np.bincount(T, minlength=5)).all()
    r, p = blk.samples()
    assert (L[r] == T[p - 1]).all()
    del T, L
    # 4e6 suffixes: 16 symbols per key, two ranking rounds per bucket; against the four plain passes
    n = 4000000
    ctx.gen_text(n, 9, 4)
    ctx.reset_timers()
    a = ctx.pmbwt(n // 2 + 1, 5000)
    La, ra, pa = a.L, *a.samples()
    assert launches(ctx, "leaf:k_bucket_emit") > 0
    ctx.close()
    os.environ.pop("BWT_HIP_BUCKET_MIN_N")
    os.environ.pop("BWT_HIP_FORCE_WIDE", None)
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    ctx.gen_text(n, 9, 4)
    b = ctx.pmbwt(n // 2 + 1, 5000)
    assert launches(ctx, "leaf:k_bucket_emit") == 0 and launches(ctx, "leaf:k_rs_scatter_u64_keys") > 0
    assert (La == b.L).all() and (a.s_rank, a.e_rank) == (b.s_rank, b.e_rank)
    rb, pb = b.samples()
    assert (ra == rb).all() and (pa == pb).all()
    ctx.close()
    print("bucket route ok")


if __name__ == "__main__":
    main()
