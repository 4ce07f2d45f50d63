"""Run by test_gpu_multi_leaf.py in a process of its own (the switch that lets small texts take the lean bucket route
is read when a context is created): ONE leaf built by several ranks that cut it by key range (bwt_hip_dist_pmbwt with
a tree of one leaf; here through bwt_hip_pmbwt_multi, the ranks being threads that share the GPU), against the oracle
and against the single-GPU leaf."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle, gen_repetitive, gen_text  # noqa: E402


def same_as_oracle(T, sigma, world, ns, want_used):
    n = len(T)
    L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * world, T, n, ns)
    assert want_used is None or used == want_used, (used, want_used)
    ob = Oracle.pmbwt(T, Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=n, ssa_n_samples=ns))
    assert (L == ob.L).all(), (sigma, world)
    assert (s_rank, e_rank) == (ob.s_rank, ob.e_rank)
    orr, op = ob.samples()
    assert len(r) == len(orr) and (r == orr).all() and (p == op).all()


def main():
    assert os.environ.get("BWT_HIP_BUCKET_MIN_N") == "1"
    for world in (2, 4, 8):
        same_as_oracle(gen_text(300000, 42, 4), 4, world, 3000, world)
    same_as_oracle(gen_text(250001, 5, 5), 5, 4, 777, 4)        # keys as numbers in base 5
    same_as_oracle(gen_text(150000, 7, 255), 255, 8, 2000, 8)   # bytes
    same_as_oracle(gen_text(100003, 9, 26), 26, 2, 100003, 2)   # every position sampled
    T = gen_text(200000, 7, 4)
    T[-3000:-1] = 1                                             # the text ends in a long run of the smallest symbol:
    same_as_oracle(T, 4, 4, 500, 1)                             # too many equal keys for the route, one rank builds it
    T = gen_text(200000, 8, 4)
    T[-12:-1] = 1                                               # a short run: keys that run into the terminator
    same_as_oracle(T, 4, 4, 500, 4)
    T = gen_text(120000, 3, 4)
    T[:-1] += ord("a") - 1
    same_as_oracle(T, ord("d"), 8, 700, 8)
    # skewed first digits: shares of very different numbers of digits, some of them empty
    rng = np.random.default_rng(4)
    T = np.concatenate([np.where(rng.random(200000) < 0.7, 1, rng.integers(1, 5, 200000)).astype(np.uint8), np.zeros(1, dtype=np.uint8)])
    same_as_oracle(T, 4, 8, 900, None)
    # a text the route gives up on (long repeats): one rank builds it
    same_as_oracle(gen_repetitive(3000, 60, 0.001, seed=43, sigma=4), 4, 4, 1000, 1)
    # 2e7 symbols against the single-GPU leaf
    n = 20000000
    ctx = bwt_amd.Context(0)
    ctx.gen_text(n, 42, 4)
    T = ctx.get_text()
    one = ctx.pmbwt(n, 100000)
    L1, (r1, p1) = one.L, one.samples()
    L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * 8, T, n, 100000)
    assert used == 8 and (L == L1).all() and (s_rank, e_rank) == (one.s_rank, one.e_rank)
    assert (r == r1).all() and (p == p1).all()
    one.free()
    ctx.close()
    print("multi leaf ok")


if __name__ == "__main__":
    main()
