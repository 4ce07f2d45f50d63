"""Run by test_gpu_bucket_route.py in a process of its own (the switches are read when the context is created):
the leaf sort's lean bucket route -- 8-byte elements that do not store the first pass's digit, the second pass over
tiles that follow the first pass's regions, ties settled on the elements' deeper key bits before the text --
forced onto small texts, against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle, gen_repetitive, gen_text  # noqa: E402
from tests.bucket_route_worker import same_as_oracle  # noqa: E402


def lean_launches(ctx):
    return sum(v[2] for k, v in ctx.timers().items() if k.startswith("leaf:bucket_emit_items=") and k.endswith(",lean"))


def main():
    assert os.environ.get("BWT_HIP_BUCKET_MIN_N") == "1" and os.environ.get("BWT_HIP_FORCE_WIDE") == "2"
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    # random DNA: one leaf, leaves with window slack, many leaves
    T = gen_text(300000, 42, 4)
    for th in (len(T), 150000, 40000):
        same_as_oracle(ctx, T, 4, th, 3000)
    assert lean_launches(ctx) >= 1 + 2 + 8, sorted(ctx.timers())
    # the second pass: offsets from a histogram of its digits, or slots reserved as it goes (BWT_HIP_LEAN_SLOTS)
    t = ctx.timers()
    hists = sum(v[2] for k, v in t.items() if k in ("leaf:k_rs_hist_u64", "leaf:k_rs_hist_bytes"))
    passes = t["leaf:k_rs_scatter_u64_keys"][2]
    firsts = t["leaf:k_rs_scatter_text"][2]
    counted = t.get("leaf:k_rs_hist_text", (0, 0, 0))[2]  # first passes that counted their digits beforehand
    mode = os.environ.get("BWT_HIP_LEAN_SLOTS", "1")
    if mode == "3":
        assert hists == 0 and passes >= 11 and counted == 0 and firsts == passes, (hists, passes, counted, firsts)
    elif mode == "2":  # every second pass void, then again with offsets (first passes: counted, the texts are too small for a sample)
        assert hists >= 11 and passes == 2 * hists and counted == hists, (hists, passes, counted)
    elif mode == "5":  # slotted first passes, every second pass void and run again with offsets over the first pass's areas
        assert hists >= 11 and passes == 2 * hists and counted == 0, (hists, passes, counted)
    elif mode == "4":  # every first pass void, then again after counting
        assert hists == 0 and counted == passes >= 11 and firsts == 2 * passes, (hists, passes, counted, firsts)
    else:  # small texts: the slots would cost far more memory than the elements
        assert hists == passes == counted >= 11, (hists, passes, counted)
    # the text ends in a run of the smallest symbol: equal keys that ran into the terminator
    T = gen_text(200000, 7, 4)
    T[-3000:-1] = 1
    same_as_oracle(ctx, T, 4, 70000, 500)
    # a text of few distinct 8-symbol words: most first-pass regions are empty, some tiles short
    rng = np.random.default_rng(12)
    words = rng.integers(1, 5, size=(5, 9), dtype=np.uint8)
    T = np.concatenate([words[rng.integers(0, 5, 30000)].reshape(-1), np.zeros(1, dtype=np.uint8)])
    same_as_oracle(ctx, T, 4, 100000, 800)
    # letters with large byte values (7 bits of preceding symbol, 2-bit codes)
    T = gen_text(120000, 3, 4)
    T[:-1] += ord("a") - 1
    before = lean_launches(ctx)
    same_as_oracle(ctx, T, ord("d"), 50000, 700, 200)
    assert lean_launches(ctx) == before + 4
    # alphabets that are not a power of two (one-based codes), 2 to 8 bits per symbol
    # (3, 5, 6 and 7 symbols: the keys are numbers in that base, ties compared on the text from the first symbol)
    for sigma in (2, 3, 5, 6, 7, 8, 15, 16, 26, 100, 255):
        T = gen_text(150000, 20 + sigma, sigma)
        before = lean_launches(ctx)
        same_as_oracle(ctx, T, sigma, 60000, 800)
        assert lean_launches(ctx) == before + 4, sigma
    # five symbols, the text ending in a run of the smallest one (keys that run into the terminator), and one symbol rare
    T = gen_text(200000, 9, 5)
    T[-2000:-1] = 1
    T[:-1][T[:-1] == 5] = np.where(rng.random(int((T[:-1] == 5).sum())) < 0.02, 5, 2).astype(np.uint8)
    same_as_oracle(ctx, T, 5, 70000, 500)
    # leaf suffix arrays in the middle of the text
    T = gen_text(250000, 11, 4)
    ctx.set_text(T)
    for a, b in [(0, 90000), (100000, 190000), (160000, 250000)]:
        assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all(), (a, b)
    # a skewed text and a repetitive one: the route gives up, the other layouts finish
    T = np.concatenate([np.where(rng.random(200000) < 0.97, 1, rng.integers(1, 5, 200000)).astype(np.uint8),
                        np.zeros(1, dtype=np.uint8)])
    same_as_oracle(ctx, T, 4, 80000, 900)
    T = gen_repetitive(3000, 60, 0.001, seed=43, sigma=4)
    same_as_oracle(ctx, T, 4, 70000, 1000)
    # 19 bucket bits (the 10-bit second digit of windows of 2^31 suffixes and more) on a small text, against the oracle
    ctx.close()
    os.environ["BWT_HIP_WIDE_BBITS"] = "19"
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    T = gen_text(400000, 31, 4)
    for th in (len(T), 130000):
        same_as_oracle(ctx, T, 4, th, 2500)
    assert lean_launches(ctx) == 1 + 4, sorted(ctx.timers())
    same_as_oracle(ctx, gen_text(300000, 32, 7), 7, 300000, 900)
    # 2e7 suffixes with 16 to 19 bucket bits (8-, 9- and 10-bit digits) against the packed route of a plain context
    n = 20000000
    res = []
    for bb in ("16", "17", "18", "19"):
        ctx.close()
        os.environ["BWT_HIP_WIDE_BBITS"] = bb
        ctx = bwt_amd.Context(0)
        ctx.set_timing(2)
        ctx.gen_text(n, 9, 4)
        blk = ctx.pmbwt(n // 2 + 1, 50000)
        assert lean_launches(ctx) == 2
        assert blk.ibwt_mismatches() == 0
        res.append((blk.L, blk.samples(), blk.s_rank))
    ctx.close()
    for k in ("BWT_HIP_WIDE_BBITS", "BWT_HIP_FORCE_WIDE", "BWT_HIP_BUCKET_MIN_N"):
        os.environ.pop(k)
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    ctx.gen_text(n, 9, 4)
    blk = ctx.pmbwt(n // 2 + 1, 50000)
    assert lean_launches(ctx) == 0
    L, (r, p) = blk.L, blk.samples()
    for L2, (r2, p2), s2 in res:
        assert (L2 == L).all() and (r2 == r).all() and (p2 == p).all() and s2 == blk.s_rank
    ctx.close()
    print("lean route ok")


if __name__ == "__main__":
    main()
