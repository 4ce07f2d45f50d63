"""The ranks of the text's difference-cover sample (what the reference's README.md:109 lists as missing there) and the
leaf pass that settles ties with them, against the oracle: the sample ranks are the oracle's suffix array restricted
to the sample positions; leaves and whole builds with the pass forced on equal the oracle's, on texts whose repeats are
much longer than any key."""
import os

import numpy as np
import pytest

from oracle import Oracle, gen_repetitive, gen_text
from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu

PERFECT = {31: [0, 1, 3, 8, 12, 18], 21: [0, 1, 4, 14, 16], 13: [0, 1, 3, 9], 7: [0, 1, 3]}


def _forced(mode="2"):
    import bwt_amd
    old = os.environ.get("BWT_HIP_DC")
    os.environ["BWT_HIP_DC"] = mode
    try:
        return bwt_amd.Context(0)
    finally:
        if old is None:
            del os.environ["BWT_HIP_DC"]
        else:
            os.environ["BWT_HIP_DC"] = old


def _texts():
    rng = np.random.default_rng(3)
    yield "dna copies", gen_repetitive(3000, 30, 0.001, seed=43, sigma=4), 31
    yield "dna random", gen_text(50000, 42, 4), 31
    yield "five symbols", np.concatenate([rng.integers(1, 6, size=2000, dtype=np.uint8)] * 12 + [np.zeros(1, dtype=np.uint8)]), 21
    yield "twelve symbols, offset", np.concatenate([rng.integers(100, 112, size=900, dtype=np.uint8)] * 20 + [np.zeros(1, dtype=np.uint8)]), 13
    yield "bytes", np.concatenate([rng.integers(1, 256, size=1500, dtype=np.uint8)] * 15 + [np.zeros(1, dtype=np.uint8)]), 7
    yield "unary", np.concatenate([np.full(5000, 2, dtype=np.uint8), np.zeros(1, dtype=np.uint8)]), 31
    yield "ends in a run of the smallest symbol", np.concatenate([rng.integers(1, 5, size=4000, dtype=np.uint8), np.full(77, 1, dtype=np.uint8), np.zeros(1, dtype=np.uint8)]), 31
    yield "tiny", np.array([3, 1, 2, 0], dtype=np.uint8), None


@pytest.mark.parametrize("name,T,period", list(_texts()), ids=[t[0] for t in _texts()])
def test_sample_ranks_are_the_suffix_array_restricted_to_the_sample(ctx, name, T, period):
    n = len(T)
    ctx.set_text(T)
    v, pos, rank = ctx.dc_sample()
    if period is not None:
        assert v == period
    D = PERFECT[v]
    want_pos = np.array([p for p in range(n) if p % v in D], dtype=np.int64)
    assert (pos == want_pos).all()
    sa = Oracle.sa_range(T, 0, n)
    isa = np.empty(n, dtype=np.int64)
    isa[sa] = np.arange(n)
    order = np.argsort(isa[pos], kind="stable")
    want_rank = np.empty(len(pos), dtype=np.int64)
    want_rank[order] = np.arange(len(pos))
    assert (rank == want_rank).all(), name


def test_every_pair_of_residues_has_a_common_shift():
    """The covers the library uses are perfect difference sets: every non-zero residue is the difference of exactly
    one pair, so two positions always have a shift below the period that makes both sampled."""
    for v, D in PERFECT.items():
        diffs = sorted((a - b) % v for a in D for b in D if a != b)
        assert diffs == list(range(1, v)), v


@pytest.mark.parametrize("name,T,period", list(_texts()), ids=[t[0] for t in _texts()])
def test_leaves_and_builds_with_the_pass_forced(name, T, period):
    n = len(T)
    c = _forced()
    try:
        c.set_text(T)
        c.set_timing(2)
        c.reset_timers()
        for a, b in [(0, n), (n // 3, n // 3 + min(5000, n - n // 3)), (0, max(1, n // 2))]:
            if b - a > 12000:
                b = a + 12000
            assert (c.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all(), (name, a, b)
        amax = int(T.max())
        for th, ns, seg in [(n, 50, 0), (max(1, n // 4 + 1), 200, 64)]:
            blk = c.pmbwt(th, min(ns, n), seg)
            ob = Oracle.pmbwt(T, Oracle.opt(n, alpha_min=1, alpha_max=amax, bwt_rec_threshold=th, ssa_n_samples=min(ns, n),
                                           build_gap_segment_sz=seg))
            assert (blk.L == ob.L).all(), (name, th)
            assert (blk.s_rank, blk.e_rank) == (ob.s_rank, ob.e_rank)
            for x, y in zip(blk.samples(), ob.samples()):
                assert (x == y).all()
            blk.free()
        if name in ("dna copies", "five symbols", "twelve symbols, offset", "bytes"):  # (one giant group, as in "unary", is left to the rounds)
            assert "leaf_dc_emit" in c.timers(), name  # the pass did run (the direct pass gives these texts up)
    finally:
        c.close()


def test_default_mode_takes_the_pass_on_dense_ties_only(ctx):
    """Random DNA never builds the sample; copies of a unit do, and the result does not depend on it."""
    ctx.set_timing(2)
    ctx.reset_timers()
    T = gen_text(300000, 42, 4)
    ctx.set_text(T)
    ctx.pmbwt(100000, 1000).free()
    assert "dc_sample" not in ctx.timers()
    T = gen_repetitive(20000, 20, 0.001, seed=5, sigma=4)
    ctx.set_text(T)
    ctx.reset_timers()
    with_pass = ctx.pmbwt(len(T) // 2 + 1, 1000)
    assert "dc_sample" in ctx.timers() and "leaf_dc_emit" in ctx.timers()
    off = _forced("0")
    try:
        off.set_text(T)
        off.set_timing(2)
        without = off.pmbwt(len(T) // 2 + 1, 1000)
        assert "dc_sample" not in off.timers()
        assert (with_pass.L == without.L).all() and (with_pass.s_rank, with_pass.e_rank) == (without.s_rank, without.e_rank)
        for x, y in zip(with_pass.samples(), without.samples()):
            assert (x == y).all()
        assert with_pass.ibwt_mismatches() == 0
    finally:
        off.close()
    ctx.set_timing(0)
