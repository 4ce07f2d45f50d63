"""Doubling rounds that step over long runs of one symbol (suffix_sort.hip, k_comp_isa): texts whose runs are long
enough for it (still tied at a depth of 16384 symbols and more: 30000 symbols and longer), against the oracle."""
import numpy as np
import pytest

from oracle import Oracle
from tests.bucket_route_worker import same_as_oracle
from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


def runs_text(seed):
    """Random DNA with runs of symbol 2 followed by a smaller and by larger symbols (two of one length), a run of the
    smallest symbol, and a run of the largest at the very end."""
    rng = np.random.default_rng(seed)
    n = 260000
    T = rng.integers(1, 5, size=n, dtype=np.uint8)
    at = 3000
    for sym, length, follower in ((2, 40000, 1), (2, 40000, 3), (2, 33000, 4), (1, 31000, 3), (2, 30500, 1)):
        T[at - 1] = 4 if sym != 4 else 1  # (the run begins here)
        T[at:at + length] = sym
        T[at + length] = follower
        at += length + int(rng.integers(500, 9000))
    T[n - 1 - 36000:n - 1] = 4
    T[n - 2 - 36000] = 1
    T[-1] = 0
    return T


@pytest.mark.parametrize("leaves", [1, 2, 3])
def test_long_runs_against_the_oracle(ctx, leaves):
    T = runs_text(5 + leaves)
    n = len(T)
    ctx.set_timing(2)
    ctx.reset_timers()
    same_as_oracle(ctx, T, 4, -(-n // leaves), 1000)
    assert "refine:k_comp_isa_run_lengths" in ctx.timers(), sorted(ctx.timers())
    ctx.set_timing(0)


def test_leaf_suffix_array_with_long_runs(ctx):
    T = runs_text(2)
    ctx.set_text(T)
    for a, b in ((0, len(T)), (20000, 130000), (100000, len(T))):
        assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all(), (a, b)
