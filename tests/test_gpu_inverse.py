"""Device forms of bwt::ibwt (reference include/bwt.h:276-285) and bwt::sa (:456-481) through the C-ABI:
the round trip of src/test/test_leaf.cc, test_merge.cc, test_rec.cc and src/pmbwt.cc:64-81, and
src/test/test_ssa.cc's sa(r) == SA[r] for every r, against the text and the oracle's suffix array."""
import numpy as np
import pytest

from oracle import Oracle, gen_repetitive, gen_text
from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sigma,n,th,ns,seg", [(4, 200000, 200000, 1000, 0), (4, 200000, 30000, 777, 0),
                                               (255, 150000, 40000, 2000, 0), (26, 99991, 7000, 500, 333),
                                               (3, 5000, 100, 20, 16), (1, 300, 300, 2, 0), (17, 40000, 9000, 40000, 0)])
def test_round_trip_whole_text(ctx, sigma, n, th, ns, seg):
    T = gen_text(n, 500 + sigma, sigma)
    ctx.set_text(T)
    blk = ctx.pmbwt(th, ns, seg)
    assert blk.ibwt_mismatches() == 0
    assert (blk.ibwt() == T).all()
    SA = Oracle.sa_range(T, 0, n)
    ranks = np.unique(np.concatenate([np.arange(0, n, max(1, n // 997)), [0, n - 1, blk.s_rank, blk.e_rank]]))
    assert (blk.sa(ranks) == SA[ranks]).all()


def test_round_trip_sub_range_blocks(ctx):
    """Leaves and merged blocks of a proper sub-range [a,b): T[a-1] is foreign to the block and hidden."""
    T = gen_text(120000, 77, 4)
    ctx.set_text(T)
    for a, b, th in [(1234, 98765, 200000), (1234, 98765, 20000), (0, 60000, 7000), (60000, 120000, 20000)]:
        blk = ctx.rec(a, b, th, 900, 100)
        assert blk.ibwt_mismatches() == 0
        assert (blk.ibwt() == T[a:b]).all()
        SA = Oracle.sa_range(T, a, b)
        ranks = np.arange(a, b, 53)
        assert (blk.sa(ranks) == SA[ranks - a]).all()
        blk.free()


def test_round_trip_repetitive(ctx):
    T = gen_repetitive(3000, 40, 0.001, seed=43, sigma=4)
    ctx.set_text(T)
    blk = ctx.pmbwt(40000, 1000)
    assert blk.ibwt_mismatches() == 0
    assert (blk.ibwt() == T).all()


def test_all_ranks_of_a_small_block(ctx):
    """src/test/test_ssa.cc: sa(r) == SA[r] for every r."""
    T = gen_text(30000, 9, 4)
    ctx.set_text(T)
    a, b = 5000, 25000
    blk = ctx.rec(a, b, 6000, 150, 64)
    SA = Oracle.sa_range(T, a, b)
    assert (blk.sa(np.arange(a, b)) == SA).all()


def test_a_wrong_block_is_noticed(ctx):
    """A block whose L or samples do not describe the text decodes to something else."""
    T = gen_text(50000, 3, 4)
    ctx.set_text(T)
    good = ctx.pmbwt(50000, 500)
    L = good.L
    r, p = good.samples()
    args = (0, len(T), L, good.s_rank, good.e_rank, r, p)
    assert ctx.upload_block(*args).ibwt_mismatches() == 0
    L2 = L.copy()
    i = 12345 if L[12345] != 0 else 12346
    L2[i] = 1 + (L2[i] % 4)
    assert ctx.upload_block(0, len(T), L2, good.s_rank, good.e_rank, r, p).ibwt_mismatches() > 0
    r2 = r.copy()
    r2[100], r2[101] = r[101], r[100]
    assert ctx.upload_block(0, len(T), L, good.s_rank, good.e_rank, r2, p).ibwt_mismatches() > 0
