"""GPU parity tests per phase of the hot path, through the C-ABI, against the CPU oracle and the golden
vectors of the real reference (tests/golden/golden.json).  Everything here is integer/byte work: bit-exact."""
import numpy as np
import pytest

from oracle import Oracle
from tests.golden_util import check_block, h64, load_golden, make_text
from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu
G = load_golden()


def _orc_opt(rec, T):
    return Oracle.opt(len(T), alpha_min=1, alpha_max=rec["alpha_max"], ssa_n_samples=rec["ssa_n_samples"],
                      build_gap_segment_sz=rec["seg"])


@pytest.mark.parametrize("rec", G["phase"], ids=lambda r: r["name"])
def test_phase_golden(ctx, rec):
    """Leaf SA, leaf blocks, C[], rank grid, gap array, merged block: each against the reference's record."""
    T = make_text(rec["text"])
    a, c, b = rec["a"], rec["c"], rec["b"]
    ns, seg = rec["ssa_n_samples"], rec["seg"]
    ctx.set_text(T)
    # sa_range (include/suffix_array.h:34-49)
    assert h64(ctx.leaf_sa(a, c)) == rec["sa_left_fnv"]
    assert h64(ctx.leaf_sa(c, b)) == rec["sa_right_fnv"]
    # bwt_leaf (include/bwt.h:615-635)
    lb = ctx.leaf(a, c, ns)
    rb = ctx.leaf(c, b, ns)
    Lfull = np.zeros(len(T), dtype=np.uint8)
    Lfull[a:c] = lb.L
    Lfull[c:b] = rb.L
    check_block(lb, Lfull, rec["left"], a, c)
    check_block(rb, Lfull, rec["right"], c, b)
    # count_alphabets + wavelet matrix (include/bwt.h:200-250)
    lb.init_extra()
    assert lb.C(rec["alpha_max"]).tolist() == rec["C_left"]
    for ch, row in zip(rec["rank_grid_c"], rec["rank_grid"]):
        got = lb.rank(np.full(len(row), ch, dtype=np.uint8), np.array(rec["rank_grid_i"]))
        assert got.tolist() == row
    # build_gap (include/bwt.h:730-787)
    gap, g = ctx.gap(lb, rb, seg)
    assert int(gap.sum()) == b - c
    assert int(gap.max()) == rec["gap_max"]
    assert h64(gap) == rec["gap_fnv"]
    assert (np.bincount(g - a, minlength=c - a + 1) == gap).all()
    # bwt_merge (include/bwt.h:791-847)
    mb = ctx.merge(lb, rb, ns, seg)
    Lm = np.zeros(len(T), dtype=np.uint8)
    Lm[a:b] = mb.L
    check_block(mb, Lm, rec["merged"], a, b)


@pytest.mark.parametrize("sigma,n,a,b", [(4, 50000, 0, 50000), (4, 50000, 777, 40001), (255, 30000, 5, 29000),
                                         (2, 20000, 100, 20000), (26, 40000, 0, 39999), (1, 300, 0, 300)])
def test_leaf_against_oracle(ctx, sigma, n, a, b):
    """src/test/test_leaf.cc re-expressed: random sub-range leaf; SA, L, special ranks and samples equal the oracle's."""
    T = make_text({"kind": "splitmix", "n": n, "seed": 100 + sigma, "sigma": sigma})
    ctx.set_text(T)
    o = Oracle.opt(n, alpha_min=1, alpha_max=max(sigma, 1), ssa_n_samples=257)
    Lo = np.zeros(n, dtype=np.uint8)
    ob, sa = Oracle.leaf(T, a, b, Lo, o, want_sa=True)
    assert (ctx.leaf_sa(a, b) == sa).all()
    gb = ctx.leaf(a, b, 257)
    assert (gb.L == Lo[a:b]).all()
    assert (gb.s_rank, gb.e_rank, gb.e_char) == (ob.s_rank, ob.e_rank, ob.e_char)
    for x, y in zip(gb.samples(), ob.samples()):
        assert (x == y).all()


def test_leaf_long_lcp_needs_doubling(ctx):
    """Repetitive text: most suffixes tie on the packed key and on the text round, so the inverse-array doubling runs."""
    T = make_text({"kind": "repetitive", "unit_len": 700, "copies": 30, "rate": 0.001, "seed": 9, "sigma": 4})
    ctx.set_text(T)
    n = len(T)
    for a, b in [(0, n), (1000, 15000), (n - 5000, n)]:
        assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all()


def test_leaf_window_overflow_retry(ctx):
    """A leaf whose suffix comparisons run far past its end: the sort has to widen its look-ahead window."""
    unit = make_text({"kind": "splitmix", "n": 40, "seed": 3, "sigma": 4})[:-1]
    T = np.concatenate([np.tile(unit, 2000), np.array([1, 2, 3, 0], dtype=np.uint8)])
    ctx.set_text(T)
    a, b = 10, 2000
    assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all()


@pytest.mark.parametrize("sigma", [4, 7, 16, 17, 200])
def test_rank_and_LF_against_oracle(ctx, sigma):
    """src/test/test_wavelet_matrix.cc re-expressed: rank(c,p) for random (c,p); LF_map_mod for every rank.
    The alphabets cover one 2-bit level, one and two 4-bit levels; the positions include rank-block and tile edges."""
    n = 30000
    T = make_text({"kind": "splitmix", "n": n, "seed": 55, "sigma": sigma})
    a, b = 1000, 26000
    ctx.set_text(T)
    o = Oracle.opt(n, alpha_min=1, alpha_max=sigma, ssa_n_samples=100)
    Lo = np.zeros(n, dtype=np.uint8)
    ob = Oracle.leaf(T, a, b, Lo, o)
    ob.init_extra()
    gb = ctx.leaf(a, b, 100)
    gb.init_extra()
    rng = np.random.RandomState(1)
    syms = rng.randint(1, sigma + 1, size=2000).astype(np.uint8)
    pos = rng.randint(0, b - a + 1, size=2000)
    pos[:3] = [0, b - a, 1]
    edges = [63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 3071, 3072, 3073, 6144, 24575, 24576]
    pos[3:3 + len(edges)] = edges
    exp = np.array([ob.rank(int(c), int(p)) for c, p in zip(syms, pos)])
    assert (gb.rank(syms, pos) == exp).all()
    ranks = np.arange(a, b, 7)
    Ls = Lo[ranks]
    keep = Ls != 0
    exp = np.array([ob.LF(int(c), int(r)) for c, r in zip(Ls[keep], ranks[keep])])
    assert (gb.LF(Ls[keep], ranks[keep]) == exp).all()


@pytest.mark.parametrize("seg", [1, 7, 64, 1024, 100000])
def test_gap_segment_sizes(ctx, seg):
    """The gap array does not depend on how the right block is cut into chains (include/opts.h:83-112)."""
    T = make_text({"kind": "splitmix", "n": 9000, "seed": 8, "sigma": 3})
    a, c, b = 100, 4000, 8999
    ctx.set_text(T)
    o = Oracle.opt(len(T), alpha_min=1, alpha_max=3, ssa_n_samples=64, build_gap_segment_sz=seg)
    Lo = np.zeros(len(T), dtype=np.uint8)
    m, gap, g = Oracle.merge(Oracle.leaf(T, a, c, Lo, o), Oracle.leaf(T, c, b, Lo, o), o, want_taps=True)
    lb, rb = ctx.leaf(a, c, 64), ctx.leaf(c, b, 64)
    ggap, gg = ctx.gap(lb, rb, seg)
    assert (ggap == gap).all() and (gg == g).all()
    mb = ctx.merge(lb, rb, 64, seg)
    assert (mb.L == Lo[a:b]).all()
    assert (mb.s_rank, mb.e_rank) == (m.s_rank, m.e_rank)
    for x, y in zip(mb.samples(), m.samples()):
        assert (x == y).all()


def test_merge_uploaded_blocks(ctx):
    """bwt_merge on blocks that come from the host (the reference API passes struct bwt by reference)."""
    T = make_text({"kind": "splitmix", "n": 20000, "seed": 77, "sigma": 4})
    a, c, b = 0, 9000, 20000
    ctx.set_text(T)
    o = Oracle.opt(len(T), alpha_min=1, alpha_max=4, ssa_n_samples=300)
    Lo = np.zeros(len(T), dtype=np.uint8)
    ol, orr = Oracle.leaf(T, a, c, Lo, o), Oracle.leaf(T, c, b, Lo, o)
    rl, pl = ol.samples()
    rr, pr = orr.samples()
    ul = ctx.upload_block(a, c, Lo[a:c], ol.s_rank, ol.e_rank, rl, pl)
    ur = ctx.upload_block(c, b, Lo[c:b], orr.s_rank, orr.e_rank, rr, pr)
    m = Oracle.merge(ol, orr, o)
    mb = ctx.merge(ul, ur, 300)
    assert (mb.L == Lo[a:b]).all()
    for x, y in zip(mb.samples(), m.samples()):
        assert (x == y).all()


def test_bad_text_is_rejected(ctx):
    import bwt_amd
    with pytest.raises(bwt_amd.BwtHipError):
        ctx.set_text(np.array([1, 2, 3], dtype=np.uint8))  # no terminator
    with pytest.raises(bwt_amd.BwtHipError):
        ctx.set_text(np.array([1, 0, 3, 0], dtype=np.uint8))  # two zeros
