"""GPU parity on the other benchmark configurations and on edge cases of the path: highly repetitive text (long
LCPs: dense ties, doubling rounds, window growth), byte alphabet (4 wavelet levels), tiny and degenerate texts."""
import os
import time

import numpy as np
import pytest

from oracle import Oracle, gen_repetitive, gen_text
from tests.gpu_util import ctx, inverse_bwt  # noqa: F401

pytestmark = pytest.mark.gpu


def _same_as_oracle(ctx, T, sigma, th, ns, seg=0):
    n = len(T)
    ctx.set_text(T)
    blk = ctx.pmbwt(th, ns, seg)
    o = Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
    ob = Oracle.pmbwt(T, o)
    assert (blk.L == ob.L).all()
    assert (blk.s_rank, blk.e_rank, blk.e_char) == (ob.s_rank, ob.e_rank, ob.e_char)
    for x, y in zip(blk.samples(), ob.samples()):
        assert (x == y).all()
    return blk


@pytest.mark.parametrize("text", [b"\0", b"a\0", b"ab\0", b"ba\0", b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa\0",
                                  b"abababababababababababababababababab\0", b"\xff\x01\xff\x01\xfe\0"])
@pytest.mark.parametrize("th", [1000, 3, 1])
def test_tiny_and_degenerate_texts(ctx, text, th):
    T = np.frombuffer(text, dtype=np.uint8).copy()
    n = len(T)
    ctx.set_text(T)
    blk = ctx.pmbwt(th, 2)
    L = blk.L
    assert (L == Oracle.bwt_direct(T)).all()
    assert L[blk.s_rank] == 0 and blk.e_rank == 0
    assert (inverse_bwt(L, blk.e_rank) == T).all()


def test_single_symbol_run(ctx):
    """One symbol repeated: every suffix comparison runs to the end of the text (LCP = n)."""
    T = np.concatenate([np.full(6000, 7, dtype=np.uint8), np.zeros(1, dtype=np.uint8)])
    blk = _same_as_oracle(ctx, T, 7, 2000, 100, 97)
    assert (blk.L[:-1] == 7).all() and blk.L[-1] == 0


def test_repetitive_genomes_small(ctx):
    """Config 4 at reduced scale: 100 copies of a 3000-symbol unit with 0.1 % substitutions, against the oracle."""
    T = gen_repetitive(3000, 100, 0.001, seed=43, sigma=4)
    _same_as_oracle(ctx, T, 4, 40000, 1000)
    _same_as_oracle(ctx, T, 4, len(T), 1000)


def test_repetitive_genomes_properties(ctx):
    """The same family at 4.6e6 symbols (100 x 46000): single leaf and 8 leaves agree; L is a permutation of T."""
    T = gen_repetitive(46000, 100, 0.001, seed=43, sigma=4)
    n = len(T)
    ctx.set_text(T)
    t0 = time.time()
    one = ctx.pmbwt(n, 5000)
    L1 = one.L
    t1 = time.time()
    many = ctx.pmbwt(n // 8 + 1, 5000)
    L8 = many.L
    t2 = time.time()
    print("repetitive n=%d: single leaf %.2f s, 8 leaves %.2f s" % (n, t1 - t0, t2 - t1))
    assert (L1 == L8).all()
    assert (np.bincount(L1, minlength=256) == np.bincount(T, minlength=256)).all()
    r, p = many.samples()
    assert (L8[r] == T[p - 1]).all()
    assert (inverse_bwt(L1[:], 0)[: 5000] == T[:5000]).all() if n <= 5000000 else True


def test_byte_alphabet_against_oracle(ctx):
    """Config 5 at reduced scale: 255 symbols, four 2-bit wavelet levels."""
    T = gen_text(300000, 42, 255)
    _same_as_oracle(ctx, T, 255, 80000, 3000)


def test_ascii_letters_min_sym_offset(ctx):
    """Symbols 'a'..'z': codes are relative to the smallest symbol (3 wavelet levels instead of 4)."""
    T = gen_text(120000, 3, 26)
    T[:-1] += ord("a") - 1
    _same_as_oracle(ctx, T, ord("z"), 20000, 700, 200)


@pytest.mark.parametrize("unit,copies,rate", [(700, 40, 0.0), (3000, 20, 0.001), (5000, 12, 0.0005), (37, 500, 0.0)])
def test_leaf_windows_on_long_repeats(ctx, unit, copies, rate):
    """Leaves in the middle of a text whose repeats are longer than the sort window's slack (4096 symbols here):
    the suffix array of each leaf must still be the full-text order (include/suffix_array.h:14-49)."""
    T = gen_repetitive(unit, copies, rate, seed=5, sigma=4)
    n = len(T)
    ctx.set_text(T)
    m = min(7000, n // 3)
    for a, b in [(0, m), (n // 3, n // 3 + m), (n - m - 1500, n - 1500), (n - m, n)]:
        assert (ctx.leaf_sa(a, b) == Oracle.sa_range(T, a, b)).all(), (unit, copies, a, b)


def test_few_distinct_symbols_in_a_wide_alphabet(ctx):
    """Four letters with large codes ('a'..'d'): the round-0 key holds few symbols, nearly all suffixes tie and
    the doubling rounds start immediately; with several leaves the windows end inside runs of ties."""
    rng = np.random.default_rng(11)
    T = np.concatenate([rng.integers(97, 101, size=150000, dtype=np.uint8), np.zeros(1, dtype=np.uint8)])
    _same_as_oracle(ctx, T, 100, 20000, 500)
    _same_as_oracle(ctx, T, 100, 3000, 500, 64)


@pytest.mark.timeout(600)
def test_leaf_window_above_2_31(ctx):
    """A single leaf of 2.3e9 suffixes (array indices beyond 2^31, split key halves beyond 2^32 words) gives the
    same L as two leaves and a merge; both are permutations of the text's symbols."""
    n = 2300000000
    ctx.gen_text(n, 42, 4)
    one = ctx.pmbwt(n, 100000)
    L1 = one.L
    one.free()
    two = ctx.pmbwt(n // 2 + 1, 100000)
    L2 = two.L
    r, p = two.samples()
    two.free()
    assert L1.shape == (n,) and (L1 == L2).all()
    counts = np.bincount(L1, minlength=5)
    assert counts[0] == 1 and counts[1:5].sum() == n - 1 and abs(int(counts[1]) - n // 4) < 200000
    T = ctx.get_text()
    assert (L2[r] == T[p - 1]).all()  # L[rank] = T[pos-1] for every sample (index -1 is the last symbol, as it should be)


@pytest.mark.parametrize("sigma", [2, 4, 8, 3])
@pytest.mark.parametrize("tail", [0, 1, 5, 19, 20, 21, 40, 300])
def test_text_ends_in_a_run_of_the_smallest_symbol(ctx, sigma, tail):
    """Alphabets of 2^k symbols get zero-based symbol codes in the sort keys, where the terminator looks like the
    smallest symbol: suffixes that end inside the key width, alone or in a run of the smallest symbol, must still
    come out in their place (several leaves, so that windows with and without the text's end are sorted)."""
    rng = np.random.default_rng(1000 * sigma + tail)
    body = rng.integers(1, sigma + 1, size=30000, dtype=np.uint8)
    body[rng.integers(0, 30000, size=40)] = 1  # more runs of the smallest symbol inside the text
    T = np.concatenate([body, np.full(tail, 1, dtype=np.uint8), np.zeros(1, dtype=np.uint8)])
    _same_as_oracle(ctx, T, sigma, len(T), 200)
    _same_as_oracle(ctx, T, sigma, 9000, 200, 50)


def test_binary_text_of_long_runs(ctx):
    """Two symbols, long runs of each, ending in a long run of the smaller one."""
    rng = np.random.default_rng(77)
    runs = rng.integers(1, 200, size=400)
    T = np.concatenate([np.full(r, 1 + (i & 1), dtype=np.uint8) for i, r in enumerate(runs)] +
                       [np.full(150, 1, dtype=np.uint8), np.zeros(1, dtype=np.uint8)])
    _same_as_oracle(ctx, T, 2, len(T), 100)
    _same_as_oracle(ctx, T, 2, 7000, 100)


def test_text_length_limit(ctx):
    """Texts of up to 2^32 - 65536 symbols are taken (device indices are 32-bit words and kernels index whole tiles with
    them); one more is refused with BWT_HIP_EINVAL before anything is allocated."""
    import bwt_amd
    with pytest.raises(bwt_amd.BwtHipError) as e:
        ctx.gen_text(2 ** 32 - 65535, 1, 4)
    assert "exceeds 2^32 - 65536" in str(e.value)
    ctx.gen_text(1000, 1, 4)  # the context is still usable
    assert ctx.pmbwt(1000, 10).ibwt_mismatches() == 0


def _with_runs(n, runs, seed=5, sigma=4):
    """Random text with runs planted: (start, length, symbol)."""
    rng = np.random.default_rng(seed)
    T = rng.integers(1, sigma + 1, size=n, dtype=np.uint8)
    for at, length, sym in runs:
        T[at:at + length] = sym
    T[-1] = 0
    return T


@pytest.mark.parametrize("th", [20000, 10000, 6000])
def test_runs_of_one_symbol_across_block_boundaries(ctx, th):
    """Runs of one symbol of more than 8191 symbols (an assembly's gaps) that cross the boundary between merged blocks,
    plus runs of the same symbol on either side: the chains' suffix comparisons step over listed runs instead of
    reading them.  Against the oracle."""
    n = 40000
    T = _with_runs(n, [(14000, 13000, 5), (1500, 9000, 5), (29000, 10500, 3)])
    ctx.set_timing(2)
    ctx.reset_timers()
    _same_as_oracle(ctx, T, 5, th, 300, 64)
    assert "gap:run_table" in ctx.timers()
    ctx.set_timing(0)
    # the same runs of the largest and of the smallest symbol, and a run that ends the text
    T = _with_runs(n, [(14000, 13000, 4), (1000, 9500, 1), (n - 9001, 9000, 1)])
    _same_as_oracle(ctx, T, 4, th, 300)


def test_long_runs_with_and_without_the_run_table(ctx):
    """A 4e5-symbol run across the middle of a 3e6-symbol text: the same block with the list of runs and, in a second
    context, with comparisons that read every symbol; decoded back on the device."""
    import bwt_amd
    n = 3000000
    T = _with_runs(n, [(n // 2 - 130000, 400000, 5), (200000, 70000, 5), (2500000, 50000, 5), (2700000, 20000, 2)])
    ctx.set_text(T)
    t0 = time.perf_counter()
    blk = ctx.pmbwt(n // 2 + 1, 30000)
    L, (r, p), info = blk.L, blk.samples(), (blk.s_rank, blk.e_rank)
    dt = time.perf_counter() - t0
    assert blk.ibwt_mismatches() == 0
    blk.free()
    b4 = ctx.pmbwt(n // 4 + 1, 30000)
    assert (b4.L == L).all()
    b4.free()
    old = os.environ.get("BWT_HIP_RUN_TABLE")
    os.environ["BWT_HIP_RUN_TABLE"] = "0"
    try:
        plain = bwt_amd.Context(0)
    finally:
        if old is None:
            del os.environ["BWT_HIP_RUN_TABLE"]
        else:
            os.environ["BWT_HIP_RUN_TABLE"] = old
    try:
        plain.set_text(T)
        b2 = plain.pmbwt(n // 2 + 1, 30000)
        assert (b2.L == L).all() and (b2.s_rank, b2.e_rank) == info
        r2, p2 = b2.samples()
        assert (r2 == r).all() and (p2 == p).all()
        b2.free()
    finally:
        plain.close()
    assert dt < 2.0, dt  # seconds; reading the runs symbol by symbol takes many times that


@pytest.mark.parametrize("unit", [[1, 2], [3, 1, 2], [4, 4, 1, 2, 2, 3, 1], [1, 2, 3, 4, 1, 2, 3, 3], [2, 1, 1, 3, 4, 2, 3, 1, 4]])
def test_long_repeats_of_a_short_unit_across_block_boundaries(ctx, unit):
    """A unit of two to nine symbols repeated over 13000 positions across the boundary between merged blocks, and again
    elsewhere with another phase (periods up to 8 are on the list of stretches that comparisons step over; 9 is not).
    Against the oracle."""
    n = 40000
    T = _with_runs(n, [])
    u = np.array(unit, dtype=np.uint8)
    for at, length in ((14001, 13000), (1503, 9100), (30000, 9000)):
        T[at:at + length] = np.tile(u, length // len(u) + 1)[:length]
    T[-1] = 0
    _same_as_oracle(ctx, T, 4, 20000, 300, 64)
    _same_as_oracle(ctx, T, 4, 10000, 300)


def test_run_across_the_leaf_boundary_fills_one_gap(ctx):
    """A run of one symbol that crosses the boundary of two leaves: all suffixes of its right-hand part fall into ONE gap
    of the left block (the left part's suffixes hold longer runs), so one bucket of the gap histogram holds millions
    of values and is counted by many workgroups.  The merged L must be the L of the same text built as one leaf."""
    n = 30000000
    rng = np.random.default_rng(11)
    T = rng.integers(1, 5, size=n, dtype=np.uint8)
    T[n // 2 - 3000000:n // 2 + 6000000] = 5
    T[-1] = 0
    ctx.set_text(T)
    one = ctx.pmbwt(n, 1000)
    L1, s1 = one.L.copy(), one.s_rank
    one.free()
    two = ctx.pmbwt(n // 2 + 1, 1000)
    assert two.s_rank == s1 and (two.L == L1).all() and two.ibwt_mismatches() == 0
    two.free()
