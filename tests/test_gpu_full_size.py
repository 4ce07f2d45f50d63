"""GPU parity at the sizes the benchmark runs: the leaf sort's bucket route as it is taken unforced (windows of
1e8 suffixes and more, buckets of thousands of suffixes, every instantiation of k_bucket_emit), the packed and
the pairs layout, config 2 (n = 1e9, 2 leaves) and config 3 (n = 3.1e9, 8 leaves) of bwt::pmbwt
(reference include/bwt.h:867-883).

Pinned to the reference itself where the serial reference finishes in minutes (tests/golden/golden_large.json:
whole builds at 1e7 and 1e8, single leaves of 3.6e8 to 5.6e8 suffixes); beyond that by size-independent
properties -- the same L from the bucket route and from the plain radix passes, the same L from different
leaf sizes, L a permutation of T, L[rank] = T[pos-1] at every sample -- and by decoding L back to the text
on the device."""
import json
import os

import numpy as np
import pytest

from tests.golden_util import check_block, h8
from tests.gpu_util import ctx  # noqa: F401

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "golden_large.json")) as f:
    GL = json.load(f)


def launches(ctx, name):
    return ctx.timers().get(name, (0, 0, 0))[2]


def variants(ctx):
    return sorted(k for k, v in ctx.timers().items() if k.startswith("leaf:bucket_emit_items=") and v[2] > 0)


def same_variants(got, want):
    """The instantiations of k_bucket_emit that ran: route flags exactly, elements per thread (set by the largest
    bucket, i.e. by the text) within one of what is expected."""
    def split(name):
        items, _, flags = name[len("leaf:bucket_emit_items="):].partition(",")
        return int(items), flags
    g, w = sorted(map(split, got), key=lambda x: x[1]), sorted(map(split, want), key=lambda x: x[1])
    return len(g) == len(w) and all(a[1] == b[1] and abs(a[0] - b[0]) <= 1 for a, b in zip(g, w))


@pytest.fixture()
def timed(ctx):
    ctx.set_timing(2)
    ctx.reset_timers()
    yield ctx
    ctx.set_timing(0)
    ctx.trim()


@pytest.mark.parametrize("rec", GL["pipeline"], ids=lambda r: r["name"])
def test_golden_large_pipeline(timed, rec):
    """Whole builds against records of the reference: n = 1e7 (16 leaves; DNA and bytes), n = 1e8 with 1, 2 and
    16 leaves.  The single leaf of 1e8 suffixes takes the bucket route by itself."""
    ctx = timed
    t = rec["text"]
    ctx.gen_text(t["n"], t["seed"], t["sigma"])
    blk = ctx.pmbwt(rec["bwt_rec_threshold"], rec["ssa_n_samples"], rec["opts"].get("seg", 0))
    check_block(blk, blk.L, rec, 0, t["n"])
    if rec["name"].endswith("n1e8_1leaf"):
        assert launches(ctx, "leaf:k_bucket_emit") == 1 and variants(ctx) == ["leaf:bucket_emit_items=4,lean"]
    blk.free()


@pytest.mark.parametrize("rec", GL.get("leaf", []), ids=lambda r: r["name"])
def test_golden_large_leaf(timed, rec):
    """bwt_leaf on windows whose buckets hold 5500 to 8500 suffixes: the 6-, 8- and 9-element instantiations of
    k_bucket_emit, with and without window slack, and one window above 2^29 suffixes, against the reference's
    leaf (L, special ranks, samples)."""
    ctx = timed
    t = rec["text"]
    ctx.gen_text(t["n"], t["seed"], t["sigma"])
    blk = ctx.leaf(rec["a"], rec["b"], rec["ssa_n_samples"])
    check_block(blk, blk.L, rec, 0, rec["b"] - rec["a"])
    # 5.6e8 suffixes are more than the packed 8-byte elements hold positions for (2^29): that window takes the lean
    # route (8-byte elements without the first pass's digit, 2^17 buckets)
    want = {"leaf_s4_whole_5.3e8": ["leaf:bucket_emit_items=9,lean"], "leaf_s4_whole_5.6e8": ["leaf:bucket_emit_items=5,lean"],
            "leaf_s4_first_5e8_of_5.01e8": ["leaf:bucket_emit_items=8,lean"],
            "leaf_s4_mid_3.6e8": ["leaf:bucket_emit_items=6,lean"]}[rec["name"]]
    assert launches(ctx, "leaf:k_bucket_emit") == len(want) and same_variants(variants(ctx), want), variants(ctx)
    blk.free()


def _plain_route_context(n):
    """A second context on the same GPU whose leaf sorts never take the bucket route."""
    import bwt_amd
    old = os.environ.get("BWT_HIP_BUCKET_MIN_N")
    os.environ["BWT_HIP_BUCKET_MIN_N"] = str(4 * n)
    try:
        return bwt_amd.Context(0)
    finally:
        if old is None:
            del os.environ["BWT_HIP_BUCKET_MIN_N"]
        else:
            os.environ["BWT_HIP_BUCKET_MIN_N"] = old


def _build(c, n, leaves, ns):
    blk = c.pmbwt(-(-n // leaves), ns)
    L = blk.L
    r, p = blk.samples()
    info = (blk.s_rank, blk.e_rank, blk.e_char)
    return blk, L, r, p, info


def _suffix_less(T, x, y):
    """T[x:] < T[y:] by reading the text itself (x != y; the unique terminator ends it)."""
    n = len(T)
    step = 64
    while True:
        k = min(step, n - x, n - y)
        a, b = T[x:x + k], T[y:y + k]
        d = np.flatnonzero(a != b)
        if len(d):
            return bool(a[d[0]] < b[d[0]])
        x, y, step = x + k, y + k, step * 4


def _adjacent_ranks_increase(blk, T, L, count=2000, seed=1):
    """The property the reference's own suffix array test checks (src/test/test_suffix_array.cc:27-48): for random
    ranks r, the suffixes at SA[r] and SA[r+1] (positions looked up on the device) are compared by reading the host copy
    of the text itself, and L[r] is the symbol in front of SA[r].  The verdict on the order comes from the text alone,
    not from the rank structure that the chains and the device decoding share."""
    n = len(T)
    rng = np.random.RandomState(seed)
    r = np.unique(np.concatenate([rng.randint(0, n - 1, size=count), [0, n - 2, blk.s_rank - 1 if blk.s_rank else 0]]))
    r = r[(r >= 0) & (r < n - 1)].astype(np.int64)
    pos = blk.sa(np.concatenate([r, r + 1]))
    p0, p1 = pos[:len(r)], pos[len(r):]
    assert (L[r] == T[p0 - 1]).all() and (L[r + 1] == T[p1 - 1]).all()
    for x, y in zip(p0.tolist(), p1.tolist()):
        assert x != y and _suffix_less(T, x, y), (x, y)


def _full_size(ctx, n, leaves, ns, expect_variants, other_leaves, sigma=4):
    ctx.set_timing(2)
    ctx.reset_timers()
    ctx.gen_text(n, 42, sigma)
    blk, L, r, p, info = _build(ctx, n, leaves, ns)
    got = variants(ctx)
    ctx.set_timing(0)
    ctx.trim()
    assert launches(ctx, "leaf:k_bucket_emit") == leaves and same_variants(got, expect_variants), got
    # decoded on the device, L gives the text back (src/pmbwt.cc:64-81 check_result)
    assert blk.ibwt_mismatches() == 0
    T = ctx.get_text()
    _adjacent_ranks_increase(blk, T, L)
    blk.free()
    # L is a permutation of T with the terminator at s_rank; every sample (rank, pos) has L[rank] = T[pos-1]
    assert (np.bincount(L, minlength=256) == np.bincount(T, minlength=256)).all()
    assert L[info[0]] == 0 and info[1] == 0 and info[2] == 0
    assert p[0] == 0 and p[-1] == n - 1 and (np.diff(p) > 0).all()
    assert (L[r] == T[p - 1]).all()
    del T
    fnv = h8(L)
    # the same tree with the plain radix passes instead of the bucket route: same L, same samples
    ctx.trim()
    plain = _plain_route_context(n)
    try:
        plain.set_timing(2)
        plain.gen_text(n, 42, sigma)
        b2, L2, r2, p2, info2 = _build(plain, n, leaves, ns)
        assert launches(plain, "leaf:k_bucket_emit") == 0
        assert info2 == info and (r2 == r).all() and (p2 == p).all()
        assert (L2 == L).all()
        b2.free()
        del L2
        # another leaf size: L does not depend on it (samples do)
        b3 = plain.pmbwt(-(-n // other_leaves), ns)
        assert h8(b3.L) == fnv
        b3.free()
    finally:
        plain.close()
    return fnv


def test_config2_n1e9_two_leaves(ctx):
    """Config 2 as bench.py --n 1000000000 --leaves 2 runs it: both leaves on the bucket route (8 elements per
    thread; the left one with window slack)."""
    _full_size(ctx, 10 ** 9, 2, 12500000, ["leaf:bucket_emit_items=8,lean"], 4)


def test_config3_n3_1e9_eight_leaves(ctx):
    """Config 3 on one GPU as bench.py runs it by default: 8 leaves of 3.875e8 suffixes, 7 merges."""
    _full_size(ctx, 3100000000, 8, 19375000, ["leaf:bucket_emit_items=7,lean"], 4)
    ctx.trim()


def test_n3_1e9_two_leaves_lean_route(ctx):
    """bench.py's default on one GPU: 3.1e9 symbols as two leaves of 1.55e9 suffixes (more than 2^29: the lean
    bucket route, 2^18 buckets) and one merge; against the same tree sorted as (key, position) pairs."""
    _full_size(ctx, 3100000000, 2, 38750000, ["leaf:bucket_emit_items=7,lean"], 8)
    ctx.trim()


def test_config5_bytes_n1e9_two_leaves(ctx):
    """Config 5: a byte alphabet (255 symbols, two 16-ary wavelet levels) at n = 1e9, two leaves and one merge; both
    leaves on the lean bucket route, against the same tree sorted as (key, position) pairs."""
    _full_size(ctx, 10 ** 9, 2, 12500000, ["leaf:bucket_emit_items=8,lean"], 4, sigma=255)
    ctx.trim()


def test_five_symbols_n1e9_two_leaves(ctx):
    """DNA with N as far as the alphabet goes (five equally frequent symbols) at n = 1e9: both leaves on the lean route
    with keys that are numbers in base 5, against the same tree sorted as (key, position) pairs."""
    _full_size(ctx, 10 ** 9, 2, 12500000, ["leaf:bucket_emit_items=5,lean"], 4, sigma=5)
    ctx.trim()


def _forced_wide_context():
    """A second context on the same GPU whose large windows take the wide route (8 + 1 byte elements)."""
    import bwt_amd
    old = os.environ.get("BWT_HIP_FORCE_WIDE")
    os.environ["BWT_HIP_FORCE_WIDE"] = "1"
    try:
        return bwt_amd.Context(0)
    finally:
        if old is None:
            del os.environ["BWT_HIP_FORCE_WIDE"]
        else:
            os.environ["BWT_HIP_FORCE_WIDE"] = old


def test_n3_1e9_one_leaf(ctx):
    """3.1e9 suffixes as one leaf (more than 2^31: lean elements with 32-bit positions, 2^19 buckets through a 10-bit
    second digit): decoded back to the text on the device; the same L as the two-leaf build, and as the wide route
    (8 + 1 byte elements, three passes) that a second context is made to take."""
    n = 3100000000
    ctx.set_timing(2)
    ctx.reset_timers()
    ctx.gen_text(n, 42, 4)
    one = ctx.pmbwt(n, 38750000)
    got = variants(ctx)
    ctx.set_timing(0)
    assert same_variants(got, ["leaf:bucket_emit_items=7,lean"]), got
    assert one.ibwt_mismatches() == 0
    f1, info1 = h8(one.L), (one.s_rank, one.e_rank)
    r1, p1 = one.samples()
    one.free()
    two = ctx.pmbwt(n // 2, 38750000)
    assert h8(two.L) == f1
    two.free()
    ctx.trim()
    wide = _forced_wide_context()
    try:
        wide.set_timing(2)
        wide.gen_text(n, 42, 4)
        w = wide.pmbwt(n, 38750000)
        assert same_variants(variants(wide), ["leaf:bucket_emit_items=7,wide"]), variants(wide)
        assert h8(w.L) == f1 and (w.s_rank, w.e_rank) == info1
        r2, p2 = w.samples()
        assert (r2 == r1).all() and (p2 == p1).all()
        w.free()
    finally:
        wide.close()
    ctx.trim()


def test_n3_1e9_both_ends_sampled_beyond_2_53(ctx):
    """(b-a)(ns-1) beyond 2^53, where the product of the reference's sample test (include/bwt.h:339-340) no longer
    converts to double exactly: with the CLI's default ssa_n_samples at n = 3.1e9 the last position's quotient rounds
    below ns-1 there.  Both ends of every leaf are sampled all the same (:296-299), ns samples in all, in position
    order, each with L[rank] = T[pos-1]."""
    n = 3100000000
    ctx.gen_text(n, 42, 4)
    for ns in (15499999, 38750000, 2 ** 25 + 1):
        blk = ctx.pmbwt(n // 2, ns)
        r, p = blk.samples()
        assert len(p) == ns and p[0] == 0 and p[-1] == n - 1 and (np.diff(p) > 0).all()
        if ns == 15499999:
            L, T = blk.L, ctx.get_text()
            assert (L[r] == T[p - 1]).all()
            del L, T
        blk.free()
    ctx.trim()


def test_config4_repetitive_4_6e8(ctx):
    """Config 4 at full size: 100 copies of a 4.6e6-symbol genome with 0.1 % substitutions (LCPs of thousands),
    two leaves and one merge.  The reference's comparison sort does not finish this text in reasonable time, so:
    L a permutation of T, L[rank] = T[pos-1] at every sample, L decoded back to the text on the device, and the same
    L from four leaves (other windows, other chains)."""
    from oracle import gen_repetitive
    T = gen_repetitive(4600000, 100, 0.001, seed=43, sigma=4)
    n = len(T)
    ctx.set_text(T)
    ctx.set_timing(2)
    ctx.reset_timers()
    blk = ctx.pmbwt(n // 2 + 1, 1000000)
    deferred, probes, emits = launches(ctx, "gap_chains_deferred"), launches(ctx, "leaf:k_repeat_probe"), launches(ctx, "leaf:k_bucket_emit")
    ctx.set_timing(0)
    assert deferred == 1  # the chains' heads came from the left block's sampled suffix array
    assert probes == 2 and emits == 0  # both leaves were probed and went straight to the refinement rounds
    assert blk.ibwt_mismatches() == 0
    L = blk.L
    r, p = blk.samples()
    assert (np.bincount(L, minlength=5) == np.bincount(T, minlength=5)).all()
    assert L[blk.s_rank] == 0 and p[0] == 0 and p[-1] == n - 1
    assert (L[r] == T[p - 1]).all()
    _adjacent_ranks_increase(blk, T, L)
    fnv = h8(L)
    blk.free()
    b4 = ctx.pmbwt(n // 4 + 1, 1000000)
    assert h8(b4.L) == fnv
    b4.free()
    ctx.trim()


def test_config3_n3_1e9_eight_ranks_share_the_gpu(ctx):
    """Config 3's schedule at full size: bwt_hip_pmbwt_multi cuts 3.1e9 symbols over 8 ranks (here threads that share
    the one GPU: the partition of every merge, the two-bit transfers of L between ranks, slices at odd offsets and the
    samples that stay with their ranks are those of an 8-GPU run; only the copies stay on one device).  L, the special
    ranks and the samples equal the single-context build of the same 8-leaf tree."""
    import bwt_amd
    n, ns = 3100000000, 19375000
    th = -(-n // 8)
    ctx.gen_text(n, 42, 4)
    one = ctx.pmbwt(th, ns)
    f1, info1 = h8(one.L), (one.s_rank, one.e_rank)
    r1, p1 = one.samples()
    one.free()
    T = ctx.get_text()
    ctx.trim()
    L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * 8, T, th, ns)
    del T
    assert used == 8 and (s_rank, e_rank) == info1
    assert h8(L) == f1
    assert len(r) == len(r1) and (r == r1).all() and (p == p1).all()


def test_n3_1e9_one_leaf_over_eight_ranks_by_key_range(ctx):
    """One leaf of 3.1e9 symbols cut over 8 ranks by key range (threads that share the one GPU; every rank counts the
    whole text, keeps its stretch of first key digits and builds that slice of L, and the sample slots go once round
    all ranks): L, the special ranks and the samples equal the single-GPU build of the same one-leaf tree."""
    import bwt_amd
    n, ns = 3100000000, 19375000
    ctx.gen_text(n, 42, 4)
    one = ctx.pmbwt(n, ns)
    f1, info1 = h8(one.L), (one.s_rank, one.e_rank)
    r1, p1 = one.samples()
    one.free()
    T = ctx.get_text()
    ctx.trim()
    L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * 8, T, n, ns)
    del T
    assert used == 8 and (s_rank, e_rank) == info1
    assert h8(L) == f1
    assert len(r) == len(r1) and (r == r1).all() and (p == p1).all()


def test_runs_of_a_fifth_symbol_n1e9(ctx):
    """n = 1e9 of random DNA with 31 runs of a fifth symbol (1e4 to 3e7 symbols, one across the leaf boundary: the N of
    an assembly's gaps): the suffixes that begin inside the runs are left out of the leaves' sort and ranked in
    closed form (phase leaf_run_block), the keys use the four-letter key space.  Decoded back to the text on the
    device, L[rank] = T[pos-1] at every sample, adjacent suffixes in increasing order by host comparison, and the
    same L from four leaves."""
    n, run = 1000000000, 30000000
    rng = np.random.default_rng(5)
    T = rng.integers(1, 5, size=n, dtype=np.uint8)
    at = n // 2 - run // 3
    T[at:at + run] = 5
    for _ in range(30):
        length = int(10 ** rng.uniform(4, np.log10(run)))
        where = int(rng.integers(0, n - length - 1))
        T[where:where + length] = 5
    T[-1] = 0
    ctx.set_text(T)
    ctx.set_timing(2)
    ctx.reset_timers()
    blk = ctx.pmbwt(n // 2, 1000000)
    blocks, emits = launches(ctx, "leaf_run_block"), launches(ctx, "leaf:k_bucket_emit")
    ctx.set_timing(0)
    assert blocks == 4 and emits == 2  # both leaves on the lean route, each with a block of run suffixes
    assert blk.ibwt_mismatches() == 0
    L = blk.L
    r, p = blk.samples()
    assert (np.bincount(L, minlength=6) == np.bincount(T, minlength=6)).all()
    assert p[0] == 0 and p[-1] == n - 1 and (L[r] == T[p - 1]).all()
    _adjacent_ranks_increase(blk, T, L)
    fnv = h8(L)
    blk.free()
    b4 = ctx.pmbwt(n // 4, 1000000)
    assert h8(b4.L) == fnv
    b4.free()
    ctx.trim()
