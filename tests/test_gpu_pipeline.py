"""GPU parity of the whole path bwt::pmbwt (include/bwt.h:867-883) through the C-ABI: golden vectors of the real
reference, the oracle on seeded inputs, and size-independent properties at sizes the CPU cannot check."""
import numpy as np
import pytest

from oracle import Oracle
from tests.golden_util import check_block, h8, load_golden, make_text
from tests.gpu_util import ctx, inverse_bwt  # noqa: F401

pytestmark = pytest.mark.gpu
G = load_golden()


@pytest.mark.parametrize("rec", G["pipeline"], ids=lambda r: r["name"])
def test_pipeline_golden(ctx, rec):
    T = make_text(rec["text"])
    ctx.set_text(T)
    blk = ctx.pmbwt(rec["bwt_rec_threshold"], rec["ssa_n_samples"], rec["opts"].get("seg", 0))
    check_block(blk, blk.L, rec, 0, len(T))


def test_mississippi(ctx):
    # src/simplest_example.cc: BWT("mississippi\0") = ipssm\0pissii
    ctx.set_text(np.frombuffer(b"mississippi\0", dtype=np.uint8))
    for th in (12, 6, 3, 1):
        blk = ctx.pmbwt(th, 4)
        assert bytes(blk.L) == b"ipssm\0pissii"
        assert (blk.s_rank, blk.e_rank) == (5, 0)


@pytest.mark.parametrize("sigma,n,th,ns,seg", [(4, 200000, 200000, 1000, 0), (4, 200000, 30000, 1000, 0),
                                               (4, 200000, 1500, 300, 50), (255, 150000, 40000, 2000, 0),
                                               (26, 99991, 7000, 500, 333), (3, 5000, 100, 20, 16)])
def test_pipeline_against_oracle(ctx, sigma, n, th, ns, seg):
    """src/test/test_rec.cc re-expressed (recursion + round trip), plus exact equality with the oracle."""
    T = make_text({"kind": "splitmix", "n": n, "seed": 1000 + sigma, "sigma": sigma})
    ctx.set_text(T)
    blk = ctx.pmbwt(th, ns, seg)
    L = blk.L
    o = Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
    ob = Oracle.pmbwt(T, o)
    assert (L == ob.L).all()
    assert (blk.s_rank, blk.e_rank, blk.e_char) == (ob.s_rank, ob.e_rank, ob.e_char)
    for x, y in zip(blk.samples(), ob.samples()):
        assert (x == y).all()
    if n <= 200000:
        assert (inverse_bwt(L, blk.e_rank) == T).all()  # src/pmbwt.cc:64-81


def test_subrange_recursion(ctx):
    """bwt_rec on a proper sub-range [a,b) of the text, as src/test/test_rec.cc does with random ranges."""
    T = make_text({"kind": "splitmix", "n": 60000, "seed": 31, "sigma": 4})
    ctx.set_text(T)
    a, b = 1234, 54321
    blk = ctx.rec(a, b, 5000, 200, 128)
    o = Oracle.opt(len(T), alpha_min=1, alpha_max=4, bwt_rec_threshold=5000, ssa_n_samples=200, build_gap_segment_sz=128)
    Lo = np.zeros(len(T), dtype=np.uint8)

    def rec(lo, hi):
        if hi - lo <= 5000:
            return Oracle.leaf(T, lo, hi, Lo, o)
        mid = (lo + hi) // 2
        return Oracle.merge(rec(lo, mid), rec(mid, hi), o)
    ob = rec(a, b)
    assert (blk.L == Lo[a:b]).all()
    assert (blk.s_rank, blk.e_rank) == (ob.s_rank, ob.e_rank)
    for x, y in zip(blk.samples(), ob.samples()):
        assert (x == y).all()


def _properties(ctx, n, sigma, seed, thresholds, ns):
    """Size-independent checks: L is a permutation of T, does not depend on the leaf size, and every sample
    (rank, pos) satisfies L[rank] = T[pos-1]."""
    ctx.gen_text(n, seed, sigma)
    T = ctx.get_text()
    ref = None
    for th in thresholds:
        blk = ctx.pmbwt(th, ns)
        L = blk.L
        assert (np.bincount(L, minlength=256) == np.bincount(T, minlength=256)).all()
        assert L[blk.s_rank] == 0 and blk.e_rank == 0
        r, p = blk.samples()
        assert p[0] == 0 and p[-1] == n - 1 and (np.diff(p) > 0).all()
        assert (L[r] == T[p - 1]).all()
        if ref is None:
            ref = L
        else:
            assert (L == ref).all()
        blk.free()
    return T, ref


def test_properties_dna_16m(ctx):
    n = 1 << 24
    T, L = _properties(ctx, n, 4, 42, [n, n // 2, n // 5], 100000)
    assert (T == make_text({"kind": "splitmix", "n": n, "seed": 42, "sigma": 4})).all()


def test_properties_bytes_8m(ctx):
    n = 1 << 23
    _properties(ctx, n, 255, 42, [n, n // 3], 50000)


def test_golden_1e6_from_device_generator(ctx):
    """The device text generator and the CPU one are the same stream: same golden hash as the reference."""
    rec = [r for r in G["pipeline"] if r["name"] == "splitmix_s4_n1000000"][0]
    ctx.gen_text(1000000, 42, 4)
    blk = ctx.pmbwt(rec["bwt_rec_threshold"], rec["ssa_n_samples"])
    assert h8(blk.L) == rec["L_fnv"] and blk.s_rank == rec["s_rank"]


@pytest.mark.parametrize("n,sigma,th,ns", [(12, 4, 3, 4), (5000, 3, 700, 50), (2000000, 4, 600000, 20000), (70000000, 4, 20000000, 100000),
                                           (3000000, 255, 3000000, 5000)])
def test_pmbwt_host_overlapped_upload(ctx, n, sigma, th, ns):
    """bwt_hip_pmbwt_host (text uploaded in chunks while the first leaf is sorted, L downloaded by the same call) gives
    what set_text + pmbwt + get_L give, and leaves the text in the context."""
    T = make_text({"kind": "splitmix", "n": n, "seed": 77, "sigma": sigma})
    Lh = np.zeros(n, dtype=np.uint8)
    blk = ctx.pmbwt_host(T, th, ns, 0, Lh)
    assert (blk.L == Lh).all() and blk.ibwt_mismatches() == 0
    ctx.set_text(T)
    ref = ctx.pmbwt(th, ns)
    assert (ref.L == Lh).all() and (ref.s_rank, ref.e_rank) == (blk.s_rank, blk.e_rank)
    for x, y in zip(blk.samples(), ref.samples()):
        assert (x == y).all()
    if n <= 5000:
        o = Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=th, ssa_n_samples=ns)
        assert (Oracle.pmbwt(T, o).L == Lh).all()


@pytest.fixture()
def pack_ctx(monkeypatch):
    """A context of its own that packs every eligible text whatever its size, in chunks of 1M symbols, so that
    small texts exercise many chunks, the reuse of the staging buffers and ragged tails."""
    import bwt_amd
    monkeypatch.setenv("BWT_HIP_HOST_PACK", "2")
    monkeypatch.setenv("BWT_HIP_XFER_CHUNK", str(1 << 20))
    monkeypatch.setenv("BWT_HIP_POISON_TEXT", "1")  # a read ahead of the upload must show
    c = bwt_amd.Context(0)
    yield c
    c.close()


def _host_equals_device_path(c, ref_ctx, T, th, ns):
    n = len(T)
    Lh = np.full(n, 0xEE, dtype=np.uint8)
    blk = c.pmbwt_host(T, th, ns, 0, Lh)
    ref_ctx.set_text(T)
    ref = ref_ctx.pmbwt(th, ns)
    assert (ref.L == Lh).all() and (blk.L == Lh).all()
    assert (ref.s_rank, ref.e_rank) == (blk.s_rank, blk.e_rank)
    for x, y in zip(blk.samples(), ref.samples()):
        assert (x == y).all()
    assert (c.get_text() == T).all()


@pytest.mark.parametrize("n", [2, 3, 5, 17, 64, 4099, 1 << 20, (1 << 20) + 1, (1 << 20) + 2, 7777777])
def test_pmbwt_host_packed_transfer_sizes(ctx, pack_ctx, n):
    """Texts of four symbol values cross the link as two bits per symbol (host_transfer.hip): same block and same L
    as the device-resident path, at lengths around the packing groups' and the chunks' edges."""
    T = make_text({"kind": "splitmix", "n": n, "seed": 5 + n, "sigma": 4})
    _host_equals_device_path(pack_ctx, ctx, T, n if n < 5 else n // 3, max(2, min(n // 3, 5000)))


@pytest.mark.parametrize("shift,sigma,n", [(96, 4, 3333333), (0, 3, 3333333), (200, 2, 2222222), (251, 4, 3333333), (0, 1, 100001)])
def test_pmbwt_host_packed_transfer_alphabets(ctx, pack_ctx, shift, sigma, n):
    """Any four consecutive symbol values pack (a..d, 252..255), fewer than four too."""
    T = make_text({"kind": "splitmix", "n": n, "seed": 99, "sigma": sigma})
    T[:-1] += shift
    _host_equals_device_path(pack_ctx, ctx, T, n // 2, 4000)


def test_pmbwt_host_packed_transfer_foreign_symbols(ctx, pack_ctx):
    """The sample that decides on packing, and that the build's symbol statistics are taken from, can miss rare
    symbols: the uploader meets them before the build reads their chunk, and the call starts over unpacked."""
    n = 5000011
    T = make_text({"kind": "splitmix", "n": n, "seed": 3, "sigma": 4})
    T[:-1] += 64
    for at, v in ((1234567, 64 + 9), (1234568, 7), (4999999, 255), (n - 2, 70)):
        T[at] = v
    _host_equals_device_path(pack_ctx, ctx, T, n // 2, 4000)
    # four values in all, but not the ones the sample saw
    T = make_text({"kind": "splitmix", "n": n, "seed": 4, "sigma": 3})
    T[:-1] += 1  # 2..4 ... and one 1 the sample does not see
    T[2500001] = 1
    _host_equals_device_path(pack_ctx, ctx, T, n // 2, 4000)


def test_pmbwt_host_packed_rejects_bad_texts(pack_ctx):
    import bwt_amd
    n = 3000000
    T = make_text({"kind": "splitmix", "n": n, "seed": 3, "sigma": 4})
    bad = T.copy()
    bad[1500000] = 0
    with pytest.raises(bwt_amd.BwtHipError):
        pack_ctx.pmbwt_host(bad, n, 100)
    bad = T.copy()
    bad[-1] = 2
    with pytest.raises(bwt_amd.BwtHipError):
        pack_ctx.pmbwt_host(bad, n, 100)
    Lh = np.zeros(n, dtype=np.uint8)
    blk = pack_ctx.pmbwt_host(T, n, 100, 0, Lh)
    assert blk.ibwt_mismatches() == 0 and (blk.L == Lh).all()


def test_pmbwt_host_rejects_bad_texts(ctx):
    import bwt_amd
    for bad in (b"abc", b"a\0b\0", b"\0\0"):
        with pytest.raises(bwt_amd.BwtHipError):
            ctx.pmbwt_host(np.frombuffer(bad, dtype=np.uint8).copy(), 10, 2)
    # the context is usable afterwards
    ctx.set_text(np.frombuffer(b"banana\0", dtype=np.uint8))
    assert bytes(ctx.pmbwt(10, 2).L) == b"annb\0aa"


def test_large_sample_set_reaches_the_host_intact(ctx):
    """Sample sets of 2^20 entries and more cross the link as 32-bit words in chunks and are widened by host threads
    (bwt_hip_block_get_samples): ranks and positions of the oracle, for a block that does not start at 0 too."""
    from oracle import gen_text
    n, ns = 2500000, 1100000
    T = gen_text(n, 21, 4)
    ctx.set_text(T)
    o = Oracle.opt(n, alpha_min=1, alpha_max=4, bwt_rec_threshold=n // 2 + 1, ssa_n_samples=ns)
    ob = Oracle.pmbwt(T, o)
    blk = ctx.pmbwt(n // 2 + 1, ns)
    r, p = blk.samples()
    orr, op = ob.samples()
    assert len(r) == ns and (r == orr).all() and (p == op).all()
    blk.free()
    # a leaf of its own range: ranks are relative to the block + its first position
    a, b = 100001, 2400000
    leaf = ctx.leaf(a, b, b - a)  # every position sampled
    r, p = leaf.samples()
    assert len(p) == b - a and (p == np.arange(a, b)).all()
    L = leaf.L
    assert (L[r - a] == T[p - 1]).all()
    leaf.free()
