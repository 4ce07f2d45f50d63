"""Texts whose symbols are scattered over the byte range (ASCII letters): the library holds them renumbered 1 .. sigma
and translates symbols wherever they enter or leave through host memory.  Everything a caller sees -- L, the text read
back, e_char, C[], rank and LF queries, decoded blocks, uploaded blocks, the host-to-host call, several ranks --
against the oracle on the text as it is."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import Oracle, gen_text
from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


def letters(n, seed, alphabet):
    """random text over the given byte values, 0-terminated"""
    T = gen_text(n, seed, len(alphabet))
    lut = np.zeros(256, dtype=np.uint8)
    lut[1:len(alphabet) + 1] = sorted(alphabet)
    return lut[T]


ACGT = [65, 67, 71, 84]
ACGTN_MIXED = [65, 67, 71, 78, 84, 97, 99, 103, 116]
EDGES = [1, 2, 9, 128, 254, 255]


def same_as_oracle(ctx, T, th, ns):
    n = len(T)
    ctx.set_text(T)
    blk = ctx.pmbwt(th, ns)
    ob = Oracle.pmbwt(T, Oracle.opt(n, alpha_min=1, alpha_max=int(T.max()), bwt_rec_threshold=th, ssa_n_samples=ns))
    assert (blk.L == ob.L).all()
    assert (blk.s_rank, blk.e_rank, blk.e_char) == (ob.s_rank, ob.e_rank, ob.e_char)
    for x, y in zip(blk.samples(), ob.samples()):
        assert (x == y).all()
    return blk


@pytest.mark.parametrize("alphabet", [ACGT, ACGTN_MIXED, EDGES], ids=["ACGT", "acgtnACGT", "edges"])
def test_pipeline_with_letters(ctx, alphabet):
    T = letters(200000, 61, alphabet)
    ctx.set_timing(2)
    ctx.reset_timers()
    blk = same_as_oracle(ctx, T, 70000, 1500)
    compact, code_of, symbol_of = ctx.symbol_codes()
    present = sorted(set(alphabet))
    assert compact and [int(code_of[v]) for v in present] == list(range(1, len(present) + 1))
    assert [int(symbol_of[k]) for k in range(1, len(present) + 1)] == present and code_of[66] == 0
    # the text comes back as it went in; a block decodes to the text's own symbols
    assert (ctx.get_text() == T).all()
    assert blk.ibwt_mismatches() == 0 and (blk.ibwt() == T).all()
    blk.free()
    ctx.set_timing(0)
    if alphabet is ACGT:  # two bits a symbol from here on: one 4-ary wavelet level, not two 16-ary ones
        assert "gap_chains" in ctx.timers()
    for th, ns in ((200000, 2), (9000, 200000)):
        same_as_oracle(ctx, T, th, ns).free()


def test_queries_and_counts_with_letters(ctx):
    """C[], rank(c, i) and LF(c, r) take and give the text's own symbols (src/test/test_wavelet_matrix.cc's checks)."""
    n = 30000
    T = letters(n, 62, ACGTN_MIXED)
    a, b = 1000, 26000
    ctx.set_text(T)
    o = Oracle.opt(n, alpha_min=1, alpha_max=int(T.max()), ssa_n_samples=100)
    Lo = np.zeros(n, dtype=np.uint8)
    ob = Oracle.leaf(T, a, b, Lo, o)
    ob.init_extra()
    gb = ctx.leaf(a, b, 100)
    gb.init_extra()
    assert gb.e_char == ob.e_char
    C = gb.C(int(T.max()))
    seg = T[a:b]
    assert all(C[k] == int((seg < k).sum()) for k in range(int(T.max()) + 1))
    rng = np.random.RandomState(1)
    syms = np.array(sorted(ACGTN_MIXED), dtype=np.uint8)[rng.randint(0, len(ACGTN_MIXED), size=1500)]
    pos = rng.randint(0, b - a + 1, size=1500)
    exp = np.array([ob.rank(int(c), int(p)) for c, p in zip(syms, pos)])
    assert (gb.rank(syms, pos) == exp).all()
    ranks = np.arange(a, b, 7)
    Ls = Lo[ranks]
    keep = Ls != 0
    exp = np.array([ob.LF(int(c), int(r)) for c, r in zip(Ls[keep], ranks[keep])])
    assert (gb.LF(Ls[keep], ranks[keep]) == exp).all()
    import bwt_amd
    with pytest.raises(bwt_amd.BwtHipError):
        gb.rank(np.array([66], dtype=np.uint8), np.array([5]))  # 'B' is not in the text


def test_uploaded_blocks_with_letters(ctx):
    T = letters(20000, 63, ACGT)
    a, c, b = 0, 9000, 20000
    ctx.set_text(T)
    o = Oracle.opt(len(T), alpha_min=1, alpha_max=84, ssa_n_samples=300)
    Lo = np.zeros(len(T), dtype=np.uint8)
    ol, orr = Oracle.leaf(T, a, c, Lo, o), Oracle.leaf(T, c, b, Lo, o)
    ul = ctx.upload_block(a, c, Lo[a:c], ol.s_rank, ol.e_rank, *ol.samples())
    ur = ctx.upload_block(c, b, Lo[c:b], orr.s_rank, orr.e_rank, *orr.samples())
    m = Oracle.merge(ol, orr, o)
    mb = ctx.merge(ul, ur, 300)
    assert (mb.L == Lo[a:b]).all()
    for x, y in zip(mb.samples(), m.samples()):
        assert (x == y).all()


def test_host_call_and_ranks_with_letters(ctx):
    import bwt_amd
    T = letters(300000, 64, ACGT)
    n = len(T)
    ob = Oracle.pmbwt(T, Oracle.opt(n, alpha_min=1, alpha_max=84, bwt_rec_threshold=80000, ssa_n_samples=900))
    L = np.empty(n, dtype=np.uint8)
    blk = ctx.pmbwt_host(T, 80000, 900, 0, L)
    assert (L == ob.L).all() and (blk.L == ob.L).all() and blk.e_char == ob.e_char
    for x, y in zip(blk.samples(), ob.samples()):
        assert (x == y).all()
    blk.free()
    Lm, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * 4, T, 80000, 900)
    assert used == 4 and (Lm == ob.L).all() and (s_rank, e_rank) == (ob.s_rank, ob.e_rank)
    orr, op = ob.samples()
    assert (r == orr).all() and (p == op).all()


def test_switched_off_gives_the_same(tmp_path):
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import bwt_amd; from oracle import Oracle; "
            "from tests.test_gpu_sparse_alphabet import letters, ACGT; T = letters(60000, 65, ACGT); c = bwt_amd.Context(0); "
            "c.set_text(T); b = c.pmbwt(20000, 100); "
            "ob = Oracle.pmbwt(T, Oracle.opt(len(T), alpha_min=1, alpha_max=84, bwt_rec_threshold=20000, ssa_n_samples=100)); "
            "assert (b.L == ob.L).all() and b.e_char == ob.e_char; print('same')") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BWT_HIP_COMPACT="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "same" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_text_in_the_callers_device_memory():
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "text_device_worker.py")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "text device ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
