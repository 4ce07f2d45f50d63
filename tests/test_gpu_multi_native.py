"""The C++ multi-GPU driver (bwt_hip_dist_pmbwt / bwt_hip_pmbwt_multi, include/bwt_hip.h): bwt::pmbwt
(reference include/bwt.h:867-883) with the tree cut at depth log2(ranks) and every upper merge spread over the
ranks of its subtree.  One GPU is at hand here, so the ranks are threads of this process that share it
(in-process transport): the schedule, the partition of every merge and the sample placement are the ones
several GPUs run; only the copies stay on one device.  Against the oracle and the single-GPU build."""
import numpy as np
import pytest

import bwt_amd
from oracle import Oracle, gen_repetitive, gen_text
from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


def _same_as_oracle(T, sigma, world, th, ns, seg=0):
    n = len(T)
    L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * world, T, th, ns, seg)
    o = Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
    ob = Oracle.pmbwt(T, o)
    assert (L == ob.L).all()
    assert (s_rank, e_rank) == (ob.s_rank, ob.e_rank)
    orr, op = ob.samples()
    assert (r == orr).all() and (p == op).all()
    return used


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_native_multi_against_oracle(world):
    T = gen_text(200000, 42, 4)
    assert _same_as_oracle(T, 4, world, 200000 // world // 2 + 1, 3000) == world  # two leaves per rank
    assert _same_as_oracle(T, 4, world, 200000 // world + 1, 777, 100) == world   # one leaf per rank


def test_native_multi_other_texts():
    assert _same_as_oracle(gen_text(150000, 7, 255), 255, 4, 20000, 2000) == 4
    assert _same_as_oracle(gen_text(99991, 9, 26), 26, 8, 7000, 500, 333) == 8
    assert _same_as_oracle(gen_repetitive(3000, 30, 0.001, seed=43, sigma=4), 4, 4, 12000, 700) == 4
    # L crosses between the ranks as two bits per symbol whenever the symbols span at most four values: letters
    # that are not 1..4, an alphabet of three, and lengths that leave every slice at an odd offset
    T = gen_text(123457, 5, 4)
    T[:-1] += 100
    assert _same_as_oracle(T, 104, 8, 123457 // 8 + 1, 900) == 8
    assert _same_as_oracle(gen_text(77773, 6, 3), 3, 4, 77773 // 8 + 1, 555, 50) == 4
    assert _same_as_oracle(gen_text(1001, 8, 2), 2, 8, 1001 // 8 + 1, 40, 16) == 8
    # skewed: nearly all right suffixes fall into few gaps, parts of very different sizes
    T = np.concatenate([gen_text(6001, 3, 4)[:-1], np.full(5999, 2, dtype=np.uint8), np.zeros(1, dtype=np.uint8)])
    assert _same_as_oracle(T, 4, 8, 1000, 300, 64) == 8


def test_fewer_subtrees_than_gpus():
    """A threshold that leaves two leaves feeds two of the four ranks."""
    T = gen_text(100000, 11, 4)
    assert _same_as_oracle(T, 4, 4, 50000, 1000) == 2
    assert _same_as_oracle(T, 4, 4, 100000, 1000) == 1


def test_native_multi_matches_single_gpu_at_2e7(ctx):
    n = 20000000
    ctx.gen_text(n, 42, 4)
    T = ctx.get_text()
    ns = 100000
    one = ctx.pmbwt(n // 8 + 1, ns)
    L1, (r1, p1) = one.L, one.samples()
    L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * 8, T, n // 8 + 1, ns)
    assert used == 8 and (L == L1).all() and (s_rank, e_rank) == (one.s_rank, one.e_rank)
    assert (r == r1).all() and (p == p1).all()
    one.free()


def test_rccl_transport_with_one_rank(ctx):
    """The RCCL transport comes up (library found, communicator created) and carries a one-rank build."""
    T = gen_text(50000, 5, 4)
    ctx.set_text(T)
    t = bwt_amd.Transport.rccl(ctx, bwt_amd.Transport.rccl_unique_id(), 0, 1)
    s = ctx.dist_pmbwt(t, 20000, 400)
    blk = ctx.pmbwt(20000, 400)
    assert (s.first, s.size) == (0, len(T)) and (s.L == blk.L).all() and (s.s_rank, s.e_rank) == (blk.s_rank, blk.e_rank)
    for x, y in zip(s.samples(), blk.samples()):
        assert (x == y).all()
    s.free()
    t.close()


@pytest.mark.parametrize("n,world", [(2 * 2 ** 21 - 6, 2), (2 * (2 ** 21 + 8191), 2), (2 * (2 ** 21 + 8192) + 1, 2), (4 * 2 ** 21 + 3, 4)])
def test_native_multi_grouped_g_exchange_boundaries(ctx, n, world):
    """Left blocks of 2^21 suffixes and more send their g values grouped by gap bucket (16-bit words + bucket bounds,
    parts cut at multiples of 8192 left ranks); smaller ones send the values.  Around the switch, with the last gap
    index at the end of a bucket and in a bucket of its own, with chains whose last one is short (fillers), and with
    both kinds of merge in one tree: the single-context build's L, special ranks and samples."""
    ctx.gen_text(n, 9, 4)
    T = ctx.get_text()
    th, ns = -(-n // world), 5000
    one = ctx.pmbwt(th, ns, 1000)
    L1, info1, (r1, p1) = one.L, (one.s_rank, one.e_rank), one.samples()
    one.free()
    L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * world, T, th, ns, 1000)
    assert used == world and (s_rank, e_rank) == info1
    assert (L == L1).all() and (r == r1).all() and (p == p1).all()


def test_layout_of_a_distributed_merge(ctx):
    """bwt_hip_dist_layout: parts are contiguous and cover the gap indices 0..m; below 2^21 left suffixes they are cut
    anywhere (shift 0, the bounds of bwt_hip_dist_bounds), from there on at multiples of 8192 (shift 13: the g values
    travel grouped by gap bucket) and still even to within a few buckets."""
    for m, parts in [(5, 2), (100000, 8), (2 ** 21 - 2, 4), (2 ** 21 - 1, 4), (2 ** 21 + 8191, 2), (387500000, 8), (1550000000, 8),
                     (2 ** 31 - 2, 2), (2 ** 31 - 1, 2), (3000000000, 4)]:
        lo, shift = ctx.dist_layout(m, parts)
        lo = [int(x) for x in lo]
        grouped = 2 ** 21 <= m + 1 < 2 ** 31
        assert shift == (13 if grouped else 0), (m, shift)
        assert lo[0] == 0 and lo[-1] == m + 1 and all(x <= y for x, y in zip(lo, lo[1:]))
        if not grouped:
            assert lo == [int(x) for x in bwt_amd.dist_bounds(m, parts)]
        else:
            assert all(x % 8192 == 0 for x in lo[:-1])
            sizes = [min(y, m + 1) - min(x, m + 1) for x, y in zip(lo, lo[1:])]
            assert max(sizes) - min(sizes) < 3 * 8192, (m, parts, sizes)  # whole buckets, and the last one may be short


def test_rccl_transport_moves_bytes(ctx):
    """ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd through the transport's one primitive: with a single rank the
    only peer is the rank itself, so a send and its receive in one batch move a block's L into another block's buffer."""
    import ctypes as C
    T = gen_text(300000, 5, 4)
    ctx.set_text(T)
    a, b = ctx.leaf(0, 100000, 10), ctx.leaf(100000, 200000, 10)
    want = a.L
    assert not (b.L == want).all()
    t = bwt_amd.Transport.rccl(ctx, bwt_amd.Transport.rccl_unique_id(), 0, 1)

    class P2P(C.Structure):  # bwt_hip_p2p
        _fields_ = [("peer", C.c_int), ("is_send", C.c_int), ("buf", C.c_void_p), ("bytes", C.c_int64)]

    class TR(C.Structure):   # bwt_hip_transport
        _fields_ = [("self", C.c_void_p), ("rank", C.c_int), ("world", C.c_int),
                    ("exchange", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(P2P))), ("destroy", C.c_void_p)]
    tr = C.cast(t._h, C.POINTER(TR)).contents
    assert (tr.rank, tr.world) == (0, 1)
    ops = (P2P * 2)(P2P(0, 1, a.L_device, 100000), P2P(0, 0, b.L_device, 100000))
    ctx.sync()
    assert tr.exchange(tr.self, 2, ops) == 0
    assert (b.L == want).all()
    t.close()
