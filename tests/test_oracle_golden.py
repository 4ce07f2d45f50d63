"""Pins the CPU oracle (oracle/pmbwt_oracle.c) to the golden vectors the real reference produced.

Each golden record was written by tests/golden/make_golden.py from taura/bwt compiled from its own
headers.  The properties checked per component follow the reference's own standalone tests
(src/test/test_leaf.cc, test_merge.cc, test_rec.cc, test_ssa.cc, test_wavelet_matrix.cc,
test_suffix_array.cc) and src/simplest_example.cc's known answer.
"""
import numpy as np
import pytest

from oracle import Oracle, Reference
from tests.golden_util import check_block, h64, load_golden, make_text

G = load_golden()


def test_mississippi_known_answer():
    # src/simplest_example.cc:5-22 prints BWT("mississippi\0") = ipssm\0pissii
    T = np.frombuffer(b"mississippi\0", dtype=np.uint8).copy()
    blk = Oracle.pmbwt(T, Oracle.opt(len(T)))
    assert bytes(blk.L) == b"ipssm\0pissii"
    assert blk.s_rank == 5 and blk.e_rank == 0
    rec = [c for c in G["pipeline"] if c["name"] == "mississippi"][0]
    assert rec["L"] == list(b"ipssm\0pissii")


@pytest.mark.parametrize("d", G["defaults"], ids=lambda d: "n%d_mb%g" % (d["n"], d["mem_budget"]))
def test_set_defaults(d):
    # include/opts.h:271-290
    o = Oracle.opt(d["n"], mem_budget=d["mem_budget"])
    assert o.bwt_rec_threshold == d["bwt_rec_threshold"]
    assert o.ssa_n_samples == d["ssa_n_samples"]


def _opt_for(rec):
    kw = dict(rec["opts"])
    seg = kw.pop("seg", 0)
    return Oracle.opt(rec["n"], build_gap_segment_sz=seg, **kw)


@pytest.mark.parametrize("rec", G["pipeline"], ids=lambda r: r["name"])
def test_pipeline(rec):
    T = make_text(rec["text"])
    assert len(T) == rec["n"]
    o = _opt_for(rec)
    assert o.bwt_rec_threshold == rec["bwt_rec_threshold"]
    assert o.ssa_n_samples == rec["ssa_n_samples"]
    blk = Oracle.pmbwt(T, o)
    check_block(blk, blk.L, rec, 0, len(T))
    if len(T) <= 200000:
        # L is *the* BWT of T whatever the leaf size (one comparison sort of all suffixes)
        assert (Oracle.bwt_direct(T) == blk.L).all()
        blk.init_extra()
        assert (blk.ibwt() == T).all()  # src/pmbwt.cc:64-81


@pytest.mark.parametrize("rec", G["phase"], ids=lambda r: r["name"])
def test_phases(rec):
    T = make_text(rec["text"])
    a, c, b = rec["a"], rec["c"], rec["b"]
    o = Oracle.opt(len(T), alpha_min=1, alpha_max=rec["alpha_max"], ssa_n_samples=rec["ssa_n_samples"],
                   build_gap_segment_sz=rec["seg"])
    L = np.zeros(len(T), dtype=np.uint8)
    lb, sa_l = Oracle.leaf(T, a, c, L, o, want_sa=True)
    rb, sa_r = Oracle.leaf(T, c, b, L, o, want_sa=True)
    assert h64(sa_l) == rec["sa_left_fnv"] and h64(sa_r) == rec["sa_right_fnv"]
    check_block(lb, L, rec["left"], a, c)
    check_block(rb, L, rec["right"], c, b)
    lb.init_extra()
    assert lb.C().tolist() == rec["C_left"]
    for ch, row in zip(rec["rank_grid_c"], rec["rank_grid"]):
        assert [lb.rank(ch, i) for i in rec["rank_grid_i"]] == row
    assert [lb.sa(r) for r in rec["sa_probe_r"]] == rec["sa_probe"]  # src/test/test_ssa.cc
    assert [lb.bin_search(x) for x in rec["bin_search_x"]] == rec["bin_search"]
    # the block is consumed by merge with init_extra undone inside; make fresh ones
    L2 = np.zeros(len(T), dtype=np.uint8)
    lb2 = Oracle.leaf(T, a, c, L2, o)
    rb2 = Oracle.leaf(T, c, b, L2, o)
    m, gap, g = Oracle.merge(lb2, rb2, o, want_taps=True)
    assert h64(gap) == rec["gap_fnv"] and int(gap.max()) == rec["gap_max"]
    # g_j is the insertion point of T[j:] among the left suffixes; its histogram is the gap array
    assert (np.bincount(g - a, minlength=c - a + 1) == gap).all()
    check_block(m, L2, rec["merged"], a, b)
    m.init_extra()
    assert (m.ibwt()[a:b] == T[a:b]).all()  # src/test/test_merge.cc


def test_suffix_array_sorted_pairs():
    # src/test/test_suffix_array.cc:27-48: adjacent SA entries are strictly increasing suffixes
    T = make_text({"kind": "splitmix", "n": 5000, "seed": 21, "sigma": 3})
    SA = Oracle.sa_range(T, 100, 4900)
    assert sorted(SA.tolist()) == list(range(100, 4900))
    tb = T.tobytes()
    for k in range(0, len(SA) - 1, 37):
        assert tb[SA[k]:] < tb[SA[k + 1]:]


def test_wavelet_rank_against_counts():
    # src/test/test_wavelet_matrix.cc:4-45: rank(c,p) equals the plain count, alphabet 'a'..'z'
    T = make_text({"kind": "splitmix", "n": 3000, "seed": 5, "sigma": 26})
    T[:-1] += ord("a") - 1
    o = Oracle.opt(len(T), alpha_min=ord("a"), alpha_max=ord("z"))
    blk = Oracle.pmbwt(T, o)
    blk.init_extra()
    Lh = blk.L.copy()
    Lh[blk.s_rank] = 0  # the hidden T[a-1], include/bwt.h:234-236
    for c in (ord("a"), ord("m"), ord("z")):
        cum = np.concatenate([[0], np.cumsum(Lh == c)])
        for p in range(0, len(T) + 1, 97):
            assert blk.rank(c, p) == cum[p]


@pytest.mark.skipif(not Reference.available(), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("sigma,n,th,ns,seg", [(4, 30000, 2000, 400, 100), (200, 30000, 9000, 1000, 0), (2, 4000, 500, 50, 32)])
def test_oracle_equals_live_reference(sigma, n, th, ns, seg):
    T = make_text({"kind": "splitmix", "n": n, "seed": 77 + sigma, "sigma": sigma})
    o = Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
    ob = Oracle.pmbwt(T, o)
    rb = Reference.pmbwt(T, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=th, ssa_n_samples=ns, seg=seg)
    assert (ob.L == rb.L).all()
    assert (ob.s_rank, ob.e_rank) == (rb.s_rank, rb.e_rank)
    for x, y in zip(ob.samples(), rb.samples()):
        assert (x == y).all()
