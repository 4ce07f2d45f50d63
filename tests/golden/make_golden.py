#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the REAL reference.

Runs only where /root/reference is mounted: it drives oracle/_ref/libbwt_ref.so,
which oracle/Makefile compiles from the reference's own headers.  The fixture
holds data only -- text recipes, and hashes / small arrays of what the
reference computed for them -- never reference source.

    python tests/golden/make_golden.py        # rewrites golden.json

Hashes are FNV-1a-64 over the raw bytes (uint8 arrays) or over the
little-endian int64 image (index arrays): offset basis 0xcbf29ce484222325,
prime 0x100000001b3, one byte per step (tests/golden_util.py).  For the
splitmix64 seed-42 DNA text of 1e6 symbols that gives L = a5b14c6aea1f33b4 with
s_rank 471118.  Hash values quoted elsewhere for the same texts with the same
s_rank but other digits (901bd18da962c3e6 for this one) came from a throw-away
harness with a different hash function, not from a different L.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import Reference  # noqa: E402
from tests.golden_util import h8, h64, make_text  # noqa: E402


def block_record(blk, lo, hi, small):
    r, p = blk.samples()
    rec = {"a": int(blk.a), "b": int(blk.b), "s_rank": int(blk.s_rank), "e_rank": int(blk.e_rank),
           "e_char": int(blk.e_char), "L_fnv": h8(blk.L[lo:hi]), "n_samples": int(len(r)),
           "sample_rank_fnv": h64(r), "sample_pos_fnv": h64(p),
           "L_head": blk.L[lo:min(hi, lo + 64)].tolist(), "L_tail": blk.L[max(lo, hi - 64):hi].tolist()}
    if small:
        rec["L"] = blk.L[lo:hi].tolist()
        rec["sample_rank"] = r.tolist()
        rec["sample_pos"] = p.tolist()
    return rec


def pipeline_case(name, recipe, **kw):
    T = make_text(recipe)
    blk = Reference.pmbwt(T, **kw)
    rec = {"name": name, "text": recipe, "n": int(len(T)), "opts": kw,
           "bwt_rec_threshold": int(blk.threshold), "ssa_n_samples": int(blk.ssa_n_samples)}
    rec.update(block_record(blk, 0, len(T), len(T) <= 64))
    blk.init_extra()
    assert (blk.ibwt() == T).all()
    return rec


def phase_case(name, recipe, a, b, alpha_max, ns, seg=0):
    """Two leaves [a,c) and [c,b), their gap array, and the merge, with every intermediate."""
    T = make_text(recipe)
    n = len(T)
    c = (a + b) // 2
    L = np.zeros(n, dtype=np.uint8)
    small = n <= 64
    rec = {"name": name, "text": recipe, "n": n, "a": a, "c": c, "b": b, "alpha_max": alpha_max,
           "ssa_n_samples": ns, "seg": seg}
    sa_l = Reference.sa_range(T, a, c)
    sa_r = Reference.sa_range(T, c, b)
    rec["sa_left_fnv"], rec["sa_right_fnv"] = h64(sa_l), h64(sa_r)
    if small:
        rec["sa_left"], rec["sa_right"] = sa_l.tolist(), sa_r.tolist()
    lb = Reference.leaf(T, a, c, L, 1, alpha_max, ns, seg)
    rb = Reference.leaf(T, c, b, L, 1, alpha_max, ns, seg)
    rec["left"] = block_record(lb, a, c, small)
    rec["right"] = block_record(rb, c, b, small)
    # C[] and a rank grid of the left block (bwt.h:200-226, wavelet_matrix.h:260-290)
    lb.init_extra()
    C = lb.C()
    rec["C_left"] = C.tolist()
    m = c - a
    grid = sorted(set([0, 1, m // 3, m // 2, m - 1, m] + list(range(0, m + 1, max(1, m // 61)))))
    grid = [i for i in grid if 0 <= i <= m]
    syms = sorted(set(int(x) for x in np.unique(T[a:c]) if x != 0))[:8]
    rec["rank_grid_i"], rec["rank_grid_c"] = grid, syms
    rec["rank_grid"] = [[int(lb.rank(ch, i)) for i in grid] for ch in syms]
    rs = sorted(set([a, a + m // 2, c - 1] + list(range(a, c, max(1, m // 17)))))
    rec["sa_probe_r"] = rs
    rec["sa_probe"] = [int(lb.sa(r)) for r in rs]
    xs = sorted(set([c, b - 1, (c + b) // 2] + list(range(c, b, max(1, (b - c) // 13)))))
    rec["bin_search_x"] = xs
    rec["bin_search"] = [int(lb.bin_search(x)) for x in xs]
    # fresh blocks for the gap and the merge (init_extra above is not undone by the harness)
    del lb, rb
    L[:] = 0
    lb = Reference.leaf(T, a, c, L, 1, alpha_max, ns, seg)
    rb = Reference.leaf(T, c, b, L, 1, alpha_max, ns, seg)
    gap = Reference.gap(lb, rb)
    assert int(gap.sum()) == b - c
    rec["gap_fnv"], rec["gap_max"] = h64(gap), int(gap.max())
    if small:
        rec["gap"] = gap.tolist()
    mb = Reference.merge(lb, rb)
    rec["merged"] = block_record(mb, a, b, small)
    mb.init_extra()
    assert (mb.ibwt()[a:b] == T[a:b]).all()
    return rec


def main():
    assert Reference.available(), "build oracle/_ref first (make -C oracle)"
    out = {"generator": "tests/golden/make_golden.py",
           "source": "taura/bwt compiled serially from its own headers (oracle/ref_harness.cc)",
           "pipeline": [], "phase": [], "defaults": []}
    for n in (12, 1000, 99999, 10 ** 6, 10 ** 7, 10 ** 8, 10 ** 9, 3100000000):
        for mb in (0.0, 3.0, 4.0, 8.0, 18.0):
            th, ns = Reference.set_defaults(n, mb)
            out["defaults"].append({"n": n, "mem_budget": mb, "bwt_rec_threshold": th, "ssa_n_samples": ns})
    P = out["pipeline"]
    P.append(pipeline_case("mississippi", {"kind": "literal", "bytes": "mississippi"}))
    P.append(pipeline_case("abracadabra_t4", {"kind": "literal", "bytes": "abracadabraabracadabra"},
                           bwt_rec_threshold=4, ssa_n_samples=3))
    for sigma in (4, 255):
        for n in (1000, 100000, 1000000):
            P.append(pipeline_case("splitmix_s%d_n%d" % (sigma, n),
                                   {"kind": "splitmix", "n": n, "seed": 42, "sigma": sigma},
                                   alpha_min=1, alpha_max=sigma))
    P.append(pipeline_case("splitmix_s4_n100000_t3000",
                           {"kind": "splitmix", "n": 100000, "seed": 7, "sigma": 4},
                           alpha_min=1, alpha_max=4, bwt_rec_threshold=3000, ssa_n_samples=500, seg=64))
    P.append(pipeline_case("splitmix_s26_n50000_2leaves",
                           {"kind": "splitmix", "n": 50000, "seed": 9, "sigma": 26},
                           alpha_min=1, alpha_max=26, bwt_rec_threshold=25000, ssa_n_samples=1000))
    P.append(pipeline_case("repetitive_8x1000",
                           {"kind": "repetitive", "unit_len": 1000, "copies": 8, "rate": 0.002, "seed": 43, "sigma": 4},
                           alpha_min=1, alpha_max=4, bwt_rec_threshold=1000, ssa_n_samples=200, seg=128))
    P.append(pipeline_case("repetitive_300x40",
                           {"kind": "repetitive", "unit_len": 40, "copies": 300, "rate": 0.0, "seed": 5, "sigma": 4},
                           alpha_min=1, alpha_max=4, bwt_rec_threshold=3001, ssa_n_samples=100, seg=100))
    P.append(pipeline_case("skew_run",
                           {"kind": "skew", "n_left": 3000, "n_run": 2999, "run_char": 2, "seed": 3, "sigma": 4},
                           alpha_min=1, alpha_max=4, bwt_rec_threshold=3000, ssa_n_samples=64, seg=256))
    H = out["phase"]
    H.append(phase_case("mississippi_halves", {"kind": "literal", "bytes": "mississippi"}, 0, 12, 255, 4))
    H.append(phase_case("splitmix_s4_n20000_sub", {"kind": "splitmix", "n": 20000, "seed": 11, "sigma": 4},
                        1234, 17891, 4, 300, 128))
    H.append(phase_case("splitmix_s255_n20000_full", {"kind": "splitmix", "n": 20000, "seed": 12, "sigma": 255},
                        0, 20000, 255, 1000, 0))
    H.append(phase_case("splitmix_s4_n100000_full", {"kind": "splitmix", "n": 100000, "seed": 13, "sigma": 4},
                        0, 100000, 4, 1000, 0))
    H.append(phase_case("repetitive_8x1000_sub",
                        {"kind": "repetitive", "unit_len": 1000, "copies": 8, "rate": 0.002, "seed": 43, "sigma": 4},
                        500, 7600, 4, 150, 64))
    H.append(phase_case("skew_run_overflow",
                        {"kind": "skew", "n_left": 3000, "n_run": 2999, "run_char": 2, "seed": 3, "sigma": 4},
                        0, 6000, 4, 64, 256))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;",
          "max gap over phase cases:", max(c["gap_max"] for c in H))


if __name__ == "__main__":
    main()
