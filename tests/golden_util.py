"""Text recipes and hashes shared by the golden-vector generator and the tests that read golden.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import fnv1a64, gen_repetitive, gen_text  # noqa: E402

GOLDEN_JSON = os.path.join(ROOT, "tests", "golden", "golden.json")


def make_text(recipe):
    k = recipe["kind"]
    if k == "literal":
        return np.frombuffer(recipe["bytes"].encode("latin1") + b"\0", dtype=np.uint8).copy()
    if k == "splitmix":
        return gen_text(recipe["n"], recipe["seed"], recipe["sigma"])
    if k == "repetitive":
        return gen_repetitive(recipe["unit_len"], recipe["copies"], recipe["rate"], recipe["seed"], recipe["sigma"])
    if k == "skew":  # random left part, then a constant run: every right suffix falls into few gaps
        left = gen_text(recipe["n_left"] + 1, recipe["seed"], recipe["sigma"])[:-1]
        run = np.full(recipe["n_run"], recipe["run_char"], dtype=np.uint8)
        return np.concatenate([left, run, np.zeros(1, dtype=np.uint8)])
    raise ValueError(k)


def h64(a):
    return "%016x" % fnv1a64(np.ascontiguousarray(a, dtype=np.int64).view(np.uint8))


def h8(a):
    return "%016x" % fnv1a64(np.ascontiguousarray(a, dtype=np.uint8))


def load_golden():
    with open(GOLDEN_JSON) as f:
        return json.load(f)


def check_block(blk, L, rec, lo, hi):
    """Compare a block (anything with s_rank/e_rank/e_char/samples()) and its L[lo:hi] with a golden record."""
    assert int(blk.a) == rec["a"] and int(blk.b) == rec["b"]
    assert int(blk.s_rank) == rec["s_rank"]
    assert int(blk.e_rank) == rec["e_rank"]
    assert int(blk.e_char) == rec["e_char"]
    Ls = np.asarray(L[lo:hi])
    assert Ls[:64].tolist() == rec["L_head"]
    assert Ls[-64:].tolist() == rec["L_tail"]
    assert h8(Ls) == rec["L_fnv"]
    r, p = blk.samples()
    assert len(r) == rec["n_samples"]
    assert h64(p) == rec["sample_pos_fnv"]
    assert h64(r) == rec["sample_rank_fnv"]
