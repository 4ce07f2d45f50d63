#!/usr/bin/env python3
"""Generate tests/golden/golden_large.json from the REAL reference: whole-pipeline records at the sizes
where the GPU build takes its large-window routes (n = 1e7 and 1e8), which the serial reference needs
minutes for.  Same record layout as the "pipeline" list of golden.json (make_golden.py); data only.

    python tests/golden/make_golden_large.py [--jobs 5] [--group pipeline|leaf|all]

Every case runs in a process of its own (the reference is serial; the cases are independent).  A group that
is not regenerated keeps the records the file already holds.

"leaf" holds single bwt_leaf calls (include/bwt.h:615-635) on windows of 3.6e8 to 5.6e8 suffixes: the sizes
at which the GPU leaf sort's 65536 buckets hold 5500 to 8600 suffixes each (about 15 minutes of std::sort
per case here).
"""
import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# name, recipe, keyword arguments of Reference.pmbwt
CASES = [
    ("splitmix_s4_n1e7_default", {"kind": "splitmix", "n": 10 ** 7, "seed": 42, "sigma": 4},
     {"alpha_min": 1, "alpha_max": 4}),
    ("splitmix_s255_n1e7_default", {"kind": "splitmix", "n": 10 ** 7, "seed": 42, "sigma": 255},
     {"alpha_min": 1, "alpha_max": 255}),
    ("splitmix_s4_n1e8_1leaf", {"kind": "splitmix", "n": 10 ** 8, "seed": 42, "sigma": 4},
     {"alpha_min": 1, "alpha_max": 4, "bwt_rec_threshold": 10 ** 8, "ssa_n_samples": 500000}),
    ("splitmix_s4_n1e8_2leaves", {"kind": "splitmix", "n": 10 ** 8, "seed": 42, "sigma": 4},
     {"alpha_min": 1, "alpha_max": 4, "bwt_rec_threshold": 5 * 10 ** 7, "ssa_n_samples": 1250000}),
    ("splitmix_s4_n1e8_default", {"kind": "splitmix", "n": 10 ** 8, "seed": 42, "sigma": 4},
     {"alpha_min": 1, "alpha_max": 4}),
]


# name, recipe, a, b, ssa_n_samples: one leaf.  b < n gives the GPU sort a window that is longer than the leaf.
LEAF_CASES = [
    ("leaf_s4_whole_5.3e8", {"kind": "splitmix", "n": 530000000, "seed": 42, "sigma": 4}, 0, 530000000, 1000000),
    ("leaf_s4_whole_5.6e8", {"kind": "splitmix", "n": 560000000, "seed": 42, "sigma": 4}, 0, 560000000, 1000000),
    ("leaf_s4_first_5e8_of_5.01e8", {"kind": "splitmix", "n": 501000000, "seed": 42, "sigma": 4}, 0, 500000000, 1000000),
    ("leaf_s4_mid_3.6e8", {"kind": "splitmix", "n": 400000000, "seed": 7, "sigma": 4}, 20000000, 380000000, 500000),
]


def run_leaf_case(case):
    name, recipe, a, b, ns = case
    import numpy as np
    from oracle import Reference
    from tests.golden.make_golden import block_record
    from tests.golden_util import make_text
    t0 = time.time()
    T = make_text(recipe)
    L = np.zeros(len(T), dtype=np.uint8)
    blk = Reference.leaf(T, a, b, L, 1, recipe["sigma"], ns, 0)
    rec = {"name": name, "text": recipe, "n": int(len(T)), "ssa_n_samples": ns}
    rec.update(block_record(blk, a, b, False))
    rec["reference_seconds"] = round(time.time() - t0, 1)
    sys.stderr.write("%s: %.0f s\n" % (name, time.time() - t0))
    return rec


def run_case(case):
    name, recipe, kw = case
    from oracle import Reference
    from tests.golden.make_golden import block_record
    from tests.golden_util import make_text
    t0 = time.time()
    T = make_text(recipe)
    fd = os.open(os.devnull, os.O_WRONLY)  # the reference prints its clocks line with printf
    saved = os.dup(1)
    os.dup2(fd, 1)
    try:
        blk = Reference.pmbwt(T, **kw)
    finally:
        os.dup2(saved, 1)
    rec = {"name": name, "text": recipe, "n": int(len(T)), "opts": kw,
           "bwt_rec_threshold": int(blk.threshold), "ssa_n_samples": int(blk.ssa_n_samples)}
    rec.update(block_record(blk, 0, len(T), False))
    rec["reference_seconds"] = round(time.time() - t0, 1)
    sys.stderr.write("%s: %.0f s\n" % (name, time.time() - t0))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=5)
    ap.add_argument("--group", default="all", choices=["all", "pipeline", "leaf"])
    ap.add_argument("--only", default="", help="comma-separated case names: regenerate these and keep the group's others")
    args = ap.parse_args()
    from oracle import Reference
    assert Reference.available(), "build oracle/_ref first (make -C oracle)"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_large.json")
    out = {"pipeline": [], "leaf": []}
    if os.path.exists(path):
        with open(path) as f:
            out.update(json.load(f))
    out["generator"] = "tests/golden/make_golden_large.py"
    out["source"] = "taura/bwt compiled serially from its own headers (oracle/ref_harness.cc)"
    with Pool(args.jobs) as pool:
        only = set(x for x in args.only.split(",") if x)
        for group, fn, cases in (("pipeline", run_case, CASES), ("leaf", run_leaf_case, LEAF_CASES)):
            if args.group not in ("all", group):
                continue
            todo = [c for c in cases if not only or c[0] in only]
            new = {r["name"]: r for r in pool.map(fn, todo, chunksize=1)}
            old = {r["name"]: r for r in out[group]} if only else {}
            old.update(new)
            out[group] = [old[c[0]] for c in cases if c[0] in old]
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
