"""Bucket route against the plain radix passes at full size (n up to 1.3e9, several leaf counts and seeds): checksums of
L, special ranks and samples must agree, and L must hold the text's symbols.  python tools/soak_full_size.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bwt_amd  # noqa: E402

CASES = ((1000000000, 2, 5), (500000001, 1, 6), (1300000000, 4, 7), (1000000000, 2, 8))
res = {}
for mode in ("bucket", "plain"):
    if mode == "plain":
        os.environ["BWT_HIP_BUCKET_MIN_N"] = "4000000000"
    else:
        os.environ.pop("BWT_HIP_BUCKET_MIN_N", None)
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    for n, leaves, seed in CASES:
        ctx.gen_text(n, seed, 4)
        ctx.reset_timers()
        b = ctx.pmbwt(-(-n // leaves), n // 100)
        L = b.L
        r, p = b.samples()
        used = ctx.timers().get("leaf:k_bucket_emit", (0, 0, 0))[2]
        assert (used > 0) == (mode == "bucket"), (mode, n, used)
        w = (np.arange(1, 1 << 20, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFF)
        acc = 0
        for i in range(0, len(L), len(w)):
            blk = L[i:i + len(w)].astype(np.uint64)
            acc = (acc * 1000003 + int(blk.dot(w[:len(blk)]))) % (1 << 61)
        sig = (acc, np.bincount(L, minlength=5).tolist(), b.s_rank, b.e_rank, len(r), int(r.sum()), int(p.sum()))
        key = (n, leaves, seed)
        if mode == "bucket":
            res[key] = sig
        else:
            assert res[key] == sig, (key, res[key], sig)
        b.free()
        print(mode, key, "ok", flush=True)
    ctx.close()
print("soak ok")
