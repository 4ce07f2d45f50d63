"""Bucket route against the plain radix passes on the same random texts (several seeds, sizes and leaf counts):
L, special ranks and samples must agree.  python tools/soak_bucket_route.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bwt_amd
res = {}
for mode in ("bucket", "plain"):
    if mode == "bucket":
        os.environ["BWT_HIP_BUCKET_MIN_N"] = "1"
    else:
        os.environ.pop("BWT_HIP_BUCKET_MIN_N", None)
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    for seed in (1, 2, 3):
        for n, leaves in ((4000000, 1), (6500000, 2), (9000001, 3), (70000000, 1)):
            ctx.gen_text(n, seed, 4)
            ctx.reset_timers()
            b = ctx.pmbwt(-(-n // leaves), 7777)
            L = b.L
            r, p = b.samples()
            key = (seed, n, leaves)
            used = ctx.timers().get("leaf:k_bucket_emit", (0, 0, 0))[2]
            sig = (bwt_amd.__name__, int(L.astype(np.uint64).dot(np.arange(1, len(L) + 1, dtype=np.uint64) % 1000003)), b.s_rank, b.e_rank, int(r.sum()), int(p.sum()))
            if mode == "bucket":
                assert used > 0, key
                res[key] = (sig, L.copy() if n < 1e7 else None)
            else:
                assert used == 0 or n >= 100000000, (key, used)
                assert res[key][0] == sig, key
                if res[key][1] is not None:
                    assert (res[key][1] == L).all(), key
            b.free()
            print(mode, key, "ok", flush=True)
    ctx.close()
print("soak ok")
