"""Bucket route against the plain radix passes on alphabets of 2, 3, 4, 5 and 8 symbols (1-, 2- and 3-bit codes, with and
without the terminator's own code) and several window sizes: L, special ranks and samples must agree.
python tools/soak_bucket_alphabets.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

res = {}
for mode in ("bucket", "plain"):
    if mode == "bucket":
        os.environ["BWT_HIP_BUCKET_MIN_N"] = "1"
    else:
        os.environ.pop("BWT_HIP_BUCKET_MIN_N", None)
    ctx = bwt_amd.Context(0)
    ctx.set_timing(2)
    for sigma in (2, 3, 4, 5, 8):
        for n, leaves in ((300000, 2), (2200000, 1), (5000001, 3)):
            ctx.gen_text(n, 100 + sigma, sigma)
            ctx.reset_timers()
            b = ctx.pmbwt(-(-n // leaves), 4321)
            L = b.L
            r, p = b.samples()
            used = ctx.timers().get("leaf:k_bucket_emit", (0, 0, 0))[2] + ctx.timers().get("leaf:k_bucket_bounds", (0, 0, 0))[2]
            key = (sigma, n, leaves)
            if mode == "bucket":
                res[key] = (L.copy(), b.s_rank, b.e_rank, r.copy(), p.copy(), used)
            else:
                assert used == 0, key
                L0, s0, e0, r0, p0, u0 = res[key]
                assert (L0 == L).all() and (s0, e0) == (b.s_rank, b.e_rank) and (r0 == r).all() and (p0 == p).all(), key
                print(key, "ok (bucket route kernels launched: %d)" % u0, flush=True)
            b.free()
    ctx.close()
print("soak ok")
