"""Times bwt_hip_pmbwt_host (text and L in host memory) at the headline size, packed and raw transfers, page-locked and
pageable host buffers, and checks every L against the device-resident build's.
Usage: python tools/host_transfer_timing.py [n] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3100000000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    th, ns = (n + 1) // 2, max(2, n // 80)
    ref_ctx = bwt_amd.Context(0)
    ref_ctx.gen_text(n, 42, 4)
    Tp = bwt_amd.HostBuffer(n)
    ref_ctx.read_text(Tp.a)
    ref = ref_ctx.pmbwt(th, ns, 1024)
    L_ref = ref.L
    info = (ref.s_rank, ref.e_rank)
    ref.free()
    ref_ctx.close()
    T_pageable = Tp.a.copy()
    Lp = bwt_amd.HostBuffer(n)
    L_pageable = np.zeros(n, dtype=np.uint8)
    for pack in ("1", "0"):
        os.environ["BWT_HIP_HOST_PACK"] = pack
        for name, T, L in (("page-locked", Tp.a, Lp.a), ("pageable", T_pageable, L_pageable)):
            os.environ["BWT_HIP_XFER_TRACE"] = "0"
            ctx = bwt_amd.Context(0)
            ctx.pmbwt_host(T, th, ns, 1024, L).free()  # warm: staging buffers, pools
            times = []
            for r in range(reps):
                L[:64] = 0
                if r == reps - 1:
                    os.environ["BWT_HIP_XFER_TRACE"] = "1"
                t0 = time.perf_counter()
                blk = ctx.pmbwt_host(T, th, ns, 1024, L)
                times.append((time.perf_counter() - t0) * 1e3)
                same = (blk.s_rank, blk.e_rank) == info
                blk.free()
            same = same and bool((L == L_ref).all())
            print("pack=%s %-11s n=%d: %s ms  (best %.1f ms = %.0f Mbase/s)  L equals the resident build's: %s"
                  % (pack, name, n, " ".join("%.1f" % t for t in times), min(times), (n - 1) / min(times) / 1e3, same), flush=True)
            ctx.close()
            if not same:
                sys.exit(1)


if __name__ == "__main__":
    main()
