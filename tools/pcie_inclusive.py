"""One build with the text coming from host memory and L going back to host memory (what bwt::pmbwt does for a
caller with host buffers), beside the device-resident time.  python tools/pcie_inclusive.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000000
ctx = bwt_amd.Context(0)
ctx.gen_text(n, 42, 4)
T = ctx.get_text()  # host copy of the same text
th = -(-n // 2)
ns = max(1000, n // 80)
ctx.pmbwt(th, ns, 0).free()  # warm-up: memory pool, code objects
for rep in range(2):
    ctx.sync()
    t0 = time.perf_counter()
    ctx.set_text(T)
    t1 = time.perf_counter()
    blk = ctx.pmbwt(th, ns, 0)
    ctx.sync()
    t2 = time.perf_counter()
    L = blk.L
    r, p = blk.samples()
    t3 = time.perf_counter()
    blk.free()
    print("n=%d: upload %.1f ms, build %.1f ms, download L and samples %.1f ms, total %.1f ms = %.1f Mbase/s"
          % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, (n - 1) / (t3 - t0) / 1e6), flush=True)
