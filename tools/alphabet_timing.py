"""Build times of two-leaf trees over n symbols for alphabets of 3 to 7 symbols (DNA with N), each decoded back to the
text on the device.  python tools/alphabet_timing.py [n] [sigmas...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 9
sigmas = [int(x) for x in sys.argv[2:]] or [3, 4, 5, 6, 7]
ctx = bwt_amd.Context(0)
for sigma in sigmas:
    ctx.gen_text(n, 100 + sigma, sigma)
    th, ns = -(-n // 2), n // 200
    ctx.pmbwt(th, ns).free()
    ctx.sync()
    ctx.set_timing(2)
    ctx.reset_timers()
    t0 = time.perf_counter()
    blk = ctx.pmbwt(th, ns)
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.set_timing(0)
    routes = sorted(k for k, v in ctx.timers().items() if k.startswith("leaf:bucket_emit_items=") and v[2] > 0)
    top = sorted(((v[0], k) for k, v in ctx.timers().items() if ":" not in k and "=" not in k), reverse=True)[:5]
    ok = blk.ibwt_mismatches() == 0
    blk.free()
    ctx.trim()
    print("n=%d sigma=%d: %.1f ms round_trip=%s routes=%s top=%s" % (n, sigma, dt * 1e3, ok, routes, [(k, round(ms, 1)) for ms, k in top]), flush=True)
