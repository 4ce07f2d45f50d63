"""Device memory at its peak for builds of one text as 1, 2 and 8 leaves, in bytes per symbol (what
bwt_hip_leaf_capacity's estimate has to stay above): python tools/leaf_memory.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3100000000
for leaves in (1, 2, 8):
    ctx = bwt_amd.Context(0)
    ctx.gen_text(n, 42, 4)
    blk = ctx.pmbwt(-(-n // leaves), max(1000, n // 80))
    ctx.sync()
    peak = ctx.mem_stats()["peak"]
    print("n=%d leaves=%d: peak %.2f GB = %.2f bytes per symbol" % (n, leaves, peak / 1e9, peak / n), flush=True)
    blk.free()
    ctx.close()
