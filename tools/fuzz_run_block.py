"""Random texts with long runs of one symbol on the forced lean route (suffixes inside the runs ranked in closed form)
against the oracle.  Test infrastructure: BWT_HIP_BUCKET_MIN_N=1 BWT_HIP_FORCE_WIDE=2 python tools/fuzz_run_block.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle  # noqa: E402


def main():
    assert os.environ.get("BWT_HIP_BUCKET_MIN_N") == "1" and os.environ.get("BWT_HIP_FORCE_WIDE") == "2"
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = bwt_amd.Context(0)
    ctx.set_timing(1)
    t0, cases, blocks = time.time(), 0, 0
    while time.time() - t0 < budget:
        n = int(rng.integers(60000, 400000))
        sigma = int(rng.choice([2, 3, 4, 4, 4, 5, 5, 6, 7, 8, 16, 255]))
        lo = int(rng.choice([1, 1, 65, 97])) if sigma < 100 else 1
        T = rng.integers(lo, lo + sigma, size=n, dtype=np.int64).astype(np.uint8)
        kind = rng.integers(0, 4)
        sym = lo + sigma - 1 if kind == 0 else (lo if kind == 1 else int(rng.integers(lo, lo + sigma)))
        if kind == 3 and lo + sigma < 256:  # a symbol of its own, behind the alphabet (as the N of an assembly coded behind the bases)
            sym = lo + sigma
        for _ in range(int(rng.integers(1, 7))):
            length = int(rng.integers(8200, 40000))
            at = int(rng.integers(0, max(1, n - length)))
            T[at:at + length] = sym
            if rng.random() < 0.3 and at + length + 8 < n:  # the same symbols behind two runs
                T[at + length:at + length + 8] = lo
        if rng.random() < 0.2:
            T[n - int(rng.integers(8200, 20000)):] = sym  # a run that ends the text
        if rng.random() < 0.3:  # the run symbol also scattered outside its runs, or (no runs kept) only scattered
            if rng.random() < 0.3:
                T = rng.integers(lo, lo + sigma, size=n, dtype=np.int64).astype(np.uint8)
                T[T == sym] = lo
            T[rng.integers(0, n - 1, size=int(rng.integers(1, 400)))] = sym
        if rng.random() < 0.3 and int(T.max()) < 120:  # symbols scattered over the byte range: held renumbered
            T = (T.astype(np.int64) * 2 + int(rng.integers(0, 10))).astype(np.uint8)
        T[-1] = 0
        leaves = int(rng.integers(1, 4))
        th = -(-n // leaves)
        ns = int(rng.integers(2, 3000))
        hi = int(T.max())
        before = ctx.timers().get("leaf_run_block", (0, 0, 0))[2]
        ctx.set_text(T)
        blk = ctx.pmbwt(th, ns)
        ob = Oracle.pmbwt(T, Oracle.opt(n, alpha_min=1, alpha_max=hi, bwt_rec_threshold=th, ssa_n_samples=ns))
        ok = (blk.L == ob.L).all() and (blk.s_rank, blk.e_rank) == (ob.s_rank, ob.e_rank)
        for x, y in zip(blk.samples(), ob.samples()):
            ok = ok and len(x) == len(y) and (x == y).all()
        if not ok:
            np.save("gpurun_out/fuzz_run_block_fail.npy", T)
            print("MISMATCH n=%d sigma=%d lo=%d sym=%d leaves=%d ns=%d seed=%d case=%d" % (n, sigma, lo, sym, leaves, ns, seed, cases), flush=True)
            sys.exit(1)
        blk.free()
        blocks += ctx.timers().get("leaf_run_block", (0, 0, 0))[2] - before
        cases += 1
        if cases % 20 == 0:
            print("%d cases, %d block phases, %.0f s" % (cases, blocks, time.time() - t0), flush=True)
    print("fuzz_run_block ok: %d cases, %d block phases" % (cases, blocks))


if __name__ == "__main__":
    main()
