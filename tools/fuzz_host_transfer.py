"""Differential fuzz of bwt_hip_pmbwt_host's packed transfers against set_text + pmbwt on a second context: random
lengths, alphabets (shifted, short, with symbols the sample misses), chunk sizes, leaf sizes and sample counts.
Usage: python tools/fuzz_host_transfer.py [iterations] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    verbose = os.environ.get("FUZZ_VERBOSE", "0") != "0"
    rng = np.random.default_rng(seed)
    ref = bwt_amd.Context(0)
    restarts = packed = 0
    for it in range(iters):
        n = int(rng.choice([rng.integers(2, 200), rng.integers(200, 70000), rng.integers(70000, 4000000)]))
        sigma = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 20]))
        if sigma == 1 and n > 50000:  # a run of one symbol makes every chain head's comparisons as long as the run: minutes
            sigma = 2
        shift = int(rng.integers(0, 256 - sigma))
        T = np.empty(n, dtype=np.uint8)
        T[:-1] = 1 + shift + rng.integers(0, sigma, n - 1, dtype=np.uint8)
        T[-1] = 0
        foreign = sigma <= 4 and n > 10 and rng.random() < 0.25
        if foreign:  # a symbol or two the sample is unlikely to see (it sees both ends and a stride)
            for _ in range(int(rng.integers(1, 3))):
                T[int(rng.integers(0, n - 1))] = int(rng.integers(1, 256))
        os.environ["BWT_HIP_HOST_PACK"] = "2" if rng.random() < 0.8 else "0"
        os.environ["BWT_HIP_POISON_TEXT"] = "1"  # a kernel that reads ahead of the upload must show in the result
        os.environ["BWT_HIP_XFER_CHUNK"] = str(int(rng.integers(64, 1 << 14)) * 64)
        c = bwt_amd.Context(0)
        th = int(rng.integers(max(1, n // 9), n + 1)) if n >= 16 else n
        ns = int(rng.integers(2, max(3, n // 2)))
        if verbose:
            print("it=%d n=%d sigma=%d shift=%d th=%d ns=%d chunk=%s foreign=%s" % (it, n, sigma, shift, th, ns, os.environ["BWT_HIP_XFER_CHUNK"], foreign), flush=True)
        Lh = np.full(n, 0xEE, dtype=np.uint8)
        blk = c.pmbwt_host(T, th, ns, 0, Lh)
        ref.set_text(T)
        want = ref.pmbwt(th, ns)
        ok = bool((want.L == Lh).all()) and bool((blk.L == Lh).all()) and (want.s_rank, want.e_rank) == (blk.s_rank, blk.e_rank)
        ok = ok and all(bool((x == y).all()) for x, y in zip(blk.samples(), want.samples())) and bool((c.get_text() == T).all())
        if not ok:
            np.save("gpurun_out/fuzz_host_transfer_fail_T.npy", T)
            print("MISMATCH it=%d n=%d sigma=%d shift=%d th=%d ns=%d chunk=%s foreign=%s" % (it, n, sigma, shift, th, ns, os.environ["BWT_HIP_XFER_CHUNK"], foreign), flush=True)
            sys.exit(1)
        span = int(T[:-1].max()) - int(T[:-1].min()) if n > 1 else 0
        packed += span <= 3
        restarts += int(foreign)
        blk.free()
        want.free()
        c.close()
        if it % 50 == 49:
            print("%d builds ok (%d with at most four symbol values, %d with planted symbols)" % (it + 1, packed, restarts), flush=True)
    print("OK: %d builds" % iters)


if __name__ == "__main__":
    main()
