#!/usr/bin/env python3
"""bwt_hip_pmbwt_multi with all ranks on GPU 0, for a kernel profile of the multi-GPU schedule's own kernels (wire packing,
grouping, multi-sender prefix sums, sample steps) at full size: the ranks' kernels share the one GPU, so the total is the
sum of all ranks' work and says nothing about a real multi-GPU build's time.
    python tools/multi_on_one_gpu.py [--text_len N] [--ranks R]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text_len", type=int, default=3100000000)
    ap.add_argument("--ranks", type=int, default=8)
    args = ap.parse_args()
    n = args.text_len
    ctx = bwt_amd.Context(0)
    ctx.gen_text(n, 42, 4)
    T = ctx.get_text()
    ctx.close()
    th, ns = -(-n // args.ranks), max(1000, int(0.4 * n / (1 + max(1, args.ranks.bit_length() - 1)) / 16))
    t0 = time.perf_counter()
    L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * args.ranks, T, th, ns)
    print("n=%d over %d ranks on one GPU: %.1f ms in all (text uploads and downloads included), s_rank %d, %d samples"
          % (n, used, (time.perf_counter() - t0) * 1e3, s_rank, len(r)))


if __name__ == "__main__":
    main()
