#!/usr/bin/env python3
"""Benchmark of the hot path bwt::pmbwt on MI355X.

One step = one whole BWT build (leaf suffix sorts + gap-array merges) of a synthetic DNA text that is
already resident in HBM; the result L stays in HBM.  Metric: Mbase/s = (n-1) / seconds per build.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n N_CHARS] [--sigma S] [--leaves P]

N = 1 runs the configuration the headline metric is quoted on for one GPU: random DNA, n = 1e9, two leaves
of 5e8 and one gap-array merge.  For N > 1 (launched by torch.distributed.run, one rank per GPU) the ranks share
ONE text of 3.1e9 symbols: each builds the leaves of its part, and every merge of the tree runs on all GPUs of its
subtree, exchanging chain heads, gap counts and block slices over RCCL (bwt_amd/multi_gpu.py, bwt_amd/dist_merge.py);
that is strong scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: the reference prints its own timing line with printf, and
    this program's standard output is one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)  # the C library's buffer, before descriptor 1 comes back
        os.dup2(self.saved, 1)
        os.close(self.saved)


def cpu_baseline(budget_n):
    """The reference itself (oracle/_ref, serial build) or, without it, the C oracle, on a bounded sample."""
    import numpy as np  # noqa: F401
    from oracle import Oracle, Reference, gen_text
    n = budget_n
    T = gen_text(n, 42, 4)
    t0 = time.perf_counter()
    if Reference.available():
        with stdout_to_stderr():
            blk = Reference.pmbwt(T, alpha_min=1, alpha_max=4)
        kind = "reference"
        what = "taura/bwt pmbwt, serial build (parallel_model_none: TBB headers absent), -O3 -DNDEBUG"
    else:
        blk = Oracle.pmbwt(T, Oracle.opt(n, alpha_min=1, alpha_max=4))
        kind = "port"
        what = "C restatement (oracle/pmbwt_oracle.c), serial"
    dt = time.perf_counter() - t0
    del blk
    return {"value": round((n - 1) / dt / 1e6, 4), "unit": "Mbase/s", "cores": 1, "kind": kind,
            "sample": "n=%d of the same splitmix64 DNA stream (seed 42), bwt_opt defaults (mem_budget 4 -> 16 leaves); %s; %.1f s"
                      % (n, what, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--text_len", "--n", dest="n", type=int, default=None, help="text length (default 1e9 on one GPU, 3.1e9 on several)")
    ap.add_argument("--sigma", type=int, default=4)
    ap.add_argument("--leaves", type=int, default=None, help="leaves (power of two; default 2 on one GPU, one per GPU on several)")
    ap.add_argument("--ssa_n_samples", type=int, default=0)
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--cpu_n", type=int, default=3 * 10 ** 7, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--no-timers", action="store_true")
    args = ap.parse_args()
    args.n_given, args.leaves_given = args.n is not None, args.leaves is not None
    if args.n is None:
        args.n = 10 ** 9
    if args.leaves is None:
        args.leaves = 2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus),
                  file=sys.stderr)
            sys.exit(2)

    import bwt_amd
    if world > 1 or os.environ.get("BWT_BENCH_FORCE_DIST"):  # the env switch rehearses the N > 1 leg on one GPU
        from bwt_amd import multi_gpu
        return multi_gpu.bench_main(args, rank, local_rank, world)

    n = args.n
    ctx = bwt_amd.Context(0)
    ctx.gen_text(n, 42, args.sigma)
    # the reference derives ssa_n_samples from mem_budget (include/opts.h:286-289); at mem_budget 4 and
    # two levels of recursion that is n/5/... ; use its default formula's value for 2 leaves: 0.4n/(1+1)/16
    ns = args.ssa_n_samples or max(1000, int(0.4 * n / (1 + max(1, args.leaves.bit_length() - 1)) / 16))
    th = -(-n // args.leaves)
    for _ in range(args.warmup):
        ctx.pmbwt(th, ns, args.seg).free()
    ctx.sync()
    ctx.set_timing(0 if args.no_timers else 2)
    ctx.reset_timers()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        blk = ctx.pmbwt(th, ns, args.seg)
        ctx.sync()
        blk.free()
    ctx.sync()
    dt = (time.perf_counter() - t0) / args.steps
    timers = ctx.timers()
    ctx.set_timing(0)
    mem = ctx.mem_stats()

    from bwt_amd.benchutil import phase_table, roofline
    roof = roofline(timers, ROOT, {"n": n, "leaves": args.leaves})
    phases = phase_table(timers, args.steps)
    out = {
        "metric": "BWT build Mbase/s end-to-end", "value": round((n - 1) / dt / 1e6, 2), "unit": "Mbase/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic (splitmix64 seed 42, T[i]=1+x%%%d, 0-terminated), generated on the device" % args.sigma,
        "config": {"workload": "random DNA n=%d on one MI355X: %d leaf suffix sorts + %d gap-array merge(s)"
                               % (n, args.leaves, args.leaves - 1),
                   "n": n, "sigma": args.sigma, "leaves": args.leaves, "bwt_rec_threshold": th, "ssa_n_samples": ns,
                   "build_gap_segment_sz": args.seg or 1024, "text_resident_in_hbm": True},
        "roofline": roof,
        "phases": phases,
        "device_mem_peak_gb": round(mem["peak"] / 1e9, 2),
    }
    ctx.close()
    if args.cpu_n:
        out["cpu_baseline"] = cpu_baseline(args.cpu_n)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
