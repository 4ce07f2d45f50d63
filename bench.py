#!/usr/bin/env python3
"""Benchmark of the hot path bwt::pmbwt on MI355X.

One step = one whole BWT build (leaf suffix sorts + gap-array merges) of ONE synthetic DNA text.
Metric: Mbase/s = (n-1) / seconds per build.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n N_CHARS] [--sigma S] [--leaves P]

Default workload: random DNA of n = 3.1e9 symbols (GRCh38-sized), the same text for every N, cut into
max(2, N) leaves: one leaf per GPU, and two leaves plus their merge on one GPU, so that every kernel of the
path runs there too.  On N GPUs (launched by torch.distributed.run, one rank per GPU) every rank builds its
leaf and each merge of the tree runs on all GPUs of its subtree (bwt_amd/dist_merge.py): strong scaling of
one text.  `--leaves 8` gives the 8-leaf tree (3.875e8 suffixes per leaf, 7 merges) on any N that divides 8;
`--n 1000000000 --leaves 2` is the one-GPU configuration "leaf sort + single gap-array merge"; `--gpus N --leaves 1`
builds the text as ONE leaf that the N GPUs share by key range (no merge: the fastest way to N GPUs for a text that
fits them).

`value` is measured with the text already in HBM and the result left in HBM, the library's timers off; the K
steps are then repeated with an event pair around every kernel for `roofline` and `phases`.  The same line also
carries `host_to_host`, over the same number of steps: text in (page-locked) host memory before, L and the
samples in host memory after, the transfers inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "BWT build Mbase/s end-to-end, 3.1 Gbp DNA, at 1/2/4/8 MI355X"
DEFAULT_N = 3100000000
VALUE_DEFINITION = ("value: (n-1) / seconds per build with the text resident in HBM before the timed region and L and the "
                    "samples left in HBM, library timers off; host_to_host (same line, same number of steps): the text in host "
                    "memory before and L and the samples in host memory after, every transfer inside the timed region "
                    "(bwt_hip_pmbwt_host, what bwt::pmbwt calls)")


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


def default_samples(n, leaves):
    """ssa_n_samples as the reference's set_defaults derives it at mem_budget 4 for this recursion depth
    (include/opts.h:286-289)."""
    return max(1000, int(0.4 * n / (1 + max(1, leaves.bit_length() - 1)) / 16))


def workload_text(n, sigma, leaves, world):
    merges = leaves - 1
    if world == 1:
        return "random DNA n=%d on one MI355X: %d leaf suffix sorts + %d gap-array merge(s)" % (n, leaves, merges)
    return ("random DNA n=%d, %d leaves sharded over %d MI355X, every merge spread over the GPUs of its subtree "
            "(RCCL point-to-point + all-to-all)" % (n, leaves, world))


def timed_builds(ctx, th, ns, seg, steps):
    t0 = time.perf_counter()
    for _ in range(steps):
        blk = ctx.pmbwt(th, ns, seg)
        ctx.sync()
        blk.free()
    ctx.sync()
    return (time.perf_counter() - t0) / steps


def host_to_host(ctx, n, th, ns, seg, steps):
    """(n-1)/time of: upload T from host memory, build, download L and the samples.  Page-locked buffers on
    the host side; every byte crosses the link inside the timed region."""
    import numpy as np
    import bwt_amd
    Th, Lh = bwt_amd.HostBuffer(n), bwt_amd.HostBuffer(n)
    ctx.read_text(Th.a)  # the text the resident runs used, now on the host
    n_samp = min(n, ns)
    Sr, Sp = bwt_amd.HostBuffer(8 * n_samp), bwt_amd.HostBuffer(8 * n_samp)
    sr, sp = Sr.a.view(np.int64), Sp.a.view(np.int64)
    parts = {"upload_build_download_L_ms": 0.0, "download_samples_ms": 0.0}

    def once(record):
        t0 = time.perf_counter()
        blk = ctx.pmbwt_host(Th.a, th, ns, seg, Lh.a)  # chunked upload under the first leaf's sort, build, L back
        t1 = time.perf_counter()
        k = blk.read_samples(sr, sp)
        t2 = time.perf_counter()
        blk.free()
        if record:
            parts["upload_build_download_L_ms"] += (t1 - t0) * 1e3
            parts["download_samples_ms"] += (t2 - t1) * 1e3
        return t2 - t0, k

    once(False)
    total = 0.0
    for _ in range(steps):
        dt, k = once(True)
        total += dt
    dt = total / steps
    ok = bool(Lh.a[0] != 0 or n < 3) and int(sp[k - 1]) == n - 1
    out = {"value": round((n - 1) / dt / 1e6, 2), "unit": "Mbase/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
           "what": "T in page-locked host memory -> bwt_hip_pmbwt_host (text up and L down as two bits per symbol, upload overlapped with the first leaf) -> L and %d samples in page-locked host memory" % k,
           "sanity": ok}
    out.update({k_: round(v / steps, 3) for k_, v in parts.items()})
    for b in (Th, Lh, Sr, Sp):
        b.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--text_len", "--n", dest="n", type=int, default=DEFAULT_N, help="text length (default 3.1e9)")
    ap.add_argument("--sigma", type=int, default=4)
    ap.add_argument("--leaves", type=int, default=None,
                    help="leaves of the whole tree (a power of two, at least one per GPU; default max(2, GPUs))")
    ap.add_argument("--ssa_n_samples", type=int, default=0)
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--cpu_n", type=int, default=3 * 10 ** 7, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--no-timers", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-to-host leg and the other tree shapes")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus),
                  file=sys.stderr)
            sys.exit(2)
    if args.leaves is None:
        args.leaves = max(2, world)
    # (one leaf on several GPUs: the GPUs share the leaf by key range, no merge at all)
    if args.leaves < 1 or args.leaves & (args.leaves - 1) or (args.leaves % world and args.leaves != 1):
        print("--leaves %d: must be a power of two and a multiple of the %d GPUs (or 1)" % (args.leaves, world), file=sys.stderr)
        sys.exit(2)

    import bwt_amd
    if world > 1 or os.environ.get("BWT_BENCH_FORCE_DIST"):  # the env switch rehearses the N > 1 leg on one GPU
        from bwt_amd import multi_gpu
        return multi_gpu.bench_main(args, rank, local_rank, world, METRIC, cpu_baseline)

    n = args.n
    ctx = bwt_amd.Context(0)
    ctx.gen_text(n, 42, args.sigma)
    ns = args.ssa_n_samples or default_samples(n, args.leaves)
    th = -(-n // args.leaves)
    for _ in range(args.warmup):
        ctx.pmbwt(th, ns, args.seg).free()
    ctx.sync()
    # The timed region runs without the library's timers (no events around the kernels); the same K steps are then
    # repeated with an event pair around every kernel launch, on the library's own stream, for `roofline` and `phases`.
    ctx.set_timing(0)
    dt = timed_builds(ctx, th, ns, args.seg, args.steps)
    timers, dt_timers = {}, None
    if not args.no_timers:
        ctx.set_timing(2)
        ctx.reset_timers()
        dt_timers = timed_builds(ctx, th, ns, args.seg, args.steps)
        timers = ctx.timers()
        ctx.set_timing(0)
    mem = ctx.mem_stats()

    from bwt_amd.benchutil import phase_table, roofline
    roof = roofline(timers, ROOT, {"n": n, "leaves": args.leaves})
    phases = phase_table(timers, args.steps)
    out = {
        "metric": METRIC, "value": round((n - 1) / dt / 1e6, 2), "unit": "Mbase/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic (splitmix64 seed 42, T[i]=1+x%%%d, 0-terminated), generated on the device" % args.sigma,
        "value_definition": VALUE_DEFINITION,
        "ms_per_step_with_kernel_timers": round(dt_timers * 1e3, 3) if dt_timers else None,
        "config": {"workload": workload_text(n, args.sigma, args.leaves, 1),
                   "n": n, "sigma": args.sigma, "leaves": args.leaves, "bwt_rec_threshold": th, "ssa_n_samples": ns,
                   "build_gap_segment_sz": args.seg or 1024, "text_resident_in_hbm": True},
        "roofline": roof,
        "phases": phases,
        "device_mem_peak_gb": round(mem["peak"] / 1e9, 2),
    }
    if not args.no_extras:
        out["host_to_host"] = host_to_host(ctx, n, th, ns, args.seg, args.steps)
        # the same text as other trees (one step each after one untimed build)
        other = {}
        for leaves in (8, 1):
            if leaves == args.leaves:
                continue
            th2, ns2 = -(-n // leaves), default_samples(n, leaves)
            ctx.pmbwt(th2, ns2, args.seg).free()
            d2 = timed_builds(ctx, th2, ns2, args.seg, 1)
            other["leaves=%d" % leaves] = {"ms_per_step": round(d2 * 1e3, 3), "Mbase_per_s": round((n - 1) / d2 / 1e6, 2)}
        out["other_trees"] = other
    ctx.close()
    if args.cpu_n:
        out["cpu_baseline"] = cpu_baseline(args.cpu_n)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
