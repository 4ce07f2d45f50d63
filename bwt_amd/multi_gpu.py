"""One text, several GPUs: leaves sharded over the ranks, merge tree over send/recv.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  Every rank holds the whole text
(a 3.1 Gbp text is 3.1 GB of 288 GB), because suffix comparisons and LF-mapping chains read it at arbitrary
positions.  The recursion tree of bwt::bwt_rec (reference include/bwt.h:854-865: split at (a+b)/2) is cut at depth
log2(world): rank r builds the subtree whose range is the r-th of those world ranges, entirely on its own GPU.  Then,
level by level, the owner of a right sibling sends its block -- L[a:b), the samples and four scalars -- to the owner
of the left sibling, which merges (include/bwt.h:791-847).  The result is bit-identical to the single-GPU recursion
because it is the same tree.  The final block ends on rank 0.

torch is plumbing here: tensors as send/recv buffers, and the process group.  All BWT work is in libbwt_hip.so.
"""
import json
import os
import sys
import time


def split_ranges(a, b, depth):
    """The 2^depth ranges that bwt_rec's midpoint splitting of [a,b) reaches at `depth`."""
    if depth == 0:
        return [(a, b)]
    c = (a + b) // 2
    return split_ranges(a, c, depth - 1) + split_ranges(c, b, depth - 1)


def leaf_threshold(n, total_leaves):
    """A bwt_rec_threshold for which midpoint splitting of [0,n) stops exactly at `total_leaves` equal-depth leaves."""
    depth = total_leaves.bit_length() - 1
    assert 1 << depth == total_leaves, "leaf count must be a power of two"
    sizes = [b - a for a, b in split_ranges(0, n, depth)]
    th = max(sizes)
    if depth > 0:
        parents = [b - a for a, b in split_ranges(0, n, depth - 1)]
        assert min(parents) > th, "n too small for %d leaves" % total_leaves
    return th


def tree_levels(world):
    """[(stride, [(dst, src), ...]), ...]: at each level the owner src of a right sibling sends to dst = src - stride."""
    levels = []
    stride = 1
    while stride < world:
        pairs = [(r, r + stride) for r in range(0, world, 2 * stride) if r + stride < world]
        levels.append((stride, pairs))
        stride *= 2
    return levels


class HipOps:
    """Block operations on one GPU through the C-ABI, and (de)serialisation of a block for transport."""

    def __init__(self, ctx, ns, seg, device_buffers, torch_device=None):
        self.ctx, self.ns, self.seg = ctx, ns, seg
        self.device_buffers = device_buffers  # True: tensors in HBM (nccl); False: host tensors (gloo)
        self.torch_device = torch_device

    def rec(self, a, b, th):
        return self.ctx.rec(a, b, th, self.ns, self.seg)

    def merge(self, l, r):
        return self.ctx.merge(l, r, self.ns, self.seg)

    def header(self, blk):
        return [blk.a, blk.b, blk.s_rank, blk.e_rank, blk.n_samples]

    def pack(self, blk):
        import torch
        if self.device_buffers:
            Lt = torch.empty(blk.b - blk.a, dtype=torch.uint8, device=self.torch_device)
            rt = torch.empty(blk.n_samples, dtype=torch.int64, device=self.torch_device)
            pt = torch.empty(blk.n_samples, dtype=torch.int64, device=self.torch_device)
            torch.cuda.synchronize()
            blk.export_device(Lt.data_ptr(), rt.data_ptr(), pt.data_ptr())
            return Lt, rt, pt
        r, p = blk.samples()
        return torch.from_numpy(blk.L), torch.from_numpy(r), torch.from_numpy(p)

    def empty(self, hdr):
        import torch
        a, b, _, _, ns = hdr
        dev = self.torch_device if self.device_buffers else "cpu"
        return (torch.empty(b - a, dtype=torch.uint8, device=dev), torch.empty(ns, dtype=torch.int64, device=dev),
                torch.empty(ns, dtype=torch.int64, device=dev))

    def unpack(self, hdr, Lt, rt, pt):
        a, b, s_rank, e_rank, ns = hdr
        if self.device_buffers:
            import torch
            torch.cuda.synchronize()
            return self.ctx.import_block_device(a, b, Lt.data_ptr(), s_rank, e_rank, rt.data_ptr(), pt.data_ptr(), ns)
        return self.ctx.upload_block(a, b, Lt.numpy(), s_rank, e_rank, rt.numpy(), pt.numpy())


def send_block(ops, blk, dst, dist, hdr_device):
    import torch
    hdr = torch.tensor(ops.header(blk), dtype=torch.int64, device=hdr_device)
    dist.send(hdr, dst)
    for t in ops.pack(blk):
        dist.send(t, dst)


def recv_block(ops, src, dist, hdr_device):
    import torch
    hdr = torch.empty(5, dtype=torch.int64, device=hdr_device)
    dist.recv(hdr, src)
    h = [int(x) for x in hdr.cpu().tolist()]
    bufs = ops.empty(h)
    for t in bufs:
        dist.recv(t, src)
    return ops.unpack(h, *bufs)


def pmbwt_tree(ops, rank, world, n, leaves_per_rank, dist, hdr_device="cpu", free=lambda b: b.free()):
    """The whole BWT of [0,n) over `world` ranks.  Returns the final block on rank 0 and None elsewhere."""
    depth = world.bit_length() - 1
    assert 1 << depth == world, "world size must be a power of two"
    th = leaf_threshold(n, world * leaves_per_rank)
    a, b = split_ranges(0, n, depth)[rank]
    blk = ops.rec(a, b, th)
    for stride, pairs in tree_levels(world):
        for dst, src in pairs:
            if rank == src:
                send_block(ops, blk, dst, dist, hdr_device)
                free(blk)
                blk = None
            elif rank == dst:
                rblk = recv_block(ops, src, dist, hdr_device)
                blk = ops.merge(blk, rblk)
    return blk


def verify_against_one_gpu(ctx, held, algo, rank, world, th, ns, seg, dist, dev, coll_device=None):
    """Builds the same tree on this rank's GPU alone and compares what the multi-GPU build left here with it: the
    slice of L byte for byte, every sample this rank holds, the special ranks where they are known, and the number of
    samples over all ranks.  Returns a dict for the JSON line (the same on every rank); never raises.  coll_device:
    where the tensors of the two small collectives live (default `dev`; "cpu" under gloo)."""
    import numpy as np
    import torch
    res = {"L": False, "samples": False, "special_ranks": False, "error": None}
    n_held = 0
    blk = None
    try:
        if algo == "native":
            r, p = held.samples()
            first, L, s_rank, e_rank = held.first, held.L, held.s_rank, held.e_rank
        else:
            k = held.ranks.index(rank)
            first = int(sum(held.sizes[:k]))
            L = held.L.cpu().numpy()
            r = held.samp_rank.cpu().numpy().view(np.uint32).astype(np.int64)
            p = held.samp_pos.cpu().numpy().view(np.uint32).astype(np.int64)
            s_rank, e_rank = held.s_rank, held.e_rank
        n_held = len(p)
        blk = ctx.pmbwt(th, ns, seg)
        ctx.sync()
        m = blk.b - blk.a
        Lt = torch.empty(m, dtype=torch.uint8, device=dev)
        rt = torch.empty(blk.n_samples, dtype=torch.int64, device=dev)
        pt = torch.empty(blk.n_samples, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        blk.export_device(Lt.data_ptr(), rt.data_ptr(), pt.data_ptr())
        ctx.sync()
        mine = Lt[first:first + len(L)].cpu().numpy()
        res["L"] = bool(len(mine) == len(L) and np.array_equal(mine, L))
        ra, pa = rt.cpu().numpy(), pt.cpu().numpy()
        at = np.searchsorted(pa, p)
        at[at >= len(pa)] = 0
        res["samples"] = bool(len(pa) > 0 and np.all(np.diff(p) > 0) and np.array_equal(pa[at], p) and np.array_equal(ra[at], r))
        res["special_ranks"] = bool((s_rank < 0 or s_rank == blk.s_rank) and (e_rank < 0 or e_rank == blk.e_rank))
        res["n_samples"] = int(blk.n_samples)
        del Lt, rt, pt
    except Exception as e:  # reported, not raised: the timing line is still worth having
        res["error"] = "rank %d: %r" % (rank, e)
    if blk is not None:
        blk.free()
    # the ranks' findings in one place: all slices equal, all samples accounted for
    flags = torch.tensor([int(res["L"]), int(res["samples"]), int(res["special_ranks"])], dtype=torch.int64, device=coll_device or dev)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    count = torch.tensor([n_held, -1 if res["error"] else 0], dtype=torch.int64, device=coll_device or dev)
    dist.all_reduce(count, op=dist.ReduceOp.SUM)
    out = {"L_slices_equal": bool(flags[0].item()), "samples_equal": bool(flags[1].item()),
           "special_ranks_equal": bool(flags[2].item()), "ranks_with_errors": int(-count[1].item()), "error": res["error"],
           "what": "after the timed steps every rank builds the same tree on its own GPU alone (bwt_hip_pmbwt) and compares its slice of L, its samples and the special ranks it knows"}
    if "n_samples" in res:
        out["samples_equal"] = out["samples_equal"] and int(count[0].item()) == res["n_samples"]
        out["n_samples"] = res["n_samples"]
    return out


def host_to_host_native(ctx, transport, n, th, ns, seg, steps, dist, dev):
    """The N > 1 build from host memory to host memory: on every rank the text goes up from page-locked host memory
    (every GPU holds the whole text: n bytes over each GPU's own link, side by side), the ranks build, and every rank's
    slice of L and its samples come down into page-locked host memory of that rank's process.  Seconds per build of
    the slowest rank, between barriers."""
    import numpy as np
    import torch
    import bwt_amd
    s = ctx.dist_pmbwt(transport, th, ns, seg)
    size, nsamp = s.size, s.n_samples
    s.free()
    # the page-locked buffers come first, and every rank hears whether all ranks got theirs: the builds below are
    # collective, a rank that dropped out would leave the others waiting
    bufs, why = [], None
    try:
        for nbytes in (n, max(1, size), 8 * max(1, nsamp), 8 * max(1, nsamp)):
            bufs.append(bwt_amd.HostBuffer(nbytes))
    except Exception as e:
        why = repr(e)
    got = torch.tensor([0 if why else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(got, op=dist.ReduceOp.MIN)
    if not int(got.item()):
        for b in bufs:
            b.close()
        return {"value": None, "skipped": why or "another rank could not allocate its page-locked buffers"}
    Th, Lh, Sr, Sp = bufs
    ctx.read_text(Th.a)
    sr, sp = Sr.a.view(np.int64), Sp.a.view(np.int64)
    ok = True

    def once():
        ctx.set_text(Th.a)
        s = ctx.dist_pmbwt(transport, th, ns, seg)
        good = s.size == size and s.n_samples == nsamp  # (the same tree every time)
        if good:
            s.read_into(Lh.a, sr, sp)
        s.free()
        return good

    ok = once() and ok
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ok = once() and ok
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    for b in (Th, Lh, Sr, Sp):
        b.close()
    per = float(dt.item()) / steps
    return {"value": round((n - 1) / per / 1e6, 2), "unit": "Mbase/s", "ms_per_step": round(per * 1e3, 3), "steps": steps,
            "what": "on every rank: T in page-locked host memory -> the GPU (n bytes over each GPU's own link, side by side) -> "
                    "the build over all GPUs -> this rank's slice of L and its samples in page-locked host memory; slowest rank, between barriers",
            "sanity": bool(int(flag.item()))}


def bench_main(args, rank, local_rank, world, metric, cpu_baseline):
    """bench.py's N > 1 leg: strong scaling of one text over `world` GPUs.  --leaves is the leaf count of the
    whole tree (the same tree as on one GPU), so every rank builds leaves / world of them."""
    # RCCL prints a version banner on standard output when its first communicator comes up.  This program's
    # standard output is one JSON line: descriptor 1 points at stderr from here on, the line goes out through
    # a copy of the original descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import bwt_amd
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world == 1:  # the one-rank rehearsal (BWT_BENCH_FORCE_DIST) is started without a launcher
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517")):
            os.environ.setdefault(k, v)
    dist.init_process_group(backend="nccl", device_id=dev)
    n = args.n
    ctx = bwt_amd.Context(local_rank)
    ctx.gen_text(n, 42, args.sigma)
    assert args.leaves & (args.leaves - 1) == 0 and (args.leaves % world == 0 or args.leaves == 1), "--leaves: a power of two, a multiple of the GPUs (or 1)"
    one_leaf = args.leaves == 1 and world > 1  # the GPUs share the one leaf by key range (bwt_hip_dist_leaf): native schedule only
    lpr = max(1, args.leaves // world)
    total_leaves = 1 if one_leaf else world * lpr
    depth = total_leaves.bit_length() - 1
    ns = args.ssa_n_samples or max(1000, int(0.4 * n / (1 + max(1, depth)) / 16))
    # "native" (default): every merge runs on all GPUs of its subtree and the result stays spread over the GPUs, the
    # schedule in C++ (bwt_hip_dist_pmbwt) over the library's own RCCL communicator; "dist": the same schedule in
    # Python over torch.distributed (bwt_amd/dist_merge.py; its g values travel as 32-bit words); "tree": right blocks
    # move to the left owner, the result ends on rank 0
    algo = os.environ.get("BWT_MGPU_ALGO", "native")
    if one_leaf and algo != "native":
        raise SystemExit("--leaves 1 on several GPUs needs the native schedule (BWT_MGPU_ALGO=native)")
    transport, fallback = None, None
    if algo == "native":
        # the library's communicator comes up on every rank or the run takes the torch.distributed schedule instead:
        # the ranks agree on that before any build starts
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        try:
            if rank == 0:
                uid.copy_(torch.from_numpy(bwt_amd.Transport.rccl_unique_id()))
        except Exception as e:  # librccl.so not loadable
            fallback = repr(e)
            ok.zero_()
        dist.broadcast(uid, 0)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()):
            try:
                transport = bwt_amd.Transport.rccl(ctx, uid.cpu().numpy(), rank, world)
            except Exception as e:
                fallback = repr(e)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()):
            if transport is not None:
                transport.close()
                transport = None
            fallback = fallback or "another rank could not bring the library's RCCL communicator up"
            print("rank %d: native transport unavailable (%s): torch.distributed schedule instead" % (rank, fallback), file=sys.stderr)
            algo = "dist"
    if algo == "dist":
        from bwt_amd import dist_merge
        dops = dist_merge.HipDistOps(ctx, ns, args.seg, dev)
        groups = dist_merge.make_groups(dist, world)
        dist_merge.warm_groups(dist, groups, rank, world, dev, False)  # communicator set-up, also with --warmup 0

        class _Held:  # what a step leaves behind: this rank's slice (torch frees it with the object)
            def __init__(self, d):
                self.d = d

            def free(self):
                self.d = None

        def step():
            d = dist_merge.pmbwt_dist(dops, rank, world, n, lpr, dist, groups, dev, False)
            ctx.sync()
            return _Held(d)
    elif algo == "native":
        # ncclSend / ncclRecv on a communicator of the library's own; torch.distributed only carried the communicator's
        # id and carries the timing
        th_native = leaf_threshold(n, total_leaves)
        ctx.dist_pmbwt(transport, th_native, min(ns, 1000), args.seg).free()  # connections between the ranks come up at first use

        def step():
            s = ctx.dist_pmbwt(transport, th_native, ns, args.seg)
            ctx.sync()
            return s
    elif algo == "tree":
        ops = HipOps(ctx, ns, args.seg, True, dev)

        def step():
            blk = pmbwt_tree(ops, rank, world, n, lpr, dist, hdr_device=dev)
            ctx.sync()
            return blk
    else:
        raise SystemExit("BWT_MGPU_ALGO=%s: native, dist or tree" % algo)

    for _ in range(args.warmup):
        blk = step()
        if blk is not None:
            blk.free()
    ctx.sync()

    def timed_steps(last):
        """K steps between barriers; the slowest rank's seconds per step, and what the last step left behind."""
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if last is not None:
                last.free()
            last = step()
        ctx.sync()
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()) / args.steps, last

    # the timed region runs with the library's timers off; the same K steps again with an event pair around every
    # kernel give `roofline` and the phase table
    ctx.set_timing(0)
    dt, last = timed_steps(None)
    timers, dt_timers = {}, None
    if not getattr(args, "no_timers", False):
        ctx.set_timing(2)
        ctx.reset_timers()
        dt_timers, last = timed_steps(last)
        timers = ctx.timers()
        ctx.set_timing(0)
    verified = None
    if algo in ("native", "dist") and not os.environ.get("BWT_MGPU_NO_VERIFY"):
        verified = verify_against_one_gpu(ctx, last.d if algo == "dist" else last, algo, rank, world,
                                          leaf_threshold(n, total_leaves), ns, args.seg, dist, dev)
    h2h = None
    if algo == "native" and not getattr(args, "no_extras", False):
        h2h = host_to_host_native(ctx, transport, n, leaf_threshold(n, total_leaves), ns, args.seg, args.steps, dist, dev)
    if rank == 0:
        from bwt_amd.benchutil import phase_table, roofline
        th = leaf_threshold(n, total_leaves)
        out = {
            "metric": metric, "value": round((n - 1) / dt / 1e6, 2), "unit": "Mbase/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic (splitmix64 seed 42, T[i]=1+x%%%d, 0-terminated), generated on every device" % args.sigma,
            "value_definition": "(n-1) / seconds per build (slowest rank, between barriers), the text resident in every GPU's HBM before the timed region, L left sliced over the GPUs, library timers off",
            "ms_per_step_with_kernel_timers": round(dt_timers * 1e3, 3) if dt_timers else None,
            "config": {"workload": ("random DNA n=%d as %d leaf shared by %d MI355X by key range (every GPU reads the whole text and sorts "
                                    "its stretch of first key digits; no merge); L stays sliced over the GPUs, every GPU keeps a stretch of the sample slots"
                                    if one_leaf else
                                    "random DNA n=%d, %d leaves sharded over %d MI355X, every merge spread over the GPUs of its "
                                    "subtree (RCCL send/recv); L stays sliced over the GPUs, every GPU keeps the samples of its own text range"
                                    if algo in ("dist", "native") else
                                    "random DNA n=%d, %d leaves sharded over %d MI355X, merge tree over RCCL send/recv; result on rank 0")
                                   % (n, total_leaves, world),
                       "n": n, "sigma": args.sigma, "leaves": total_leaves, "bwt_rec_threshold": th, "ssa_n_samples": ns,
                       "build_gap_segment_sz": args.seg or 1024, "text_resident_in_hbm": True,
                       "parallelism": ("one leaf cut by key range x%d (C++ driver; RCCL carries the special ranks and the sample slots)" % world) if one_leaf else
                                      "leaf-sharded x%d, %s" % (world, {"dist": "distributed merges (torch.distributed)",
                                                                        "native": "distributed merges (C++ driver, RCCL send/recv)"}.get(algo, "pairwise merge tree"))},
            # rank 0's own kernels (it builds one leaf and does the left half of every merge level)
            "roofline": roofline(timers), "cpu_baseline": None, "schedule": algo, "native_transport_fallback": fallback,
            # outside the timed region: every rank's slice against the same tree built on that rank's GPU alone
            "verified_against_one_gpu": verified,
            "phases_rank0": phase_table(timers, args.steps),
            "host_to_host": h2h,
        }
    if last is not None:
        last.free()
    if transport is not None:
        transport.close()
    ctx.close()
    dist.destroy_process_group()
    if rank == 0:
        # after the process group is gone: the other ranks are not kept waiting in a collective meanwhile
        if args.cpu_n:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n)
        import ctypes
        ctypes.CDLL(None).fflush(None)  # whatever C code still holds in its buffers goes to stderr first
        os.write(json_fd, (json.dumps(out) + "\n").encode())
