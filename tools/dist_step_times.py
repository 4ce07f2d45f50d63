#!/usr/bin/env python3
"""What ONE GPU does in the top-level merge of an n-symbol text spread over G GPUs (bwt_amd/dist_merge.py), timed
step by step on a single GPU: the left half is built locally, then rank k's share is run with synthetic exchange
partners (the g values and right symbols it would receive are produced locally).  For sizing and for catching
problems that only show at full size; the exchange itself is not part of it.

    python tools/dist_step_times.py --text_len 3100000000 --parts 8 --k 3
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text_len", type=int, default=3100000000)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--left_leaves", type=int, default=4)
    ap.add_argument("--seg", type=int, default=0, help="build_gap_segment_sz (0: the default, 1024)")
    args = ap.parse_args()
    import torch
    import bwt_amd
    from bwt_amd import dist_merge
    dev = torch.device("cuda", 0)
    n, G, k = args.text_len, args.parts, args.k
    ctx = bwt_amd.Context(0)
    ctx.gen_text(n, 42, 4)
    a, c, b = 0, n // 2, n
    m, r = c - a, b - c
    ns = max(1000, n // 160)
    ops = dist_merge.HipDistOps(ctx, ns, args.seg, dev)

    def timed(name, f):
        torch.cuda.synchronize()
        ctx.sync()
        t0 = time.perf_counter()
        out = f()
        torch.cuda.synchronize()
        ctx.sync()
        print("  %-44s %8.2f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True)
        return out

    left = timed("left half [0,n/2) built locally (%d leaves)" % args.left_leaves,
                 lambda: ops.local_block(a, c, -(-m // args.left_leaves), 0))
    lb = timed("left block from its L (copy)", lambda: ops.left_block(a, c, left.L, left.s_rank, left.e_rank))
    p_lo, p_hi = c + k * r // G, c + (k + 1) * r // G
    g = timed("chains of %d right positions (incl. rank structure of %d)" % (p_hi - p_lo, m), lambda: ops.chains(lb, p_lo, p_hi))
    g2 = timed("chains again (rank structure cached)", lambda: ops.chains(lb, p_lo, p_hi))
    assert bool((g == g2).all())
    del g2
    parts, counts = timed("partition of g into %d parts" % G, lambda: ops.partition(g, m, G))
    print("     part sizes:", counts)
    lo = [int(x) for x in bwt_amd.dist_bounds(m, G)]
    i_lo, i_hi = min(lo[k], m), min(lo[k + 1], m)
    # stand-in for what the all-to-alls deliver: G times this GPU's own values of part k, that many right symbols
    off = sum(counts[:k])
    mine = parts[off:off + counts[k]]
    g_in = mine.repeat(G)
    rL = torch.randint(1, 5, (g_in.numel(),), dtype=torch.uint8, device=dev)
    out, S = timed("prefix sums + merge of the slice (mL=%d, rn=%d)" % (i_hi - i_lo, g_in.numel()),
                   lambda: ops.merge_slice(left.L[i_lo:i_hi], rL, g_in))
    # this GPU's own samples (an eighth of the left block's, as a left holder): requests, answers, thinning
    nl = left.n_samples
    share = slice(k * nl // G, (k + 1) * nl // G)
    rank, pos = left.samp_rank[share].contiguous(), left.samp_pos[share].contiguous()

    def sample_steps():
        queries, perm, ask = ops.samples_route(True, rank, pos, a, c, m, r, G)
        mine_q = queries[sum(ask[:k]):sum(ask[:k + 1])]  # the requests that stay on this GPU, answered G times over
        q_in = mine_q.repeat(G)
        ans = ops.samples_answer(q_in, int(q_in.numel()), 0, S, i_lo, 0, a, g)
        return ops.samples_place(True, rank, pos, queries, perm, a, c, share.start, 2 * nl, ns)
    timed("samples: route + answer + place (%d of %d + %d)" % (rank.numel(), nl, nl), sample_steps)
    mem = ctx.mem_stats()
    print("  library device memory peak %.1f GB; torch %.1f GB" % (mem["peak"] / 1e9, torch.cuda.max_memory_allocated() / 1e9))


if __name__ == "__main__":
    main()
