"""World-size-2 (or more) run over gloo of the one-leaf schedule with numpy standing in for the GPU: every rank takes
its key range of the oracle's suffix array, fills the sample slots of the suffixes it owns, the slots are exchanged,
and rank 0 gathers slices and samples for the test to compare with the oracle's leaf."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bwt_amd import dist_leaf  # noqa: E402
from oracle import Oracle, gen_text  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--text_len", type=int, default=20000)
    ap.add_argument("--ns", type=int, default=300)
    ap.add_argument("--sigma", type=int, default=4)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, ns = args.text_len, min(args.ns, args.text_len)
    T = gen_text(n, 42, args.sigma)
    sa = Oracle.sa_range(T, 0, n)
    # the first key digit of a suffix: its first two symbols (any prefix-monotone digit cuts the order into stretches)
    first = T[sa].astype(np.int64) * 256 + np.where(sa + 1 < n, T[np.minimum(sa + 1, n - 1)], 0)
    digit_counts = np.bincount(first, minlength=256 * 256)
    cuts, bounds = dist_leaf.key_range_cuts(digit_counts, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    assert ((first[lo:hi] >= cuts[rank]) & (first[lo:hi] < cuts[rank + 1])).all()
    L_slice = T[sa[lo:hi] - 1]  # index -1 wraps to the terminator, as the reference's sa_to_bwt does
    slots = np.full(ns, dist_leaf.EMPTY, dtype=np.uint64)
    pos = sa[lo:hi]
    rel = pos.astype(np.float64)
    x = np.floor((pos - 1) * (ns - 1) / float(n - 1)).astype(np.int64)  # include/bwt.h:339-340
    y = np.floor(pos * (ns - 1) / float(n - 1)).astype(np.int64)
    take = (x != y) | (pos == 0)
    slots[y[take]] = (pos[take].astype(np.uint64) << np.uint64(32)) | (np.arange(lo, hi)[take]).astype(np.uint64)
    mine = dist_leaf.share_exchange(rank, world, slots, dist)
    assert (mine != dist_leaf.EMPTY).all()
    # gather on rank 0
    sizes = [bounds[j + 1] - bounds[j] for j in range(world)]
    if rank == 0:
        Ls, Ss = [L_slice], [mine]
        for j in range(1, world):
            t = torch.empty(sizes[j], dtype=torch.uint8)
            dist.recv(t, j)
            Ls.append(t.numpy())
            s_lo, s_hi = dist_leaf.slot_range(ns, j, world)
            t = torch.empty(s_hi - s_lo, dtype=torch.int64)
            dist.recv(t, j)
            Ss.append(t.numpy().view(np.uint64))
        S = np.concatenate(Ss)
        np.savez(args.out, L=np.concatenate(Ls), samp_pos=(S >> np.uint64(32)).astype(np.int64), samp_rank=(S & np.uint64(0xFFFFFFFF)).astype(np.int64))
    else:
        dist.send(torch.from_numpy(np.ascontiguousarray(L_slice)), 0)
        dist.send(torch.from_numpy(mine.view(np.int64).copy()), 0)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
