"""One leaf shared by several ranks, cut by key range: the arithmetic of bwt_hip_dist_leaf / bwt_hip_dist_pmbwt's
one-leaf case restated with numpy, for the CPU tests of the schedule (the GPU path is bwt_amd/csrc: radix_sort.hip
leaf_sort_text_lean, dist_driver.hip dist_leaf).

Every rank counts the first key digit of every suffix of the leaf, cuts the digit values into `parts` stretches of
about equal counts -- the same cut on every rank, nothing exchanged -- and builds the stretch of the suffix order
whose keys begin with its digits: ranks [bounds[k], bounds[k+1]) of the leaf.  Sample slots (position-ordered, one per
sampled position) are filled by whoever owns the sampled suffix; rank j then collects the j-th stretch of slots from
every rank and keeps the filled ones."""
import numpy as np

EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


def key_range_cuts(digit_counts, parts):
    """First digit of every part (parts + 1 entries): part j begins at the first digit at which j / parts of the
    suffixes lie below (leaf_sort_text_lean's `cut`)."""
    cum = np.concatenate([[0], np.cumsum(np.asarray(digit_counts, dtype=np.int64))])
    n, nd = int(cum[-1]), len(digit_counts)
    cuts = [0]
    for j in range(1, parts):
        want = n * j // parts
        d = 0
        while d < nd and cum[d] < want:
            d += 1
        cuts.append(d)
    cuts.append(nd)
    return cuts, [int(cum[d]) for d in cuts]


def slot_range(ns, j, world):
    return ns * j // world, ns * (j + 1) // world


def merge_slots(arrays):
    """Slot-wise: the filled one of the ranks' slots (empty = all ones: the minimum)."""
    return np.minimum.reduce(np.stack(arrays))


def share_exchange(rank, world, slots, dist):
    """Rank j gets the j-th stretch of every rank's sample slots and keeps the filled ones (dist_driver.hip dist_leaf:
    one all-to-all)."""
    import torch
    ns = len(slots)
    lo, hi = slot_range(ns, rank, world)
    send = [torch.from_numpy(slots[slice(*slot_range(ns, j, world))].view(np.int64).copy()) for j in range(world)]
    recv = [torch.empty(hi - lo, dtype=torch.int64) for _ in range(world)]
    reqs = []
    for j in range(world):
        if j == rank:
            recv[j].copy_(send[j])
        else:
            reqs.append(dist.isend(send[j], j))
            reqs.append(dist.irecv(recv[j], j))
    for r in reqs:
        r.wait()
    return merge_slots([t.numpy().view(np.uint64) for t in recv])
