"""One text, several GPUs, every merge spread over all GPUs of its subtree.

bwt_amd.multi_gpu.pmbwt_tree leaves the upper levels of the merge tree to fewer and fewer GPUs.  Here a block of
the tree that spans q GPUs stays distributed over them as q slices of its L (rank ranges), with the samples and the
special ranks on the first of them, and a merge of two such blocks (reference include/bwt.h:791-847) runs on all 2q:

  1. the left block's L is sent to every GPU of the group (point-to-point, all links at once); each builds the left
     rank structure; 2. the right block's text positions are split evenly and every GPU runs the LF chains of its share
     (g_j = number of left suffixes below T[j:] depends on the left block and the text only, and the text is on every
     GPU); 3. the merge is cut by left rank into 2q parts (bwt_hip_dist_bounds): an all-to-all moves every g value to
     the GPU that owns its part, a second one the right symbols each part interleaves; 4. every GPU builds the prefix
     sums of its part and writes its slice of the merged L; 5. the samples of both blocks go to every GPU, each
     computes the new ranks it knows (left samples through its prefix sums, right samples through its chains) and a sum
     to the group's first GPU completes the merged sample set.

Exchanged per merge and GPU: the left L in (m bytes, or m/4 when the text's symbols span at most four values and
L crosses the links as two bits per symbol), about 4r/G bytes of g values and r/G (or r/4G) bytes of right symbols
through the all-to-alls, 8 bytes per sample.  The result is the same block the
single-GPU recursion builds: L bit for bit (the slices concatenated), and the same samples, because the same tree is
merged with the same rules.

torch is plumbing: tensors are buffers, torch.distributed is the transport (backend "nccl" is RCCL over xGMI; gloo with
host staging in the tests).  All BWT work goes through the C-ABI (include/bwt_hip.h, bwt_hip_dist_*).
"""
import numpy as np

from . import dist_bounds


class DistBlock:
    """Block [a,b) spread over `ranks` (global ranks, ascending): slice i of L, `sizes[i]` symbols, is on ranks[i].
    samples (rank, pos: int64 tensors) and s_rank / e_rank (block-relative + a) are valid on ranks[0] only."""

    def __init__(self, a, b, ranks, sizes, L, s_rank=-1, e_rank=-1, samp_rank=None, samp_pos=None):
        self.a, self.b, self.ranks, self.sizes, self.L = a, b, list(ranks), list(sizes), L
        self.s_rank, self.e_rank, self.samp_rank, self.samp_pos = s_rank, e_rank, samp_rank, samp_pos

    @property
    def n_samples(self):
        return 0 if self.samp_rank is None else int(self.samp_rank.numel())


class Comm:
    """The few collectives the merge needs, over one process group.  With host_staging (gloo) device tensors travel
    through host copies."""

    def __init__(self, dist, group, ranks, me, device, host_staging):
        self.dist, self.group, self.ranks, self.me = dist, group, list(ranks), me
        self.device, self.host_staging = device, host_staging
        self.k = self.ranks.index(me)

    def _wire(self, t):
        return t.cpu() if self.host_staging else t

    def _home(self, t):
        return t.to(self.device) if self.host_staging else t

    def all_gather_i64(self, values):
        import torch
        wire_dev = "cpu" if self.host_staging else self.device
        mine = torch.tensor(list(values), dtype=torch.int64, device=wire_dev)
        out = [torch.empty_like(mine) for _ in self.ranks]
        self.dist.all_gather(out, mine, group=self.group)
        return [[int(x) for x in t.cpu().tolist()] for t in out]

    def all_to_all(self, send, send_splits, recv_splits, pad=0):
        """send: 1-D tensor cut into len(ranks) consecutive pieces of send_splits; returns the received pieces, by source
        (followed by `pad` unused elements)."""
        import torch
        src = self._wire(send).contiguous()
        total = int(sum(recv_splits))
        out = torch.empty(total + pad, dtype=src.dtype, device=src.device)
        self.dist.all_to_all_single(out[:total], src, [int(x) for x in recv_splits], [int(x) for x in send_splits], group=self.group)
        return self._home(out)

    def spread(self, pieces, out, offsets, sizes=None):
        """pieces[j] (or None) is the tensor rank ranks[j] owns; afterwards out[offsets[j]:offsets[j+1]] holds it on every rank
        (with `sizes`: out[offsets[j]:offsets[j]+sizes[j]], the pieces need not touch)."""
        import torch
        ops, stage = [], []
        for j, owner in enumerate(self.ranks):
            lo = offsets[j]
            hi = offsets[j + 1] if sizes is None else lo + sizes[j]
            if hi == lo:
                continue
            if owner == self.me:
                out[lo:hi].copy_(pieces[j])
                w = self._wire(pieces[j]).contiguous()
                for peer in self.ranks:
                    if peer != self.me:
                        ops.append(self.dist.P2POp(self.dist.isend, w, peer, group=self.group))
            else:
                if self.host_staging:
                    buf = torch.empty(hi - lo, dtype=out.dtype)
                    stage.append((lo, hi, buf))
                else:
                    buf = out[lo:hi]
                ops.append(self.dist.P2POp(self.dist.irecv, buf, owner, group=self.group))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for lo, hi, buf in stage:
            out[lo:hi].copy_(buf)

    def broadcast(self, t, src_rank):
        w = self._wire(t)
        self.dist.broadcast(w, src=src_rank, group=self.group)
        if self.host_staging:
            t.copy_(w)
        return t

    def sum_to(self, t, dst_rank):
        w = self._wire(t)
        self.dist.reduce(w, dst=dst_rank, op=self.dist.ReduceOp.SUM, group=self.group)
        if self.host_staging and self.me == dst_rank:
            t.copy_(w)
        return t


class HipDistOps:
    """The per-GPU steps through the C-ABI; tensors are torch tensors on this process's GPU."""

    def __init__(self, ctx, ns, seg, torch_device):
        import torch
        self.torch, self.ctx, self.ns, self.seg, self.device = torch, ctx, ns, seg, torch_device

    def empty(self, n, dtype):
        return self.torch.empty(int(n), dtype=dtype, device=self.device)

    def zeros(self, n, dtype):
        return self.torch.zeros(int(n), dtype=dtype, device=self.device)

    def _ready(self):
        self.torch.cuda.synchronize(self.device)  # the library queues on its own stream

    def local_block(self, a, b, th, me):
        blk = self.ctx.rec(a, b, th, self.ns, self.seg)
        L = self.empty(b - a, self.torch.uint8)
        r = self.empty(blk.n_samples, self.torch.int64)
        p = self.empty(blk.n_samples, self.torch.int64)
        self._ready()
        blk.export_device(L.data_ptr(), r.data_ptr(), p.data_ptr())
        self.ctx.sync()
        d = DistBlock(a, b, [me], [b - a], L, blk.s_rank, blk.e_rank, r, p)
        blk.free()
        return d

    def left_block(self, a, c, L, s_rank, e_rank):
        self._ready()
        return self.ctx.import_block_device(a, c, L.data_ptr(), s_rank, e_rank, 0, 0, 0)

    def chains(self, lb, p_lo, p_hi):
        g = self.empty(p_hi - p_lo, self.torch.int32)  # holds unsigned 32-bit values
        if p_hi > p_lo:
            self._ready()
            self.ctx.dist_chains(lb, p_lo, p_hi, self.seg, g.data_ptr())
        return g

    def partition(self, g, m, parts):
        out = self.empty(g.numel(), self.torch.int32)
        self._ready()
        counts = self.ctx.dist_partition(g.data_ptr(), g.numel(), m, parts, out.data_ptr())
        return out, [int(x) for x in counts]

    def wire_packable(self):
        return self.ctx.dist_wire_packable()

    def wire_pack(self, t):
        n = int(t.numel())
        out = self.empty(wire_bytes(n), self.torch.uint8)
        if n:
            self._ready()
            self.ctx.dist_wire_pack(t.data_ptr(), n, out.data_ptr())
        return out

    def wire_unpack(self, packed, count, dst):
        """dst[0:count) from the packed stream at the start of `packed`, which is followed by WIRE_PAD readable bytes."""
        if count:
            self._ready()
            self.ctx.dist_wire_unpack(packed.data_ptr(), count, dst.data_ptr())

    def merge_slice(self, lL, rL, g_rel):
        mL, rn = int(lL.numel()), int(rL.numel())
        out = self.empty(mL + rn, self.torch.uint8)
        S = self.empty(mL + 1, self.torch.int32)
        self._ready()
        self.ctx.dist_merge_slice(lL.data_ptr(), mL, rL.data_ptr(), rn, g_rel.data_ptr(), out.data_ptr(), S.data_ptr())
        return out, S

    def resample(self, l_rank, l_pos, r_rank, r_pos, a, c, ns_out, S, i_lo, i_hi, right_before, g, p_lo, p_hi):
        nl, nr = int(l_rank.numel()), int(r_rank.numel())
        kept = self.ctx.dist_resample(0, 0, nl, 0, 0, nr, a, c, ns_out, 0, 0, 0, 0, 0, 0, 0, 0, 0)
        o_rank, o_pos = self.zeros(kept, self.torch.int64), self.zeros(kept, self.torch.int64)
        self._ready()
        self.ctx.dist_resample(l_rank.data_ptr(), l_pos.data_ptr(), nl, r_rank.data_ptr(), r_pos.data_ptr(), nr, a, c, ns_out,
                               S.data_ptr(), i_lo, i_hi, right_before, g.data_ptr(), p_lo, p_hi, o_rank.data_ptr(),
                               o_pos.data_ptr())
        return o_rank, o_pos

    def free_block(self, lb):
        lb.free()


WIRE_PAD = 8  # bytes the unpacking kernel may read past a packed stream (BWT_HIP_WIRE_PAD)


def wire_bytes(count):
    """Bytes of `count` symbols at two bits each."""
    return (int(count) + 3) // 4


def _overlap(lo0, hi0, lo1, hi1):
    return max(0, min(hi0, hi1) - max(lo0, lo1))


def dist_merge(ops, comm, blk, ns_req):
    """Merges the left block (held by the first half of comm.ranks) and the right block (second half); every rank of the
    group calls this with the block it holds a slice of.  Returns the merged DistBlock over all of comm.ranks."""
    torch = ops.torch
    ranks, me, k = comm.ranks, comm.me, comm.k
    G, q = len(ranks), len(ranks) // 2
    root, rroot = ranks[0], ranks[q]
    mine = blk.ranks.index(me)
    meta = comm.all_gather_i64([blk.a, blk.b, blk.s_rank, blk.e_rank, blk.n_samples, blk.sizes[mine]])
    a, c, b = meta[0][0], meta[0][1], meta[q][1]
    if meta[q][0] != c:
        raise RuntimeError("blocks are not adjacent: [%d,%d) and [%d,%d)" % (a, c, meta[q][0], b))
    m, r = c - a, b - c
    lsizes = [meta[j][5] for j in range(q)]
    rsizes = [meta[q + j][5] for j in range(q)]
    if sum(lsizes) != m or sum(rsizes) != r:
        raise RuntimeError("slice sizes do not add up to the blocks")
    s_rank_l, e_rank_l, nl, nr = meta[0][2], meta[0][3], meta[0][4], meta[q][4]

    # 1. the left block's L on every GPU of the group
    loff = [0]
    for x in lsizes:
        loff.append(loff[-1] + x)
    fullL = ops.empty(m, torch.uint8)
    packed_wire = ops.wire_packable()  # the same on every rank: they hold the same text
    if not packed_wire:
        comm.spread([blk.L if (j < q and ranks[j] == me) else None for j in range(G)], fullL, loff + [m] * q)
    else:
        # two bits per symbol on the wire: a quarter of the bytes of what is, at the top of the tree, the largest
        # transfer of the build (every GPU receives the whole left block)
        psz = [wire_bytes(x) for x in lsizes] + [0] * q
        poff = [0]
        for x in psz:
            poff.append(poff[-1] + ((x + WIRE_PAD + 15) & ~15))
        staged = ops.empty(poff[-1], torch.uint8)
        comm.spread([ops.wire_pack(blk.L) if (j < q and ranks[j] == me) else None for j in range(G)], staged, poff, psz)
        for j in range(q):
            if ranks[j] == me:
                fullL[loff[j]:loff[j + 1]].copy_(blk.L)
            else:
                ops.wire_unpack(staged[poff[j]:], lsizes[j], fullL[loff[j]:loff[j + 1]])
        if a == 0:  # the block that starts at text position 0 holds the terminator at s_rank, which has no code
            fullL[s_rank_l - a] = 0
    lb = ops.left_block(a, c, fullL, s_rank_l, e_rank_l)

    # 2. LF chains of this GPU's share of the right positions
    p_lo, p_hi = c + k * r // G, c + (k + 1) * r // G
    g = ops.chains(lb, p_lo, p_hi)
    ops.free_block(lb)

    # 3. every g value to the GPU that owns its part; then the right symbols each part interleaves
    g_parts, counts = ops.partition(g, m, G)
    table = comm.all_gather_i64(counts)  # table[src][part]
    g_in = comm.all_to_all(g_parts, counts, [table[s][k] for s in range(G)])
    R = [sum(table[s][d] for s in range(G)) for d in range(G)]  # right suffixes per part
    rb = [0]
    for x in R:
        rb.append(rb[-1] + x)
    if rb[-1] != r:
        raise RuntimeError("parts hold %d of %d right suffixes" % (rb[-1], r))
    roff = [0]
    for x in rsizes:
        roff.append(roff[-1] + x)
    if k >= q:
        own = (roff[k - q], roff[k - q + 1])
        send_splits = [_overlap(own[0], own[1], rb[d], rb[d + 1]) for d in range(G)]
        send = blk.L
    else:
        send_splits = [0] * G
        send = ops.empty(0, torch.uint8)
    recv_splits = [0] * q + [_overlap(roff[j], roff[j + 1], rb[k], rb[k + 1]) for j in range(q)]
    if not packed_wire:
        rL = comm.all_to_all(send, send_splits, recv_splits)
    else:
        pieces, at = [], 0
        for d in range(G):
            if send_splits[d]:
                pieces.append(ops.wire_pack(send[at:at + send_splits[d]]))
            at += send_splits[d]
        send_pk = torch.cat(pieces) if pieces else ops.empty(0, torch.uint8)
        got = comm.all_to_all(send_pk, [wire_bytes(x) for x in send_splits], [wire_bytes(x) for x in recv_splits], pad=WIRE_PAD)
        rL = ops.empty(sum(recv_splits), torch.uint8)
        at_b = at_s = 0
        for d in range(G):
            if recv_splits[d]:
                ops.wire_unpack(got[at_b:], recv_splits[d], rL[at_s:at_s + recv_splits[d]])
            at_b += wire_bytes(recv_splits[d])
            at_s += recv_splits[d]

    # 4. this part's slice of the merged block
    lo = [int(x) for x in dist_bounds(m, G)]
    i_lo = min(lo[k], m)
    i_hi = min(lo[k + 1], m)
    out, S = ops.merge_slice(fullL[i_lo:i_hi], rL, g_in)
    sizes = [min(lo[d + 1], m) - min(lo[d], m) + R[d] for d in range(G)]

    # 5. samples: everybody learns both sample sets, places what it can, the first GPU adds up.  Ranks and positions
    # are below 2^32, so they travel as 32-bit words (bit patterns in int32 tensors; adding zeros cannot carry).
    def narrow(t):
        return t.to(torch.int32)

    def widen(t):
        return t.to(torch.int64) & 0xFFFFFFFF

    wire = []
    for owner, cnt, src in ((root, nl, blk.samp_rank), (root, nl, blk.samp_pos), (rroot, nr, blk.samp_rank), (rroot, nr, blk.samp_pos)):
        t = narrow(src) if me == owner else ops.empty(cnt, torch.int32)
        comm.broadcast(t, owner)
        wire.append(widen(t))
    l_rank, l_pos, r_rank, r_pos = wire
    ns_out = min(ns_req, m + r)
    o_rank, o_pos = ops.resample(l_rank, l_pos, r_rank, r_pos, a, c, ns_out, S, i_lo, i_hi, rb[k], g, p_lo, p_hi)
    o_rank, o_pos = narrow(o_rank), narrow(o_pos)
    comm.sum_to(o_rank, root)
    comm.sum_to(o_pos, root)
    o_rank, o_pos = widen(o_rank), widen(o_pos)
    new = DistBlock(a, b, ranks, sizes, out)
    if me == root:
        first = [int(x) for x in o_rank[:1].cpu().tolist()] + [int(x) for x in o_rank[-1:].cpu().tolist()]
        last_pos = int(o_pos[-1:].cpu().tolist()[0])
        if last_pos != b - 1:
            raise RuntimeError("merge [%d,%d): last sample is at %d, not at b-1" % (a, b, last_pos))
        new.s_rank, new.e_rank, new.samp_rank, new.samp_pos = first[0], first[1], o_rank, o_pos
    return new


def make_groups(dist, world):
    """Process groups of every subtree: groups[G][i] covers ranks [i*G, (i+1)*G).  Every rank must call this."""
    groups = {}
    G = 2
    while G <= world:
        groups[G] = [dist.new_group(list(range(i, i + G))) for i in range(0, world, G)]
        G *= 2
    return groups


def warm_groups(dist, groups, rank, world, device, host_staging):
    """One tiny call of every collective a merge uses, on each group this rank belongs to: the transport sets up its
    communicators and peer connections at first use, which is set-up cost and not part of a build."""
    import torch
    dev = "cpu" if host_staging else device
    G = 2
    while G <= world:
        i = rank // G
        comm = Comm(dist, groups[G][i], range(i * G, (i + 1) * G), rank, dev, False)
        comm.all_gather_i64([rank])
        got = comm.all_to_all(torch.full((G,), rank, dtype=torch.int32, device=dev), [1] * G, [1] * G)
        if [int(x) for x in got.cpu().tolist()] != list(comm.ranks):
            raise RuntimeError("all-to-all warm-up returned %r on rank %d" % (got.cpu().tolist(), rank))
        out = torch.zeros(G, dtype=torch.uint8, device=dev)
        mine = torch.full((1,), comm.k + 1, dtype=torch.uint8, device=dev)
        comm.spread([mine if r == rank else None for r in comm.ranks], out, list(range(G + 1)))
        if [int(x) for x in out.cpu().tolist()] != list(range(1, G + 1)):
            raise RuntimeError("point-to-point warm-up returned %r on rank %d" % (out.cpu().tolist(), rank))
        t = torch.ones(1, dtype=torch.int32, device=dev)
        comm.broadcast(t, comm.ranks[0])
        comm.sum_to(t, comm.ranks[0])
        G *= 2
    if not host_staging and getattr(device, "type", str(device)) == "cuda":
        torch.cuda.synchronize(device)


def pmbwt_dist(ops, rank, world, n, leaves_per_rank, dist, groups, device, host_staging):
    """The whole BWT of [0,n) over `world` ranks; returns this rank's DistBlock of the final block (slice `rank` of L;
    samples and special ranks on rank 0)."""
    from .multi_gpu import leaf_threshold, split_ranges
    depth = world.bit_length() - 1
    assert 1 << depth == world, "world size must be a power of two"
    th = leaf_threshold(n, world * leaves_per_rank)
    a, b = split_ranges(0, n, depth)[rank]
    blk = ops.local_block(a, b, th, rank)
    G = 2
    while G <= world:
        i = rank // G
        comm = Comm(dist, groups[G][i], range(i * G, (i + 1) * G), rank, device, host_staging)
        blk = dist_merge(ops, comm, blk, ops.ns)
        G *= 2
    return blk


def gather_L(blk, rank, world, dist, device, host_staging):
    """All slices of the final block on rank 0 as one numpy array (tests; the benchmark leaves them where they are)."""
    import torch
    wire_dev = "cpu" if host_staging else device
    mine = blk.L.cpu() if host_staging else blk.L
    if rank == 0:
        parts = [mine]
        for src in range(1, world):
            buf = torch.empty(blk.sizes[src], dtype=torch.uint8, device=wire_dev)
            dist.recv(buf, src)
            parts.append(buf)
        return np.concatenate([p.cpu().numpy() for p in parts])
    dist.send(mine.contiguous(), 0)
    return None
