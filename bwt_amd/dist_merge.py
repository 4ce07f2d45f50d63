"""One text, several GPUs, every merge spread over all GPUs of its subtree.

bwt_amd.multi_gpu.pmbwt_tree leaves the upper levels of the merge tree to fewer and fewer GPUs.  Here a block of
the tree that spans q GPUs stays distributed over them as q slices of its L (rank ranges), each keeping the block's
samples of its own text range, and a merge of two such blocks (reference include/bwt.h:791-847) runs on all 2q:

  1. the left block's L is sent to every GPU of the group (point-to-point, all links at once); each builds the left
     rank structure; 2. the right block's text positions are split evenly and every GPU runs the LF chains of its share
     (g_j = number of left suffixes below T[j:] depends on the left block and the text only, and the text is on every
     GPU); 3. the merge is cut by left rank into 2q parts (bwt_hip_dist_layout): an all-to-all moves every g value to
     the GPU that owns its part, a second one the right symbols each part interleaves; 4. every GPU builds the prefix
     sums of its part and writes its slice of the merged L; 5. the samples do not move: every GPU keeps those of its
     own text range, asks the GPU that knows for the number each new rank needs (the part's prefix sums for a left
     sample, g for a right one: two small all-to-alls) and thins its own by the reference's rule.

Exchanged per merge and GPU: the left L in (m bytes, or m/4 when the text's symbols span at most four values and
L crosses the links as two bits per symbol), about 4r/G bytes of g values and r/G (or r/4G) bytes of right symbols
through the all-to-alls (half of that where the g values travel grouped, see dist_merge), 8 bytes per sample.  The result is the same block the
single-GPU recursion builds: L bit for bit (the slices concatenated), and the same samples, because the same tree is
merged with the same rules.

torch is plumbing: tensors are buffers, torch.distributed is the transport (backend "nccl" is RCCL over xGMI; gloo with
host staging in the tests).  All BWT work goes through the C-ABI (include/bwt_hip.h, bwt_hip_dist_*).
"""
import numpy as np


class DistBlock:
    """Block [a,b) spread over `ranks` (global ranks, ascending): slice i of L, `sizes[i]` symbols, is on ranks[i].
    Every rank keeps the block's samples whose positions lie in the text range it built its own subtree on (rank =
    block-relative rank + a, pos: 32-bit words held in int32 tensors, sorted by position; the ranks in order hold the
    samples in order).  s_rank is known on ranks[0], e_rank on ranks[-1]."""

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
        d = DistBlock(a, b, [me], [b - a], L, blk.s_rank, blk.e_rank, r.to(self.torch.int32), p.to(self.torch.int32))
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

    def samples_route(self, is_left, rank, pos, a, c, m, r, members, shift):
        n = int(pos.numel())
        queries = self.empty(n, self.torch.int32)
        perm = self.empty(n if is_left else 0, self.torch.int32)
        self._ready()
        counts = self.ctx.dist_samples_route(is_left, rank.data_ptr(), pos.data_ptr(), n, a, c, m, r, members, shift,
                                             queries.data_ptr(), perm.data_ptr())
        return queries, perm, [int(x) for x in counts]

    def layout(self, m, parts):
        return self.ctx.dist_layout(m, parts)

    def chains_grouped(self, lb, p_lo, p_hi):
        """(g in text order, 16-bit words grouped by gap bucket, bucket bounds [buckets + 1], buckets, fillers among the words)"""
        cnt = p_hi - p_lo
        seg = self.seg if self.seg > 0 else 1024
        cap = cnt + seg
        m = lb.b - lb.a
        g = self.empty(cnt, self.torch.int32)
        words = self.empty(cap, self.torch.int16)
        starts = self.empty((m >> 13) + 2, self.torch.int32)
        self._ready()
        n_words, nb, fillers = self.ctx.dist_chains_grouped(lb, p_lo, p_hi, self.seg, g.data_ptr(), words.data_ptr(), cap, starts.data_ptr())
        return g, words[:n_words], starts[:nb + 1], nb, fillers

    def merge_slice_grouped(self, lL, rL, words, word_off, starts, nbk):
        mL, rn = int(lL.numel()), int(rL.numel())
        out = self.empty(mL + rn, self.torch.uint8)
        S = self.empty(mL + 1, self.torch.int32)
        self._ready()
        self.ctx.dist_merge_slice_grouped(lL.data_ptr(), mL, rL.data_ptr(), rn, words.data_ptr(), word_off, starts.data_ptr(), nbk,
                                          out.data_ptr(), S.data_ptr())
        return out, S

    def samples_answer(self, queries, n_left, n_right, S, i_lo, right_before, a, g):
        out = self.empty(n_left + n_right, self.torch.int32)
        self._ready()
        self.ctx.dist_samples_answer(queries.data_ptr(), n_left, n_right, S.data_ptr(), i_lo, right_before, a, g.data_ptr(),
                                     out.data_ptr())
        return out

    def samples_place(self, is_left, rank, pos, answers, perm, a, c, first_index, n_all, ns_out):
        n = int(pos.numel())
        kept = self.ctx.dist_samples_place(is_left, 0, 0, n, 0, 0, a, c, first_index, n_all, ns_out, 0, 0)
        o_rank, o_pos = self.empty(kept, self.torch.int32), self.empty(kept, self.torch.int32)
        self._ready()
        self.ctx.dist_samples_place(is_left, rank.data_ptr(), pos.data_ptr(), n, answers.data_ptr(), perm.data_ptr(), a, c,
                                    first_index, n_all, ns_out, o_rank.data_ptr(), o_pos.data_ptr())
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
    s_rank_l, e_rank_l = meta[0][2], meta[q - 1][3]

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

    # Where the gap histogram groups its values in two passes (left blocks of 2^21 suffixes and more) the g values
    # travel grouped: the sender groups the values of its own chains by gap bucket and sends each part the 16-bit
    # words of the part's buckets plus the buckets' bounds -- half the bytes of the values; parts then begin at
    # multiples of 8192 left ranks (shift 13).
    lo, shift = ops.layout(m, G)
    lo = [int(x) for x in lo]

    # 2. LF chains of this GPU's share of the right positions
    p_lo, p_hi = c + k * r // G, c + (k + 1) * r // G
    if not shift:
        g = ops.chains(lb, p_lo, p_hi)
        ops.free_block(lb)

        # 3. every g value to the GPU that owns its part; then the right symbols each part interleaves
        g_parts, counts = ops.partition(g, m, G)
        table = comm.all_gather_i64(counts)  # table[src][part]
        g_in = comm.all_to_all(g_parts, counts, [table[s][k] for s in range(G)])
        R = [sum(table[s][d] for s in range(G)) for d in range(G)]  # right suffixes per part
    else:
        g, words, starts, nb, fillers = ops.chains_grouped(lb, p_lo, p_hi)
        ops.free_block(lb)
        B = [min(x >> shift, nb) for x in lo[:G]] + [nb]  # part d owns the buckets B[d] .. B[d+1]-1
        at = [int(x) for x in starts[torch.tensor(B, dtype=torch.int64, device=starts.device)].cpu().tolist()]
        wcount = [at[d + 1] - at[d] for d in range(G)]
        table = comm.all_gather_i64(wcount + [fillers])  # table[src][part]; [src][G] = the sender's fillers, all in the last part
        # (as bytes: the transports carry no 16-bit integers)
        w_in = comm.all_to_all(words[:at[G]].view(torch.uint8), [2 * x for x in wcount], [2 * table[s][k] for s in range(G)],
                               pad=16).view(torch.int16)
        nbk = B[k + 1] - B[k]
        st_out = torch.cat([starts[B[d]:B[d + 1] + 1] for d in range(G)])
        st_in = comm.all_to_all(st_out, [B[d + 1] - B[d] + 1 for d in range(G)], [nbk + 1] * G)
        R = [sum(table[s][d] for s in range(G)) for d in range(G)]
        R[G - 1] -= sum(table[s][G] for s in range(G))
        word_off = [sum(table[s][k] for s in range(j)) for j in range(G)]
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
    i_lo = min(lo[k], m)
    i_hi = min(lo[k + 1], m)
    if not shift:
        out, S = ops.merge_slice(fullL[i_lo:i_hi], rL, g_in)
    else:
        out, S = ops.merge_slice_grouped(fullL[i_lo:i_hi], rL, w_in, word_off, st_in, nbk)
    sizes = [min(lo[d + 1], m) - min(lo[d], m) + R[d] for d in range(G)]

    # 5. the samples stay with their holders: requests out, answers back, every holder thins its own
    held = [meta[j][4] for j in range(G)]
    n_all, first_index = sum(held), sum(held[:k])
    ns_out = min(ns_req, m + r)
    queries, perm, ask = ops.samples_route(k < q, blk.samp_rank, blk.samp_pos, a, c, m, r, G, shift)
    asked = comm.all_gather_i64(ask)  # asked[src][member]
    incoming = [asked[j][k] for j in range(G)]
    q_in = comm.all_to_all(queries, ask, incoming)
    answers = ops.samples_answer(q_in, sum(incoming[:q]), sum(incoming[q:]), S, i_lo, rb[k], a, g)
    a_in = comm.all_to_all(answers, incoming, ask)
    o_rank, o_pos = ops.samples_place(k < q, blk.samp_rank, blk.samp_pos, a_in, perm, a, c, first_index, n_all, ns_out)
    new = DistBlock(a, b, ranks, sizes, out, samp_rank=o_rank, samp_pos=o_pos)

    def word(t, i):  # element i of an int32 tensor that holds unsigned 32-bit values
        return int(t[i:i + 1].cpu().tolist()[0]) & 0xFFFFFFFF

    # the block's first sample is at position a (the first holder's first), its last at b - 1 (the last holder's last)
    if k == 0:
        if word(o_pos, 0) != a:
            raise RuntimeError("merge [%d,%d): first sample is at %d, not at a" % (a, b, word(o_pos, 0)))
        new.s_rank = word(o_rank, 0)
    if k == G - 1:
        last = int(o_pos.numel()) - 1
        if word(o_pos, last) != b - 1:
            raise RuntimeError("merge [%d,%d): last sample is at %d, not at b-1" % (a, b, word(o_pos, last)))
        new.e_rank = word(o_rank, last)
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


def gather_samples(blk, rank, world, dist, device, host_staging):
    """The final block's samples (ranks, positions: int64 numpy arrays) and its (s_rank, e_rank) on rank 0 (tests; the
    benchmark leaves them where they are)."""
    import torch
    wire_dev = "cpu" if host_staging else device
    counts = torch.zeros(world, dtype=torch.int64, device=wire_dev)
    counts[rank] = blk.n_samples
    dist.all_reduce(counts)
    counts = [int(x) for x in counts.cpu().tolist()]
    ends = torch.tensor([blk.e_rank if rank == world - 1 else 0], dtype=torch.int64, device=wire_dev)
    dist.all_reduce(ends)
    mine = torch.stack([blk.samp_rank, blk.samp_pos]).to(wire_dev).contiguous()
    if rank == 0:
        parts = [mine]
        for src in range(1, world):
            buf = torch.empty((2, counts[src]), dtype=torch.int32, device=wire_dev)
            dist.recv(buf, src)
            parts.append(buf)
        both = np.concatenate([p.cpu().numpy() for p in parts], axis=1).view(np.uint32).astype(np.int64)
        return both[0], both[1], blk.s_rank, int(ends.item())
    dist.send(mine, 0)
    return None
