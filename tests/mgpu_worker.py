"""Worker of the multi-rank tests (launched by torch.distributed.run): builds one BWT over all ranks with
bwt_amd.multi_gpu.pmbwt_tree and writes rank 0's result to a file.

    --ops hip     blocks live on the GPU (every rank uses GPU 0 of a one-GPU box), transport over gloo host tensors
    --ops oracle  blocks are computed by the CPU oracle (schedule/transport test without any GPU)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleOps:
    """The same interface as bwt_amd.multi_gpu.HipOps, with the CPU oracle doing the block work (tests only)."""

    def __init__(self, T, opt):
        from oracle import Oracle
        self.O, self.T, self.opt = Oracle, T, opt
        self.L = np.zeros(len(T), dtype=np.uint8)

    def rec(self, a, b, th):
        if b - a <= th:
            return self.O.leaf(self.T, a, b, self.L, self.opt)
        c = (a + b) // 2
        return self.O.merge(self.rec(a, c, th), self.rec(c, b, th), self.opt)

    def merge(self, l, r):
        return self.O.merge(l, r, self.opt)

    def header(self, blk):
        return [blk.a, blk.b, blk.s_rank, blk.e_rank, blk.n_samples]

    def pack(self, blk):
        import torch
        r, p = blk.samples()
        return torch.from_numpy(self.L[blk.a:blk.b].copy()), torch.from_numpy(r), torch.from_numpy(p)

    def empty(self, hdr):
        import torch
        a, b, _, _, ns = hdr
        return torch.empty(b - a, dtype=torch.uint8), torch.empty(ns, dtype=torch.int64), torch.empty(ns, dtype=torch.int64)

    def unpack(self, hdr, Lt, rt, pt):
        # the oracle has no "upload": rebuild the block from the text (deterministic) and check it against the wire
        a, b, s_rank, e_rank, ns = hdr
        blk = self.rec(a, b, self.th)
        assert (self.L[a:b] == Lt.numpy()).all() and (blk.s_rank, blk.e_rank) == (s_rank, e_rank)
        r, p = blk.samples()
        assert (r == rt.numpy()).all() and (p == pt.numpy()).all()
        return blk


class NumpyDistOps:
    """The interface of bwt_amd.dist_merge.HipDistOps on the CPU (tests only): leaves from the oracle, the per-part
    steps of a distributed merge restated in numpy from their definitions (g from a global suffix array)."""

    def __init__(self, T, opt, ns, seg, th):
        import torch
        from oracle import Oracle
        self.torch, self.T, self.ns, self.seg = torch, T, ns, seg
        self.base = OracleOps(T, opt)
        self.base.th = th
        n = len(T)
        sa = Oracle.sa_range(T, 0, n)
        self.rank_of = np.empty(n, dtype=np.int64)
        self.rank_of[sa] = np.arange(n)

    def empty(self, n, dtype):
        return self.torch.zeros(int(n), dtype=dtype)

    def local_block(self, a, b, th, me):
        from bwt_amd.dist_merge import DistBlock
        blk = self.base.rec(a, b, th)
        r, p = blk.samples()
        t = self.torch
        return DistBlock(a, b, [me], [b - a], t.from_numpy(self.base.L[a:b].copy()), blk.s_rank, blk.e_rank,
                         self._words(r), self._words(p))

    def _words(self, v):  # unsigned 32-bit values in an int32 tensor, as the device path holds them
        return self.torch.from_numpy(np.asarray(v).astype(np.uint32).view(np.int32).copy())

    @staticmethod
    def _ints(t):
        return t.numpy().view(np.uint32).astype(np.int64)

    def left_block(self, a, c, L, s_rank, e_rank):
        return (a, c)

    # two bits per symbol on the wire, as include/bwt_hip.h defines them: code = (symbol - smallest symbol) & 3,
    # four symbols per byte, the first in the lowest bits
    def _min_sym(self):
        return int(self.T[self.T > 0].min())

    def wire_packable(self):
        body = self.T[self.T > 0]
        return int(body.max()) - int(body.min()) <= 3

    def wire_pack(self, t):
        v = ((t.numpy().astype(np.int64) - self._min_sym()) & 3).astype(np.uint8)
        v = np.concatenate([v, np.zeros(-len(v) % 4, dtype=np.uint8)]).reshape(-1, 4)
        return self.torch.from_numpy((v[:, 0] | (v[:, 1] << 2) | (v[:, 2] << 4) | (v[:, 3] << 6)).astype(np.uint8))

    def wire_unpack(self, packed, count, dst):
        p = packed.numpy()[:(count + 3) // 4]
        codes = np.stack([(p >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(-1)[:count]
        dst.copy_(self.torch.from_numpy((codes + self._min_sym()).astype(np.uint8)))

    def free_block(self, lb):
        pass

    def chains(self, lb, p_lo, p_hi):
        a, c = lb
        left = np.sort(self.rank_of[a:c])
        g = np.searchsorted(left, self.rank_of[p_lo:p_hi]).astype(np.uint32)
        return self.torch.from_numpy(g.view(np.int32).copy())

    @staticmethod
    def shift_of(m):
        """13 where the g values of a merge travel grouped by gap bucket: left blocks whose gap indices 0..m need 9 to
        18 bits above the 13 a bucket spans (include/bwt_hip.h, bwt_hip_dist_layout)."""
        return 13 if 8 < (m + 1).bit_length() - 13 <= 18 else 0

    @classmethod
    def bounds(cls, m, parts, shift=0):
        scale = (parts << 32) // ((m >> shift) + 1)
        return [(((k << 32) + scale - 1) // scale) << shift for k in range(parts)] + [m + 1], scale

    def layout(self, m, parts):
        shift = self.shift_of(m)
        return self.bounds(m, parts, shift)[0], shift

    def chains_grouped(self, lb, p_lo, p_hi):
        """bwt_hip_dist_chains_grouped restated: the values' low 13 bits grouped by bucket (here in descending order
        inside a bucket: any order is allowed), bucket bounds, and a few fillers m + 1 as a short last chain leaves them."""
        a, c = lb
        m = c - a
        g = self.chains(lb, p_lo, p_hi)
        v = self._ints(g)
        nb = (m >> 13) + 1
        n_fill = 5 if len(v) else 0
        allv = np.concatenate([v, np.full(n_fill, m + 1, dtype=np.int64)])
        order = np.lexsort((-allv, allv >> 13))
        allv = allv[order]
        starts = np.searchsorted(allv >> 13, np.arange(nb + 1), side="left")
        n_words = int(starts[nb])
        fillers = n_fill if ((m + 1) >> 13) < nb else 0
        words = self.torch.from_numpy((allv[:n_words] & 8191).astype(np.uint16).view(np.int16).copy())
        return g, words, self._words(starts), nb, fillers

    def merge_slice_grouped(self, lL, rL, words, word_off, starts, nbk):
        mL = int(lL.numel())
        w = words.numpy().view(np.uint16).astype(np.int64)
        st = self._ints(starts).reshape(len(word_off), nbk + 1)
        vals = []
        for s_, off in enumerate(word_off):
            for b in range(nbk):
                lo_, hi_ = off + st[s_][b] - st[s_][0], off + st[s_][b + 1] - st[s_][0]
                vals.append(w[lo_:hi_] + (b << 13))
        g = np.concatenate(vals) if vals else np.zeros(0, dtype=np.int64)
        g = g[g <= mL]  # fillers lie above the part's last gap index
        return self.merge_slice(lL, rL, self._words(g))

    def partition(self, g, m, parts):
        lo, scale = self.bounds(m, parts)
        v = g.numpy().view(np.uint32).astype(np.uint64)
        part = ((v * np.uint64(scale)) >> np.uint64(32)).astype(np.int64)
        order = np.argsort(part, kind="stable")[::-1].copy()   # any order inside a part is allowed: use a reversed one
        order = order[np.argsort(part[order], kind="stable")]
        out = (v[order] - np.array(lo, dtype=np.uint64)[part[order]]).astype(np.uint32)
        counts = np.bincount(part, minlength=parts)
        return self.torch.from_numpy(out.view(np.int32).copy()), [int(x) for x in counts]

    def merge_slice(self, lL, rL, g_rel):
        mL, rn = int(lL.numel()), int(rL.numel())
        g = g_rel.numpy().view(np.uint32).astype(np.int64)
        S = np.cumsum(np.bincount(g, minlength=mL + 1))[:mL + 1]
        out = np.zeros(mL + rn, dtype=np.uint8)
        is_left = np.zeros(mL + rn, dtype=bool)
        where = np.arange(mL) + S[:mL]
        is_left[where] = True
        out[where] = lL.numpy()
        out[~is_left] = rL.numpy()
        return self.torch.from_numpy(out), self.torch.from_numpy(S.astype(np.uint32).view(np.int32).copy())

    # the three sample steps as include/bwt_hip.h defines them (bwt_hip_dist_samples_route / _answer / _place)
    def samples_route(self, is_left, rank, pos, a, c, m, r, members, shift):
        if is_left:
            lo, scale = self.bounds(m, members, shift)
            rk = self._ints(rank) - a
            part = (((rk >> shift).astype(np.uint64) * np.uint64(scale)) >> np.uint64(32)).astype(np.int64)
            order = np.argsort(part, kind="stable")[::-1].copy()   # any order inside a part is allowed: a reversed one
            order = order[np.argsort(part[order], kind="stable")]
            queries = rk[order] - np.array(lo, dtype=np.int64)[part[order]]
            return self._words(queries), self._words(order), [int(x) for x in np.bincount(part, minlength=members)]
        p = self._ints(pos)
        bound = np.array([c + d * r // members for d in range(members + 1)], dtype=np.int64)
        first = np.searchsorted(p, bound, side="left")
        assert first[0] == 0 and first[-1] == len(p)
        member = np.searchsorted(first[1:], np.arange(len(p)), side="right")
        return self._words(p - bound[member]), self._words([]), [int(x) for x in np.diff(first)]

    def samples_answer(self, queries, n_left, n_right, S, i_lo, right_before, a, g):
        qv, Sn, gn = self._ints(queries), self._ints(S), self._ints(g)
        left = a + i_lo + right_before + qv[:n_left] + Sn[qv[:n_left]]
        right = gn[qv[n_left:n_left + n_right]]
        return self._words(np.concatenate([left, right]))

    def samples_place(self, is_left, rank, pos, answers, perm, a, c, first_index, n_all, ns_out):
        A, B = ns_out - 1, n_all - 1

        def taken(i):  # include/bwt.h:412, :436 in closed form
            if i == 0:
                return 0
            if B == 0:
                return 1
            return min(i, 1 + ((i - 1) * A) // B)
        rk, ps, ans = self._ints(rank), self._ints(pos), self._ints(answers)
        owner = self._ints(perm) if is_left else np.arange(len(ps))
        base = taken(first_index)
        kept = taken(first_index + len(ps)) - base
        o_rank, o_pos = np.zeros(kept, dtype=np.int64), np.zeros(kept, dtype=np.int64)
        for x in range(len(ps)):
            i = int(owner[x])
            t0, t1 = taken(first_index + i), taken(first_index + i + 1)
            if t1 != t0:
                o_rank[t0 - base] = ans[x] if is_left else rk[i] - (c - a) + ans[x]
                o_pos[t0 - base] = ps[i]
        return self._words(o_rank), self._words(o_pos)


def export_import_check():
    """Single process: a block goes out through torch device tensors and comes back; merge results must not change.
    torch is imported before the HIP library is loaded, so both use the one HIP runtime torch ships."""
    import torch
    import bwt_amd
    from oracle import gen_text
    n = 50000
    T = gen_text(n, 5, 4)
    ctx = bwt_amd.Context(0)
    ctx.set_text(T)
    a, b = 100, 49000
    c = (a + b) // 2  # where bwt_rec splits
    lb, rb = ctx.leaf(a, c, 300), ctx.leaf(c, b, 300)
    dev = torch.device("cuda", 0)
    Lt = torch.empty(b - c, dtype=torch.uint8, device=dev)
    rt = torch.empty(rb.n_samples, dtype=torch.int64, device=dev)
    pt = torch.empty(rb.n_samples, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    rb.export_device(Lt.data_ptr(), rt.data_ptr(), pt.data_ptr())
    r, p = rb.samples()
    assert (Lt.cpu().numpy() == rb.L).all() and (rt.cpu().numpy() == r).all() and (pt.cpu().numpy() == p).all()
    rb2 = ctx.import_block_device(c, b, Lt.data_ptr(), rb.s_rank, rb.e_rank, rt.data_ptr(), pt.data_ptr(), rb.n_samples)
    ref = ctx.rec(a, b, c - a + 1, 300)  # the same two leaves and merge in one call
    m = ctx.merge(lb, rb2, 300)
    assert (m.L == ref.L).all() and (m.s_rank, m.e_rank) == (ref.s_rank, ref.e_rank)
    for x, y in zip(m.samples(), ref.samples()):
        assert (x == y).all()
    print("EXPORT_IMPORT_OK")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "export_import":
        return export_import_check()
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="hip")
    ap.add_argument("--algo", default="tree", help="tree: blocks move to the left owner; dist: every merge on all GPUs of its subtree")
    ap.add_argument("--text_len", dest="n", type=int, default=100000)
    ap.add_argument("--sigma", type=int, default=4)
    ap.add_argument("--lpr", type=int, default=2)
    ap.add_argument("--ns", type=int, default=500)
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--direct", type=int, default=0, help="dist algo: 1 = hand the ops' tensors to the transport as they are (what an nccl run does)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import torch.distributed as dist
    from bwt_amd import multi_gpu
    from oracle import gen_text
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    T = gen_text(args.n, 42, args.sigma)
    if args.algo == "dist":
        from bwt_amd import dist_merge
        th = multi_gpu.leaf_threshold(args.n, world * args.lpr)
        if args.ops == "hip":
            import torch
            import bwt_amd
            dev = torch.device("cuda", 0)
            ctx = bwt_amd.Context(0)
            ctx.set_text(T)
            ops = dist_merge.HipDistOps(ctx, args.ns, args.seg, dev)
        else:
            from oracle import Oracle
            dev = "cpu"
            opt = Oracle.opt(args.n, alpha_min=1, alpha_max=args.sigma, ssa_n_samples=args.ns, build_gap_segment_sz=args.seg)
            ops = NumpyDistOps(T, opt, args.ns, args.seg, th)
        groups = dist_merge.make_groups(dist, world)
        staging = not args.direct
        dist_merge.warm_groups(dist, groups, rank, world, dev, staging)
        blk = dist_merge.pmbwt_dist(ops, rank, world, args.n, args.lpr, dist, groups, dev, staging)
        if args.ops == "hip":
            # what bench.py's N > 1 leg reports as verified_against_one_gpu: true here, false for a slice with one byte changed
            v = multi_gpu.verify_against_one_gpu(ctx, blk, "dist", rank, world, th, args.ns, args.seg, dist, dev, coll_device="cpu")
            assert v["L_slices_equal"] and v["samples_equal"] and v["special_ranks_equal"] and not v["ranks_with_errors"], v
            bad = dist_merge.DistBlock(blk.a, blk.b, blk.ranks, blk.sizes, blk.L.clone(), blk.s_rank, blk.e_rank, blk.samp_rank, blk.samp_pos)
            if rank == world - 1:
                bad.L[bad.L.numel() // 2] ^= 1
            v = multi_gpu.verify_against_one_gpu(ctx, bad, "dist", rank, world, th, args.ns, args.seg, dist, dev, coll_device="cpu")
            assert not v["L_slices_equal"] and v["samples_equal"], v
        L = dist_merge.gather_L(blk, rank, world, dist, dev, staging)
        samples = dist_merge.gather_samples(blk, rank, world, dist, dev, staging)
        if rank == 0:
            np.savez(args.out, L=L, rank=samples[0], pos=samples[1], s_rank=samples[2], e_rank=samples[3], sizes=np.array(blk.sizes))
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.ops == "hip":
        import bwt_amd
        ctx = bwt_amd.Context(0)
        ctx.set_text(T)
        ops = multi_gpu.HipOps(ctx, args.ns, args.seg, False)
        free = lambda b: b.free()
    else:
        from oracle import Oracle
        opt = Oracle.opt(args.n, alpha_min=1, alpha_max=args.sigma, ssa_n_samples=args.ns, build_gap_segment_sz=args.seg)
        ops = OracleOps(T, opt)
        ops.th = multi_gpu.leaf_threshold(args.n, world * args.lpr)
        free = lambda b: None
    blk = multi_gpu.pmbwt_tree(ops, rank, world, args.n, args.lpr, dist, free=free)
    if rank == 0:
        L = blk.L if args.ops == "hip" else ops.L
        r, p = blk.samples()
        np.savez(args.out, L=np.asarray(L), rank=r, pos=p, s_rank=blk.s_rank, e_rank=blk.e_rank)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
