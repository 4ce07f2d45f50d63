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
    ap.add_argument("--text_len", dest="n", type=int, default=100000)
    ap.add_argument("--sigma", type=int, default=4)
    ap.add_argument("--lpr", type=int, default=2)
    ap.add_argument("--ns", type=int, default=500)
    ap.add_argument("--seg", type=int, default=0)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import torch.distributed as dist
    from bwt_amd import multi_gpu
    from oracle import gen_text
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    T = gen_text(args.n, 42, args.sigma)
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
