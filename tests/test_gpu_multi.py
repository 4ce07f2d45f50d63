"""GPU tests of the multi-rank path on a one-GPU box: block transport through device buffers, and a two-rank
merge tree (both ranks on GPU 0, transport over gloo) against the single-process build."""
import os

import numpy as np
import pytest

from oracle import gen_text
from tests.gpu_util import ctx  # noqa: F401
from tests.test_multi_gpu_cpu import _launch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_block_export_import_through_device_tensors(ctx):
    import torch
    n = 50000
    T = gen_text(n, 5, 4)
    ctx.set_text(T)
    a, c, b = 100, 20000, 49000
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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,lpr", [(2, 2), (4, 1)])
def test_two_rank_tree_equals_single_process(ctx, tmp_path, world, lpr):
    from bwt_amd import multi_gpu
    n, ns = 400000, 2000
    out = str(tmp_path / "res.npz")
    _launch(world, ["--ops", "hip", "--text_len", str(n), "--lpr", str(lpr), "--ns", str(ns)], out, timeout=800)
    res = np.load(out)
    ctx.set_text(gen_text(n, 42, 4))
    ref = ctx.pmbwt(multi_gpu.leaf_threshold(n, world * lpr), ns)
    assert (res["L"] == ref.L).all()
    assert (int(res["s_rank"]), int(res["e_rank"])) == (ref.s_rank, ref.e_rank)
    r, p = ref.samples()
    assert (res["rank"] == r).all() and (res["pos"] == p).all()
