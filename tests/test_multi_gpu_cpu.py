"""CPU tests of the multi-rank path: range/threshold arithmetic, the merge-tree schedule, and a world_size-2 run
over gloo in which the CPU oracle stands in for the GPU (transport and schedule are what is tested)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from bwt_amd import multi_gpu
from oracle import Oracle, gen_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_ranges_follow_midpoint_recursion():
    # include/bwt.h:860 of the reference: c = (a + b) / 2
    assert multi_gpu.split_ranges(0, 10, 0) == [(0, 10)]
    assert multi_gpu.split_ranges(0, 10, 1) == [(0, 5), (5, 10)]
    assert multi_gpu.split_ranges(0, 11, 2) == [(0, 2), (2, 5), (5, 8), (8, 11)]
    r = multi_gpu.split_ranges(0, 3100000000, 3)
    assert r[0][0] == 0 and r[-1][1] == 3100000000 and all(x[1] == y[0] for x, y in zip(r, r[1:]))


@pytest.mark.parametrize("n,leaves", [(1000, 2), (100001, 8), (10 ** 9, 2), (3100000000, 8), (3100000000, 16)])
def test_leaf_threshold_reproduces_the_tree(n, leaves):
    th = multi_gpu.leaf_threshold(n, leaves)

    def rec(a, b):
        return [(a, b)] if b - a <= th else rec(a, (a + b) // 2) + rec((a + b) // 2, b)
    assert rec(0, n) == multi_gpu.split_ranges(0, n, leaves.bit_length() - 1)


def test_tree_levels():
    assert multi_gpu.tree_levels(1) == []
    assert multi_gpu.tree_levels(2) == [(1, [(0, 1)])]
    assert multi_gpu.tree_levels(8) == [(1, [(0, 1), (2, 3), (4, 5), (6, 7)]), (2, [(0, 2), (4, 6)]), (4, [(0, 4)])]


def _launch(world, extra, out, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000),
           os.path.join(ROOT, "tests", "mgpu_worker.py"), "--out", out] + extra
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]


@pytest.mark.parametrize("world,lpr", [(2, 2), (4, 1)])
def test_gloo_tree_with_oracle_blocks(tmp_path, world, lpr):
    n, ns, seg = 20000, 200, 64
    out = str(tmp_path / "res.npz")
    _launch(world, ["--ops", "oracle", "--text_len", str(n), "--lpr", str(lpr), "--ns", str(ns), "--seg", str(seg)], out)
    res = np.load(out)
    T = gen_text(n, 42, 4)
    th = multi_gpu.leaf_threshold(n, world * lpr)
    o = Oracle.opt(n, alpha_min=1, alpha_max=4, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
    ob = Oracle.pmbwt(T, o)
    assert (res["L"] == ob.L).all()
    assert (int(res["s_rank"]), int(res["e_rank"])) == (ob.s_rank, ob.e_rank)
    r, p = ob.samples()
    assert (res["rank"] == r).all() and (res["pos"] == p).all()


def test_dist_bounds_partition_the_left_ranks():
    """bwt_hip_dist_bounds is host arithmetic (the library loads without a GPU): parts are contiguous, cover [0, m]
    and agree with part(v) = (v * scale) >> 32."""
    import bwt_amd
    for m, parts in [(0, 2), (1, 2), (5, 8), (1000, 2), (12345, 8), (387500000, 8), (1550000000, 8), (2 ** 32 - 5, 4)]:
        lo = [int(x) for x in bwt_amd.dist_bounds(m, parts)]
        scale = (parts << 32) // (m + 1)
        assert lo[0] == 0 and lo[-1] == m + 1 and all(x <= y for x, y in zip(lo, lo[1:]))
        for k in range(1, parts):
            v = lo[k]
            if v <= m:
                assert (v * scale) >> 32 >= k and ((v - 1) * scale) >> 32 < k
        assert (m * scale) >> 32 < parts


@pytest.mark.parametrize("world,lpr,direct,sigma", [(2, 1, 0, 4), (4, 1, 0, 4), (4, 2, 1, 4), (8, 1, 1, 4), (2, 1, 1, 4),
                                                    (4, 1, 1, 6), (2, 2, 0, 3)])
def test_gloo_distributed_merge_with_numpy_parts(tmp_path, world, lpr, direct, sigma):
    """Every merge spread over all ranks of its subtree (bwt_amd/dist_merge.py), the per-part steps done in numpy:
    the exchange pattern and the bookkeeping are what is tested; the result must be the block of the same tree.
    direct = 1 hands the tensors to the transport without host staging: receives land in slices of the destination
    buffer, broadcasts and sums work in place -- the code path of an nccl run.  Alphabets of up to four symbols send
    L as two bits per symbol (slices of any length and offset, the terminator of the first block put back); six
    symbols take the plain bytes."""
    n, ns, seg = 24000 + sigma, 300, 64
    out = str(tmp_path / "res.npz")
    _launch(world, ["--ops", "oracle", "--algo", "dist", "--text_len", str(n), "--lpr", str(lpr), "--ns", str(ns),
                    "--seg", str(seg), "--direct", str(direct), "--sigma", str(sigma)], out)
    res = np.load(out)
    T = gen_text(n, 42, sigma)
    th = multi_gpu.leaf_threshold(n, world * lpr)
    o = Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
    ob = Oracle.pmbwt(T, o)
    assert (res["L"] == ob.L).all()
    assert (int(res["s_rank"]), int(res["e_rank"])) == (ob.s_rank, ob.e_rank)
    r, p = ob.samples()
    assert (res["rank"] == r).all() and (res["pos"] == p).all()
    assert int(res["sizes"].sum()) == n and len(res["sizes"]) == world


def test_gloo_distributed_merge_with_grouped_g_exchange(tmp_path):
    """A left block of 2^21 suffixes and more: the g values travel grouped by gap bucket (16-bit words + bucket bounds,
    parts cut at multiples of 8192 left ranks; include/bwt_hip.h, bwt_hip_dist_layout / _chains_grouped /
    _merge_slice_grouped), restated in numpy with a descending order inside the buckets and a few fillers."""
    n, ns, seg, world = 4200100, 3000, 1000, 2
    out = str(tmp_path / "res.npz")
    _launch(world, ["--ops", "oracle", "--algo", "dist", "--text_len", str(n), "--lpr", "1", "--ns", str(ns), "--seg", str(seg),
                    "--direct", "1"], out, timeout=600)
    res = np.load(out)
    T = gen_text(n, 42, 4)
    th = multi_gpu.leaf_threshold(n, world)
    o = Oracle.opt(n, alpha_min=1, alpha_max=4, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
    ob = Oracle.pmbwt(T, o)
    assert (res["L"] == ob.L).all()
    assert (int(res["s_rank"]), int(res["e_rank"])) == (ob.s_rank, ob.e_rank)
    r, p = ob.samples()
    assert (res["rank"] == r).all() and (res["pos"] == p).all()
    assert res["sizes"][0] % 8192 != 0 and int(res["sizes"].sum()) == n  # (the cut is at a multiple of 8192 LEFT ranks)


def test_key_range_cuts_cover_and_balance():
    """The cut of one leaf over several ranks by first key digit: stretches of digits that cover everything once, with
    about equal counts where no single digit is heavy, and the same for every caller (it depends on the counts only)."""
    from bwt_amd import dist_leaf
    rng = np.random.default_rng(5)
    for nd, parts in ((512, 2), (512, 8), (256, 4), (16, 8)):
        counts = rng.integers(0, 1000, size=nd)
        cuts, bounds = dist_leaf.key_range_cuts(counts, parts)
        assert cuts[0] == 0 and cuts[-1] == nd and all(a <= b for a, b in zip(cuts, cuts[1:]))
        assert bounds[0] == 0 and bounds[-1] == counts.sum()
        if nd >= 256:
            sizes = np.diff(bounds)
            assert sizes.max() - sizes.min() <= 2 * counts.max()
    # one digit holds nearly everything: some parts are empty, nothing is lost
    counts = np.array([1, 2, 1000000, 3, 4])
    cuts, bounds = dist_leaf.key_range_cuts(counts, 4)
    assert bounds[-1] == counts.sum() and sorted(bounds) == bounds


@pytest.mark.parametrize("world,sigma", [(2, 4), (4, 26)])
def test_gloo_one_leaf_by_key_range(tmp_path, world, sigma):
    """The one-leaf schedule over gloo with numpy shares: slices of L in rank order and the exchanged sample slots
    equal the oracle's leaf of the whole text."""
    n, ns = 20000, 300
    out = str(tmp_path / "leaf.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(31500 + os.getpid() % 2000),
           os.path.join(ROOT, "tests", "dist_leaf_worker.py"), "--out", out, "--text_len", str(n), "--ns", str(ns), "--sigma", str(sigma)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    res = np.load(out)
    T = gen_text(n, 42, sigma)
    ob = Oracle.pmbwt(T, Oracle.opt(n, alpha_min=1, alpha_max=sigma, bwt_rec_threshold=n, ssa_n_samples=ns))
    assert (res["L"] == ob.L).all()
    r, p_ = ob.samples()
    assert (res["samp_pos"] == p_).all() and (res["samp_rank"] == r).all()
