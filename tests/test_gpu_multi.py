"""GPU tests of the multi-rank path on a one-GPU box: block transport through device buffers, and a two-rank
merge tree (both ranks on GPU 0, transport over gloo) against the single-process build."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gen_text
from tests.gpu_util import ctx  # noqa: F401
from tests.test_multi_gpu_cpu import _launch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_block_export_import_through_device_tensors():
    # runs in a process of its own: torch must be imported before libbwt_hip.so is loaded, so that the
    # library binds to the HIP runtime torch ships instead of bringing a second one into the process
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mgpu_worker.py"), "export_import"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=550, cwd=ROOT)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "EXPORT_IMPORT_OK" in out, out[-3000:]


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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,lpr,n,sigma", [(2, 1, 400000, 4), (4, 1, 400000, 4), (4, 2, 300001, 4), (8, 1, 250000, 4),
                                               (4, 1, 200000, 255), (2, 1, 3000, 4), (4, 1, 20000000, 4)])
def test_distributed_merge_equals_single_process(ctx, tmp_path, world, lpr, n, sigma):
    """Every merge spread over all ranks of its subtree through the bwt_hip_dist_* entry points (all ranks on GPU 0,
    exchange over gloo with host staging): same L, special ranks and samples as the single-process build."""
    from bwt_amd import multi_gpu
    ns = max(50, n // 200)
    out = str(tmp_path / "res.npz")
    _launch(world, ["--ops", "hip", "--algo", "dist", "--text_len", str(n), "--sigma", str(sigma), "--lpr", str(lpr),
                    "--ns", str(ns)], out, timeout=800)
    res = np.load(out)
    ctx.set_text(gen_text(n, 42, sigma))
    ref = ctx.pmbwt(multi_gpu.leaf_threshold(n, world * lpr), ns)
    assert (res["L"] == ref.L).all()
    assert (int(res["s_rank"]), int(res["e_rank"])) == (ref.s_rank, ref.e_rank)
    r, p = ref.samples()
    assert (res["rank"] == r).all() and (res["pos"] == p).all()
    assert int(res["sizes"].sum()) == n and len(res["sizes"]) == world


@pytest.mark.parametrize("algo", ["native", "dist"])
def test_bench_multi_gpu_leg_on_one_rank(algo):
    """bench.py's N > 1 leg, rehearsed with one rank: the JSON line, the schedule it names, and its own check of the
    result against a build on the rank's GPU alone."""
    import json
    env = dict(os.environ, BWT_BENCH_FORCE_DIST="1", BWT_MGPU_ALGO=algo, MASTER_PORT="29531")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "3000001", "--leaves", "4", "--steps", "1",
                        "--warmup", "1", "--cpu_n", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT,
                       env=env, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["schedule"] == algo and line["native_transport_fallback"] is None
    v = line["verified_against_one_gpu"]
    assert v["L_slices_equal"] and v["samples_equal"] and v["special_ranks_equal"], v
    assert v["ranks_with_errors"] == 0 and v["error"] is None and v["n_samples"] > 0
    assert line["n_gpus"] == 1 and line["config"]["leaves"] == 4 and line["value"] > 0


@pytest.mark.parametrize("base,sigma", [(1, 4), (100, 3), (252, 4), (253, 2), (253, 3), (254, 2)])
def test_wire_codec_with_the_terminator_inside(ctx, base, sigma):
    """Two bits per symbol between GPUs: every symbol comes back as it went, at any offset and length, also next to
    the terminator (which has no code of its own and is put back by the receiver) and for alphabets that end at 255,
    where symbol + 3 does not fit a byte.  Device buffers: the L arrays of uploaded blocks."""
    from bwt_amd import dist_merge
    n = 70000
    T = gen_text(n, 13, sigma)
    T[:-1] += base - 1
    ctx.set_text(T)
    assert ctx.dist_wire_packable()
    two = np.array([0, 1], dtype=np.int64)

    def device_bytes(a):
        return ctx.upload_block(0, len(a), a, 0, 0, two, two)

    for count, off, z in ((n - 500, 0, 17), (n - 501, 0, 31), (4099, 3, 4098), (4099, 5, 0), (50001, 1, 25000), (15, 0, 7), (33, 2, 16), (16, 0, 15)):
        src = T[off:off + count].copy()
        src[z] = 0
        s = device_bytes(np.concatenate([np.zeros(off, dtype=np.uint8), src]))
        packed = device_bytes(np.zeros(dist_merge.wire_bytes(count) + 64, dtype=np.uint8))
        out = device_bytes(np.full(off + count + 64, 77, dtype=np.uint8))
        ctx.dist_wire_pack(s.L_device + off, count, packed.L_device)
        ctx.dist_wire_unpack(packed.L_device, count, out.L_device + off)
        ctx.sync()
        got = out.L
        for blk in (s, packed, out):
            blk.free()
        assert (got[:off] == 77).all() and (got[off + count:] == 77).all()
        got = got[off:off + count]
        keep = np.arange(count) != z
        assert (got[keep] == src[keep]).all(), (base, sigma, count, off, z, np.nonzero(got != src)[0][:8])


@pytest.mark.parametrize("base,sigma", [(253, 2), (252, 4), (254, 2), (253, 3)])
def test_native_multi_alphabets_that_end_at_255(base, sigma):
    """Merges whose left block is spread over several ranks, on alphabets at the top of the byte range."""
    import bwt_amd
    from oracle import Oracle
    n = 8438
    T = gen_text(n, 21, sigma)
    T[:-1] += base - 1
    for world, th in ((4, 2110), (8, 1055)):
        L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * world, T, th, 40, 7)
        o = Oracle.opt(n, alpha_min=base, alpha_max=base + sigma - 1, bwt_rec_threshold=th, ssa_n_samples=40, build_gap_segment_sz=7)
        ob = Oracle.pmbwt(T, o)
        assert (L == ob.L).all() and (s_rank, e_rank) == (ob.s_rank, ob.e_rank)
        orr, op = ob.samples()
        assert (r == orr).all() and (p == op).all()
