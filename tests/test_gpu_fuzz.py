"""A short differential run over random small configurations (tools/fuzz_small.py holds the generator): text shape,
alphabet and its offset, leaf size, sample count and chain length all vary; everything is compared with the oracle."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import Oracle
from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _generator():
    spec = importlib.util.spec_from_file_location("fuzz_small", os.path.join(os.path.dirname(HERE), "tools", "fuzz_small.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.make


@pytest.mark.parametrize("seed", [11, 12])
def test_random_small_configurations(ctx, seed):
    make = _generator()
    rng = np.random.default_rng(seed)
    for k in range(60):
        T, amax, th, ns, seg, kind = make(rng)
        n = len(T)
        tag = "seed %d case %d: n=%d max_sym=%d th=%d ns=%d seg=%d %s" % (seed, k, n, amax, th, ns, seg, kind)
        ctx.set_text(T)
        blk = ctx.pmbwt(th, ns, seg)
        o = Oracle.opt(n, alpha_min=1, alpha_max=amax, bwt_rec_threshold=th, ssa_n_samples=ns, build_gap_segment_sz=seg)
        ob = Oracle.pmbwt(T, o)
        assert (blk.L == ob.L).all(), tag
        assert (blk.s_rank, blk.e_rank, blk.e_char) == (ob.s_rank, ob.e_rank, ob.e_char), tag
        for x, y in zip(blk.samples(), ob.samples()):
            assert len(x) == len(y) and (x == y).all(), tag
        blk.free()


def test_random_multi_rank_configurations(ctx):
    """The same for the C++ multi-GPU driver (tools/fuzz_multi.py holds the generator): random lengths, alphabets,
    rank counts (threads that share the GPU), leaf sizes, sample counts and chain lengths against the single-context
    build of the same tree; some texts large enough for the grouped exchange of the g values."""
    import bwt_amd
    spec = importlib.util.spec_from_file_location("fuzz_multi", os.path.join(os.path.dirname(HERE), "tools", "fuzz_multi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(5)
    for k in range(30):
        T, world, th, ns, seg = mod.make(rng)
        tag = "case %d: n=%d symbols %d..%d world=%d th=%d ns=%d seg=%d" % (k, len(T), T[:-1].min(), T[:-1].max(), world, th, ns, seg)
        ctx.set_text(T)
        one = ctx.pmbwt(th, ns, seg)
        L1, info1, (r1, p1) = one.L, (one.s_rank, one.e_rank), one.samples()
        one.free()
        L, s_rank, e_rank, r, p, used = bwt_amd.pmbwt_multi([0] * world, T, th, ns, seg)
        assert (s_rank, e_rank) == info1 and (L == L1).all(), tag
        assert len(r) == len(r1) and (r == r1).all() and (p == p1).all(), tag


def test_random_small_configurations_with_the_step_over_runs_forced():
    """tools/fuzz_small.py in a process of its own with BWT_HIP_RUN_JUMP=2: the doubling rounds then step over the runs
    of one symbol in every window (by themselves only in windows that hold a run of 4096 symbols or more), and the
    generator's skewed, run and small-alphabet texts are full of short runs."""
    import subprocess
    import sys
    env = dict(os.environ, BWT_HIP_RUN_JUMP="2")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "fuzz_small.py"), "250", "77"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz ok: 250 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
