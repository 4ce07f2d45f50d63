"""bwt_hip_dist_merge_slice_noalloc (scratch in the library's static device array) against
bwt_hip_dist_merge_slice (scratch from the pool) and against a numpy restatement of the merge loop
(reference include/bwt.h:815-829): two calls in a row, the second smaller, through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


class Hip:
    """Plain device buffers through the HIP runtime the library itself links (no torch in this process)."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.live = []

    def up(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(a.nbytes, 16)) == 0
        assert self.rt.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
        self.live.append(p)
        return p.value

    def empty(self, nbytes):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        self.live.append(p)
        return p.value

    def down(self, ptr, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2) == 0
        return out

    def free_all(self):
        for p in self.live:
            self.rt.hipFree(p)
        self.live = []


def expected(lL, rL, g):
    m, rn = len(lL), len(rL)
    S = np.cumsum(np.bincount(g, minlength=m + 1)).astype(np.uint32)  # S[i] = #{g <= i}
    out = np.empty(m + rn, dtype=np.uint8)
    left_at = np.arange(m, dtype=np.int64) + S[:m]
    is_left = np.zeros(m + rn, dtype=bool)
    is_left[left_at] = True
    out[is_left] = lL
    out[~is_left] = rL
    return out, S


def one_call(ctx, hip, mL, rn, seed):
    rng = np.random.RandomState(seed)
    lL = rng.randint(1, 5, size=mL).astype(np.uint8)
    rL = rng.randint(1, 5, size=rn).astype(np.uint8)
    g = rng.randint(0, mL + 1, size=rn).astype(np.uint32)
    if rn > 10:
        g[: rn // 3] = g[0]  # one heavy gap
    want_out, want_S = expected(lL, rL, g)
    d_l, d_r = hip.up(lL), hip.up(rL)
    res = {}
    for name, fn in (("pool", ctx.dist_merge_slice), ("static", ctx.dist_merge_slice_noalloc)):
        d_g = hip.up(g)  # destroyed by the call
        d_out, d_S = hip.empty(mL + rn), hip.empty(4 * (mL + 1))
        before = ctx.mem_stats()
        fn(d_l, mL, d_r, rn, d_g, d_out, d_S)
        after = ctx.mem_stats()
        if name == "static":  # nothing came from the pool or the driver
            assert after["reserved"] == before["reserved"] and after["peak"] == before["peak"]
        res[name] = (hip.down(d_out, np.uint8, mL + rn), hip.down(d_S, np.uint32, mL + 1))
    hip.free_all()
    for name in res:
        assert (res[name][1] == want_S).all(), name
        assert (res[name][0] == want_out).all(), name
    assert (res["pool"][0] == res["static"][0]).all() and (res["pool"][1] == res["static"][1]).all()


def test_no_alloc_merge_slice_twice(ctx):
    hip = Hip()
    one_call(ctx, hip, 3000000, 4194304, 1)  # the maximum right count
    one_call(ctx, hip, 1000, 777, 2)         # smaller: the array still holds the first call's data
    one_call(ctx, hip, 70000, 0, 3)
    one_call(ctx, hip, 0, 5000, 4)


@pytest.mark.timeout(600)
def test_no_alloc_largest_layout(ctx):
    """The maximum right count next to a left part of more than 2^30 symbols: the grouping with the most buckets and
    tiles, i.e. the most scratch the call can need.  Static against pool, and S against numpy."""
    hip = Hip()
    mL, rn = (1 << 30) + 12345, 4194304
    rng = np.random.RandomState(9)
    lL = rng.randint(1, 5, size=mL, dtype=np.uint8)
    rL = rng.randint(1, 5, size=rn).astype(np.uint8)
    g = rng.randint(0, mL + 1, size=rn).astype(np.uint32)
    g[:1000] = mL  # the last gap index
    g[1000:2000] = 0
    want_S = np.cumsum(np.bincount(g, minlength=mL + 1), dtype=np.uint32)
    d_l, d_r = hip.up(lL), hip.up(rL)
    res = {}
    for name, fn in (("pool", ctx.dist_merge_slice), ("static", ctx.dist_merge_slice_noalloc)):
        d_g = hip.up(g)
        d_out, d_S = hip.empty(mL + rn), hip.empty(4 * (mL + 1))
        before = ctx.mem_stats()
        fn(d_l, mL, d_r, rn, d_g, d_out, d_S)
        after = ctx.mem_stats()
        if name == "static":
            assert after["reserved"] == before["reserved"] and after["peak"] == before["peak"]
        res[name] = (hip.down(d_out, np.uint8, mL + rn), hip.down(d_S, np.uint32, mL + 1))
    hip.free_all()
    assert (res["static"][1] == want_S).all() and (res["pool"][1] == want_S).all()
    assert (res["pool"][0] == res["static"][0]).all()
    out = res["static"][0]
    probe = rng.randint(0, mL, size=100000)  # left symbol i sits at i + S[i]
    assert (out[probe + want_S[probe].astype(np.int64)] == lL[probe]).all()
    assert (np.bincount(out, minlength=5) == np.bincount(lL, minlength=5) + np.bincount(rL, minlength=5)).all()


def test_no_alloc_rejects_oversize(ctx):
    import bwt_amd
    hip = Hip()
    rn = 4194304 + 1
    d = hip.empty(4 * rn)
    with pytest.raises(bwt_amd.BwtHipError) as e:
        ctx.dist_merge_slice_noalloc(d, 10, d, rn, d, d, d)
    assert "exceeds" in str(e.value)
    hip.free_all()
