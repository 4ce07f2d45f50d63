"""GPU unit tests of the device primitives behind the hot path: radix sort passes and scans."""
import numpy as np
import pytest

from tests.gpu_util import ctx  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 4097, 100000, 1 << 20, (1 << 22) + 12345])
def test_scan_exclusive_sum(ctx, n):
    rng = np.random.RandomState(n % 1000)
    d = rng.randint(0, 1000, size=n).astype(np.uint32)
    got = ctx.test_scan(d, 0)
    exp = (np.cumsum(d, dtype=np.uint64) - d).astype(np.uint32)
    assert (got == exp).all()


@pytest.mark.parametrize("n", [1, 100, 4096, 16385, 3000000])
def test_scan_inclusive_max(ctx, n):
    rng = np.random.RandomState(n % 1000 + 1)
    d = (rng.randint(0, 1 << 30, size=n) * (rng.rand(n) < 0.01)).astype(np.uint32)
    got = ctx.test_scan(d, 1)
    assert (got == np.maximum.accumulate(d)).all()


@pytest.mark.parametrize("n", [1, 2, 64, 1000, 4096, 4097, 70001, 1 << 20, 3333333])
@pytest.mark.parametrize("bits", [(0, 64), (1, 64), (0, 35), (8, 24)])
def test_sort_pairs_u64_stable(ctx, n, bits):
    rng = np.random.RandomState(n % 997 + bits[1])
    keys = rng.randint(0, 1 << 62, size=n, dtype=np.int64).astype(np.uint64) * np.uint64(3)
    if n > 100:
        keys[: n // 2] &= np.uint64(0xFFFF0000FFFF)  # many equal digits
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = ctx.test_sort(keys, vals, bits[0], bits[1])
    width = bits[1] - bits[0]
    mask = np.uint64((1 << width) - 1) if width < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    sub = (keys >> np.uint64(bits[0])) & mask
    order = np.argsort(sub, kind="stable")
    assert (gv == vals[order]).all()
    assert (gk == keys[order]).all()


@pytest.mark.parametrize("n", [5, 4096, 100001, 2500000])
def test_sort_keys_u32(ctx, n):
    rng = np.random.RandomState(n % 991)
    keys = rng.randint(0, 1 << 29, size=n).astype(np.uint32)
    gk, _ = ctx.test_sort(keys, None, 0, 29)
    assert (gk == np.sort(keys)).all()
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = ctx.test_sort(keys, vals, 0, 32)
    order = np.argsort(keys, kind="stable")
    assert (gk == keys[order]).all() and (gv == vals[order]).all()


def test_sort_skewed_digits(ctx):
    n = 300000
    keys = np.zeros(n, dtype=np.uint64)
    keys[::7] = np.uint64(1) << np.uint64(40)
    keys[::1013] = np.uint64(0xFFFFFFFFFFFFFFFF)
    vals = np.arange(n, dtype=np.uint32)
    gk, gv = ctx.test_sort(keys, vals)
    order = np.argsort(keys, kind="stable")
    assert (gk == keys[order]).all() and (gv == vals[order]).all()


@pytest.mark.parametrize("n,lo,hi", [(7, 13, 31), (8192, 13, 30), (100001, 14, 32), (3000000, 13, 31), (3000000, 0, 17)])
def test_sort_keys_u32_on_17_or_18_bits(ctx, n, lo, hi):
    """The gap histogram's sort on the high bits of g (two passes of 9-bit digits, the first in any order): the result
    is ordered on those bits and is a permutation of the input."""
    rng = np.random.RandomState(n % 977 + lo)
    keys = rng.randint(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    keys[:: 5] &= np.uint32(0x0000FFFF)  # a crowded digit
    gk, _ = ctx.test_sort(keys, None, lo, hi)
    field = (gk.astype(np.uint64) >> np.uint64(lo)) & np.uint64((1 << (hi - lo)) - 1)
    assert (np.diff(field.astype(np.int64)) >= 0).all()
    assert (np.sort(gk) == np.sort(keys)).all()
