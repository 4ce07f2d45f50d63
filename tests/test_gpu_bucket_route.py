"""The leaf sort's bucket route (two passes on the top 16 key bits, the rest per bucket in LDS) only pays on windows of
tens of millions of suffixes; here its threshold is lowered so that small texts take it, in a process of its own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_bucket_route_on_small_texts():
    env = dict(os.environ, BWT_HIP_BUCKET_MIN_N="1", BWT_HIP_FORCE_WIDE="3")  # 3: not the lean route, which comes first otherwise
    r = subprocess.run([sys.executable, os.path.join(HERE, "bucket_route_worker.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "bucket route ok" in r.stdout


def test_wide_bucket_route_on_small_texts():
    """The route of windows above 2^29 suffixes (8 + 1 byte elements, three passes, 2^17 and more buckets), forced
    onto small texts."""
    env = dict(os.environ, BWT_HIP_BUCKET_MIN_N="1", BWT_HIP_FORCE_WIDE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "wide_route_worker.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "wide route ok" in r.stdout


@pytest.mark.parametrize("slots", ["1", "3", "2", "4", "5"])
def test_lean_bucket_route_on_small_texts(slots):
    """The route of windows between 2^29 and 2^31 suffixes (8-byte elements without the first pass's digit, two
    passes, 2^16 to 2^18 buckets), forced onto small texts.  slots: the second pass as it chooses itself (offsets
    from a histogram on small texts, a fixed number of slots per bucket on large ones), with slots on every text, and
    with too few slots in the second pass or in the first (the pass notices, and runs again with offsets)."""
    env = dict(os.environ, BWT_HIP_BUCKET_MIN_N="1", BWT_HIP_FORCE_WIDE="2", BWT_HIP_LEAN_SLOTS=slots)
    r = subprocess.run([sys.executable, os.path.join(HERE, "lean_route_worker.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "lean route ok" in r.stdout
