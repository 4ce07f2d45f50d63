"""One leaf over several GPUs, cut by key range (bwt_hip_dist_pmbwt with bwt_rec_threshold >= n): the ranks are threads
that share the one GPU at hand; small texts are let onto the lean bucket route that the split needs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_one_leaf_over_ranks_by_key_range():
    env = dict(os.environ, BWT_HIP_BUCKET_MIN_N="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "multi_leaf_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "multi leaf ok" in r.stdout
