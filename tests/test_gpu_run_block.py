"""Texts with long runs of one symbol on the lean bucket route (the run's suffixes ranked in closed form), forced onto
small texts, against the oracle; in a process of its own because the switches are read when a context is created."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_runs_in_closed_form_against_oracle():
    env = dict(os.environ, BWT_HIP_BUCKET_MIN_N="1", BWT_HIP_FORCE_WIDE="2")
    r = subprocess.run([sys.executable, os.path.join(HERE, "run_block_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "run block ok" in r.stdout
