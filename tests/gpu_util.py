"""Shared helpers of the GPU parity tests: one context per session, reference blocks from the oracle."""
import numpy as np
import pytest


@pytest.fixture(scope="session")
def ctx():
    import bwt_amd
    c = bwt_amd.Context(0)
    yield c
    c.close()


def inverse_bwt(L, e_rank=0):
    """Text from the BWT of a whole 0-terminated text (numpy LF table + one serial walk)."""
    n = len(L)
    order = np.argsort(L, kind="stable")
    LF = np.empty(n, dtype=np.int64)
    LF[order] = np.arange(n)
    out = np.empty(n, dtype=np.uint8)
    i = e_rank
    out[n - 1] = 0
    LFl, Ll = LF.tolist(), L.tolist()
    for j in range(n - 1, 0, -1):
        c = Ll[i]
        out[j - 1] = c
        i = LFl[i]
    return out
