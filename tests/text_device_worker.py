"""Run by test_gpu_sparse_alphabet.py in a process of its own (torch has to be imported before the library is loaded):
a text that already lies in device memory of the caller (bwt_hip_set_text_device), as it is and as letters, against the
oracle; the caller's memory is left as it was."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bwt_amd  # noqa: E402
from oracle import Oracle, gen_text  # noqa: E402


def main():
    ctx = bwt_amd.Context(0)
    lut = np.zeros(256, dtype=np.uint8)
    lut[1:5] = [65, 67, 71, 84]
    for name, T in (("1..4", gen_text(150000, 71, 4)), ("ACGT", lut[gen_text(150000, 72, 4)])):
        n = len(T)
        d = torch.from_numpy(T.copy()).cuda()
        assert d.data_ptr() % 16 == 0
        torch.cuda.synchronize()
        ctx.set_text_device(d.data_ptr(), n)
        blk = ctx.pmbwt(50000, 700)
        ob = Oracle.pmbwt(T, Oracle.opt(n, alpha_min=1, alpha_max=int(T.max()), bwt_rec_threshold=50000, ssa_n_samples=700))
        assert (blk.L == ob.L).all() and (blk.s_rank, blk.e_rank, blk.e_char) == (ob.s_rank, ob.e_rank, ob.e_char), name
        for x, y in zip(blk.samples(), ob.samples()):
            assert (x == y).all(), name
        assert (ctx.get_text() == T).all(), name
        assert ctx.symbol_codes()[0] == (name == "ACGT")
        ctx.sync()
        assert (d.cpu().numpy() == T).all(), name  # the caller's buffer is untouched
        blk.free()
    ctx.close()
    print("text device ok")


if __name__ == "__main__":
    main()
