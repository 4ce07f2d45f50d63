"""The host-side structures a user of the reference's headers also gets -- bwt::gap_array, bwt::succinct_bit_vector,
concurrent sampled_suffix_array::put, psort / check_sorted -- checked by the REFERENCE's own test programs
(src/test/test_gap.cc, test_succinct_bit_vector.cc, test_ssa_conc.cc, test_psort.cc), compiled unmodified and where they
lie against this repo's include/ (oracle/Makefile, target dropin).  They never touch the GPU: the MI355X path keeps
these structures inside its kernels.  The binaries exist only where the reference tree was mounted at build time."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "dropin")


@pytest.mark.parametrize("prog,args", [("test_gap", [100000, 316, 12, 20]), ("test_gap", [3000, 5, 10, 7]),
                                       ("test_succinct_bit_vector", [1000000, 60]), ("test_succinct_bit_vector", [77, 40]),
                                       ("test_ssa_conc", [10000, 20]), ("test_psort", [100000, 3])])
def test_reference_host_tests_unmodified(prog, args):
    path = os.path.join(REF, prog)
    if not os.path.exists(path):
        pytest.skip("%s not built" % os.path.relpath(path, ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="8")  # test_gap and test_ssa_conc insert from several threads at once
    p = subprocess.run([path] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=170, env=env)
    out = p.stdout.decode("latin1")
    assert p.returncode == 0 and "NG" not in out, out[-2000:]
