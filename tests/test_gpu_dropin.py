"""GPU tests of the drop-in boundary: C++ programs written against the bwt:: header API.

tools/bin/* are this repo's own programs (tools/Makefile).  oracle/_ref/dropin/* are the REFERENCE's own
programs (src/simplest_example.cc, src/pmbwt.cc, src/test/test_*.cc) compiled, unmodified and where they lie,
against this repo's include/ and linked to libbwt_hip.so (oracle/Makefile, target dropin); they exist only
where the reference tree was mounted at build time and are skipped otherwise.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OWN = os.path.join(ROOT, "tools", "bin")
REF = os.path.join(ROOT, "oracle", "_ref", "dropin")

pytestmark = pytest.mark.gpu


def run(path, *args, timeout=170):
    p = subprocess.run([path] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return p.returncode, p.stdout.decode("latin1")


def need(path):
    if not os.path.exists(path):
        pytest.skip("%s not built" % os.path.relpath(path, ROOT))
    return path


def test_own_dropin_check():
    rc, out = run(need(os.path.join(OWN, "dropin_check")))
    assert rc == 0 and "ALL OK" in out, out[-2000:]


def test_own_simplest_example():
    rc, out = run(need(os.path.join(OWN, "simplest_example")))
    assert rc == 0 and "bwt:   ard$rcaaaabb" in out, out


@pytest.mark.parametrize("args", [
    ["100000"],                                                           # the reference CLI's defaults: 'a'..'z'
    ["--alpha_min", "a", "--alpha_max", "d", "--algo_alpha", "1", "-m", "3.0", "2000000"],
    ["--generator", "nrand48", "--alpha_max", "z", "-t", "5000", "--ssa_n_samples", "300", "--build_gap_segment_sz", "77", "200000"],
    ["--stat_level", "1", "--mstat_level", "1", "--repeat", "2", "50000"],
    ["--check", "2", "--alpha_min", "a", "--alpha_max", "d", "--algo_alpha", "1", "300000"],  # the host's serial ibwt
    # 2e8 symbols: host T -> device, 4 leaves, round trip decoded on the device
    ["--alpha_min", "a", "--alpha_max", "d", "--algo_alpha", "1", "-t", "50000000", "200000000"],
])
def test_own_cli_round_trip(args):
    rc, out = run(need(os.path.join(OWN, "pmbwt")), *args)
    assert rc == 0 and "checking result ... OK" in out and "NG" not in out, out[-2000:]
    assert "clocks to build bwt for" in out  # the line the reference prints, include/bwt.h:881


def test_reference_simplest_example_unmodified():
    rc, out = run(need(os.path.join(REF, "simplest_example")))
    assert rc == 0 and "ipssm" in out and "pissii" in out, out


@pytest.mark.parametrize("args", [["100000"], ["--check", "1", "-m", "3", "1000000"], ["--stat_level", "1", "30000"]])
def test_reference_cli_unmodified(args):
    rc, out = run(need(os.path.join(REF, "pmbwt")), *args)
    assert rc == 0 and "checking result ... OK" in out, out[-2000:]


@pytest.mark.parametrize("prog,args", [("test_leaf", [20000, 5]), ("test_merge", [20000, 5]), ("test_ssa", [3000, 3]),
                                       ("test_rec", [20000, 3]), ("test_suffix_array", []), ("test_wavelet_matrix", [])])
def test_reference_tests_unmodified(prog, args):
    rc, out = run(need(os.path.join(REF, prog)), *args)
    assert rc == 0 and "NG" not in out, out[-2000:]


def test_stat_table_credits_every_device_phase():
    """pmbwt with stat_level 1 prints the reference's event table (include/util.h:93-98, 102-135 there): the rows the
    reference fills on this path must be there with the right call counts -- 4 leaves, 3 merges -- and non-zero time."""
    import re
    rc, out = run(need(os.path.join(OWN, "pmbwt")), "--alpha_min", "a", "--alpha_max", "d", "--algo_alpha", "1",
                  "--stat_level", "1", "-t", "60000", "--check", "0", "200000")
    assert rc == 0, out[-2000:]
    rows = {m.group(1): (int(m.group(2)), int(m.group(3)))
            for m in re.finditer(r"^\s*(\w+) :\s*(\d+) /\s*(\d+) =", out, re.M)}
    want = {"bwt_leaf": 4, "bwt_sample_sa": 4, "bwt_count_alphabets": 3, "bwt_build_wavelet_matrix": 3, "bwt_build_gap": 3,
            "gap_array_set_prefix_sum": 3, "bwt_merge_loop": 3, "bwt_resample_sa": 3, "pmbwt": 1, "ssa_put": 1}
    for ev, calls in want.items():
        assert ev in rows, (ev, sorted(rows))
        assert rows[ev][1] == calls and rows[ev][0] > 0, (ev, rows[ev])


def test_cli_over_several_ranks_that_share_the_gpu():
    """bwt_opt::n_gpus through the CLI: four ranks (threads, one context each; BWT_HIP_DEVICES puts them on the one
    GPU at hand), host T in, host L and samples out, round trip by the host's ibwt (the blocks died with the ranks)."""
    env = dict(os.environ, BWT_HIP_DEVICES="0,0,0,0")
    r = subprocess.run([need(os.path.join(OWN, "pmbwt")), "--alpha_min", "a", "--alpha_max", "d", "--algo_alpha", "1", "--n_gpus", "4",
                        "-t", "30000", "400000"], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "checking result ... OK" in out, out[-2000:]
