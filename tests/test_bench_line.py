"""The bench line's `roofline` and `phases` objects are built from the library's timers by bwt_amd/benchutil.py:
checked here on made-up timers (no GPU)."""
import json
import os

from bwt_amd import benchutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _timers():
    # name -> (ms summed over launches, work summed over launches, launches)
    return {
        "gap_chains": (21.0, 2 * 5.3125e8, 2),                      # sectors
        "leaf:k_bucket_emit": (14.0, 4 * 4.5e9, 4),                  # bytes
        "leaf:k_rs_scatter_u64_keys": (6.8, 4 * 8.0e9, 4),
        "leaf:k_rs_scatter_text": (6.9, 4 * 4.5e9, 4),
        "merge_scatter": (2.5, 2 * 4.0e9, 2),
    }


def test_roofline_describes_the_kernel_with_the_largest_total_time():
    r = benchutil.roofline(_timers())
    assert r["kernel"].startswith("k_chains") and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert r["access"] == "random 64-byte sectors" and r["launches"] == 2
    achieved = 2 * 5.3125e8 * 64 / 21.0e-3 / 1e9
    assert abs(r["achieved"] - achieved) < 0.1 and abs(r["frac"] - achieved / 8000.0) < 1e-3
    assert r["bytes_per_launch"] == 5.3125e8 * 64 and r["traffic"] is None
    assert [o["kernel"].split(" ")[0] for o in r["others"]] == ["k_bucket_emit", "k_rs_scatter<text>", "k_rs_scatter<u64,keys>"]
    t = _timers()
    del t["gap_chains"]
    assert benchutil.roofline(t)["kernel"].startswith("k_bucket_emit")
    assert benchutil.roofline({"merge_scatter": (1.0, 1.0, 1)}) is None


def test_roofline_traffic_comes_from_the_committed_counter_summary():
    tfile = os.path.join(ROOT, "profiles", "r1_traffic.json")
    t = json.load(open(tfile))
    r = benchutil.roofline(_timers(), ROOT, t["workload"])
    assert r["traffic"] == t["kernels"]["gap_chains"]["hbm_bytes_per_launch"]
    # per launch the counters must not be far from the algorithmic bytes (reads of the rank blocks + g written)
    assert 0.9 < r["traffic"] / (5e8 * (1.0625 * 64 + 4)) < 1.2
    assert benchutil.roofline(_timers(), ROOT, {"n": 1, "leaves": 1})["traffic"] is None


def test_phase_table_units():
    p = benchutil.phase_table(_timers(), steps=2)
    assert p["gap_chains"]["rate_unit"] == "Gtransactions/s" and p["merge_scatter"]["rate_unit"] == "GB/s"
    assert abs(p["gap_chains"]["ms_per_step"] - 10.5) < 1e-9
    assert list(p)[0] == "gap_chains"


def test_no_phase_of_a_recorded_bench_line_exceeds_the_peak():
    """Every bench line kept under profiles/r2_bench*.json prices its phases against the 8 TB/s peak: a fraction above 1
    means the byte model is not what the kernels do."""
    import glob
    seen = 0
    for path in glob.glob(os.path.join(ROOT, "profiles", "r2_bench*.json")):
        line = json.load(open(path))
        line = line.get("parsed", line)
        for name, row in (line.get("phases") or {}).items():
            assert row.get("frac_of_peak", 0) <= 1.0, (path, name, row)
            seen += 1
        assert line["roofline"]["frac"] <= 1.0
    assert seen > 0


def test_bench_rejects_leaf_counts_that_do_not_fit_the_gpus():
    import subprocess
    import sys
    for leaves in ("3", "0", "6"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--leaves", leaves], capture_output=True, text=True)
        assert r.returncode == 2 and "power of two" in r.stderr
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--leaves", "2"], capture_output=True,
                       text=True, env=env)
    assert r.returncode == 2 and "multiple of the 4 GPUs" in r.stderr


def test_bench_line_says_what_value_measures():
    """`value` is the HBM-resident rate with the library's timers off; the host-to-host rate of bwt::pmbwt's own
    call travels in the same line under `host_to_host`, over the same number of steps."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = bench.VALUE_DEFINITION
    assert "resident in HBM" in d and "timers off" in d and "host_to_host" in d and "bwt_hip_pmbwt_host" in d
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "host_to_host(ctx, n, th, ns, args.seg, args.steps)" in src
    assert src.index("ctx.set_timing(0)\n    dt = timed_builds") < src.index("ctx.set_timing(2)")
