#!/usr/bin/env python3
"""Condenses the output of tools/profile_round.sh into the files kept under profiles/:
   <tag>_kernel_stats.csv, <tag>_pmc_traffic.txt, <tag>_traffic.json, <tag>_bench_under_rocprof.json"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def find(d, pat):
    m = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return m[0] if m else None


def per_kernel(path, counter):
    acc, calls = defaultdict(float), defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            acc[row["Kernel_Name"]] += float(row["Counter_Value"])
            calls[row["Kernel_Name"]].add(row["Dispatch_Id"])
    return {k: (acc[k] / len(calls[k]), len(calls[k])) for k in acc}


def main():
    out, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "profiles_" + tag)
    os.makedirs(dst, exist_ok=True)
    stats = find(os.path.join(out, "kt"), "*kernel_stats.csv")
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(out, "bench_under_rocprof.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
    cfg = json.load(open(os.path.join(out, "bench_under_rocprof.json")))["config"]
    workload = {"n": cfg["n"], "leaves": cfg["leaves"]}
    fetch = per_kernel(find(os.path.join(out, "fetch"), "*counter_collection.csv"), "FETCH_SIZE")
    write = per_kernel(find(os.path.join(out, "write"), "*counter_collection.csv"), "WRITE_SIZE")
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0 --cpu_n 0 --no-extras --no-timers (n=%d, %d leaves); per call = average over the kernel's launches" % (workload["n"], workload["leaves"]),
             "values are KB as reported, shown as MB per call; FETCH_SIZE counts a 128-byte request of a coalesced streaming read as 64 bytes on gfx950",
             "%-100s %6s %14s %14s" % ("kernel", "calls", "FETCH MB/call", "WRITE MB/call")]
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        lines.append("%-100s %6d %14.1f %14.1f" % (k[:100], fetch[k][1], fetch[k][0] / 1e3, write.get(k, (0, 0))[0] / 1e3))
    open(os.path.join(dst, tag + "_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")
    # per timer of bench.py's roofline candidates: the kernel's traffic per launch.  Streaming kernels: FETCH_SIZE x2
    # (gfx950 counts a 128-byte request of a coalesced streaming read as 64 bytes).  k_chains reads random 64-byte
    # sectors, which the counter reports as they are (checked against the known count: one sector per LF step).
    pats = [("leaf:k_rs_scatter_u64_keys", "k_rs_scatter<unsigned long, false, 0, false, false,", 2),
            ("leaf:k_rs_scatter_u64_pairs", "k_rs_scatter<unsigned long, true, 0, false, true,", 2),
            ("gap_chains", "bwtamd::k_chains(", 1)]
    kernels = {}
    for timer, pat, mult in pats:
        for k in fetch:
            if pat in k:
                f_kb, n_launch = fetch[k]
                w_kb = write.get(k, (0, 0))[0]
                kernels[timer] = {"kernel": k.split("(")[0], "launches": n_launch, "FETCH_SIZE_KB_per_launch": f_kb,
                                  "WRITE_SIZE_KB_per_launch": w_kb,
                                  "correction": "FETCH_SIZE x%d; WRITE_SIZE as is" % mult,
                                  "hbm_bytes_per_launch": (mult * f_kb + w_kb) * 1024}
                break
    json.dump({
        "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --steps 1 --warmup 0 --cpu_n 0 --no-extras --no-timers",
        "workload": workload, "kernels": kernels,
    }, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, tag + "_traffic.json")).read())
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
