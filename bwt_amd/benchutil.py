"""Turns the library's phase and kernel timers into the bench line's `roofline` and `phases` objects."""
import json
import os

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec sheet); a plain copy kernel reaches about 6300 GB/s
SECTOR = 64            # bytes the memory side moves for one random rank-block read
# isolated random 64-byte reads per second this chip serves, whatever the kernel around them does
# (tools/random_sector_rate.hip, results under profiles/): the ceiling of the LF chains' access shape
RANDOM_READS_PER_S = 5.6e10

# Kernels that can dominate a build: (timer, kernel, what the timer's `work` counts, algorithmic unit).
#   bytes:   `work` is algorithmic bytes summed over launches
#   sectors: `work` is random 64-byte reads summed over launches (one rank block per wavelet level and LF step,
#            one text sector every 16th step); achieved bytes = sectors * 64
CANDIDATES = [
    ("gap_chains", "k_chains (LF-mapping chains of the gap array)", "sectors",
     "read: 64 B per wavelet level and right-block position (one rank block) + 4 B every 16th position (text); written, not counted in `achieved`: 4 B per position"),
    ("leaf:k_bucket_emit", "k_bucket_emit (rank a bucket in LDS, order ties by text comparison, emit the leaf)", "bytes",
     "8 B per window suffix read + 1 B per leaf suffix written"),
    ("leaf:k_rs_scatter_u64_keys", "k_rs_scatter<u64,keys> (packed 8-byte elements)", "bytes", "16 B per window suffix"),
    ("leaf:k_rs_scatter_u64_pairs", "k_rs_scatter<u64,pairs,split>", "bytes", "24 B per window suffix"),
    ("leaf:k_rs_scatter_text", "k_rs_scatter<text> (first pass: keys rolled from the text)", "bytes",
     "1 B read + 8 or 12 B written per window suffix"),
]

# what a timer's `work` counts, for timers that do not count bytes
TRANSACTION_TIMERS = {"gap_chains": "64-byte random reads (one rank block per wavelet level and LF step, text every 16th step)"}


def _rate(timers, name, kind):
    ms, work, calls = timers[name]
    return work * (SECTOR if kind == "sectors" else 1) / (ms * 1e-3) / 1e9


def roofline(timers, repo_root=None, workload=None):
    """HBM roofline of the dominant kernel = the candidate with the largest total time in this run.  Every candidate
    is bound by HBM: streaming bytes for the sort and emit kernels, random 64-byte sectors for the LF chains.
    `others` lists the remaining candidates' rates."""
    found = [d for d in CANDIDATES if d[0] in timers and timers[d[0]][2] > 0 and timers[d[0]][0] > 0]
    if not found:
        return None
    found.sort(key=lambda d: -timers[d[0]][0])
    name, kernel, kind, unit = found[0]
    ms, work, calls = timers[name]
    achieved = _rate(timers, name, kind)
    per_launch = work * (SECTOR if kind == "sectors" else 1) / calls
    roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel,
            "access": "random 64-byte sectors" if kind == "sectors" else "streaming",
            "algorithmic_unit": unit, "bytes_per_launch": per_launch, "avg_launch_ms": round(ms / calls, 4),
            "launches": calls, "share_of_timed_kernels": None}
    # HBM bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE and WRITE_SIZE in separate rocprofv3
    # runs, corrected per kernel as the file says); only valid for the workload they were taken on
    if repo_root and workload:
        import glob
        for tfile in sorted(glob.glob(os.path.join(repo_root, "profiles", "r*_traffic*.json")), reverse=True):
            t = json.load(open(tfile))
            if t.get("workload") == workload and name in t.get("kernels", {}):
                roof["traffic"] = t["kernels"][name]["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/" + os.path.basename(tfile)
                break
    if kind == "sectors":
        roof["random_read_ceiling"] = {"reads_per_s": RANDOM_READS_PER_S, "achieved_reads_per_s": round(work / (ms * 1e-3), 1),
                                       "frac": round(work / (ms * 1e-3) / RANDOM_READS_PER_S, 4),
                                       "source": "profiles/r1_random_sector_rate.txt (tools/random_sector_rate.hip)"}
    roof["others"] = [{"kernel": d[1], "ms_total": round(timers[d[0]][0], 3), "launches": timers[d[0]][2],
                       "achieved": round(_rate(timers, d[0], d[2]), 1),
                       "frac": round(_rate(timers, d[0], d[2]) / HBM_PEAK_GBS, 4)} for d in found[1:]]
    del roof["share_of_timed_kernels"]
    return roof


def phase_table(timers, steps):
    """Per phase: ms per step, achieved rate against the HBM peak (bytes, or 64-byte transactions for the LF chains)."""
    out = {}
    for k, v in sorted(timers.items(), key=lambda kv: -kv[1][0]):
        ms, work, calls = v
        if "=" in k:  # launch counters (which kernel variant ran), not phases
            out[k] = {"calls": calls}
            continue
        row = {"ms_per_step": round(ms / steps, 3), "calls": calls}
        if ms > 0 and work > 0:
            if k in TRANSACTION_TIMERS:
                tx = work / (ms * 1e-3)
                row["rate"] = round(tx / 1e9, 2)
                row["rate_unit"] = "Gtransactions/s"
                row["frac_of_peak"] = round(tx * SECTOR / 1e9 / HBM_PEAK_GBS, 4)
                row["counts"] = TRANSACTION_TIMERS[k]
            else:
                gbs = work / (ms * 1e-3) / 1e9
                row["rate"] = round(gbs, 1)
                row["rate_unit"] = "GB/s"
                row["frac_of_peak"] = round(gbs / HBM_PEAK_GBS, 4)
        out[k] = row
    return out
