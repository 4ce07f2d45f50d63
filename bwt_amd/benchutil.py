"""Turns the library's phase and kernel timers into the bench line's `roofline` and `phases` objects."""
import json
import os

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec sheet); a plain copy kernel reaches about 6300 GB/s
SECTOR = 64            # bytes the memory side moves for one random rank-block read

# the leaf sort's scatter pass, in whichever layout the run used: (timer, kernel, algorithmic bytes per suffix)
DOMINANT = [("leaf:k_rs_scatter_u64_keys", "k_rs_scatter<u64,keys> (packed 8-byte elements)", 16),
            ("leaf:k_rs_scatter_u64_pairs", "k_rs_scatter<u64,pairs,split>", 24)]

# what a timer's `work` counts, for timers that do not count bytes
TRANSACTION_TIMERS = {"gap_chains": "64-byte random reads (one rank block per wavelet level and LF step, text every 16th step)"}


def roofline(timers, repo_root=None, workload=None):
    """HBM roofline of the dominant kernel: the leaf sort's scatter pass, which reads and writes every element
    once (16 algorithmic bytes per suffix with packed elements, 24 with key + position pairs).  `work` of its
    timer is bytes summed over launches."""
    found = [d for d in DOMINANT if d[0] in timers and timers[d[0]][2] > 0]
    if not found:
        return None
    name, kernel, per_suffix = max(found, key=lambda d: timers[d[0]][0])
    ms, work, calls = timers[name]
    achieved = work / (ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kernel,
            "algorithmic_bytes_per_suffix": per_suffix,
            "bytes_per_launch": work / calls, "avg_launch_ms": round(ms / calls, 4), "launches": calls}
    # HBM bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE and WRITE_SIZE in separate rocprofv3
    # runs, corrected as the gfx950 notes prescribe); only valid for the workload they were taken on
    if repo_root and workload:
        tfile = os.path.join(repo_root, "profiles", "r1_traffic_rs_scatter.json")
        if os.path.exists(tfile):
            t = json.load(open(tfile))
            if t.get("workload") == workload and t.get("timer", "leaf:k_rs_scatter_u64_pairs") == name:
                roof["traffic"] = t["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/r1_traffic_rs_scatter.json"
    return roof


def phase_table(timers, steps):
    """Per phase: ms per step, achieved rate against the HBM peak (bytes, or 64-byte transactions for the LF chains)."""
    out = {}
    for k, v in sorted(timers.items(), key=lambda kv: -kv[1][0]):
        ms, work, calls = v
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
