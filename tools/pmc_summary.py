#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run (the *_counter_collection.csv it writes).

    python tools/pmc_summary.py gpurun_out/pmc/*_counter_collection.csv [substring-of-kernel-name ...]
"""
import csv
import sys
from collections import defaultdict


def main():
    path, filters = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if filters and not any(s in name for s in filters):
                continue
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name].add(row["Dispatch_Id"])
    for name in sorted(acc, key=lambda k: -sum(acc[k].values())):
        n = len(calls[name])
        print("%s  (%d dispatches)" % (name[:110], n))
        for cname, v in sorted(acc[name].items()):
            print("    %-28s %16.1f per dispatch" % (cname, v / n))


if __name__ == "__main__":
    main()
