"""Per-kernel averages of a rocprofv3 --pmc run's counter_collection CSV:
   python tools/pmc_per_kernel.py <dir with *counter_collection.csv> [substring of the kernel names to keep]"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit("no counter_collection.csv under " + root)
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if want not in name:
                    continue
                tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[name].add(row["Dispatch_Id"])
    for name in sorted(tot, key=lambda k: -tot[k].get("SQ_BUSY_CYCLES", 0)):
        nd = len(disp[name])
        print("%s  (%d dispatches)" % (name[:110], nd))
        for cname in sorted(tot[name]):
            print("    %-28s %18.1f per dispatch" % (cname, tot[name][cname] / nd))


if __name__ == "__main__":
    main()
