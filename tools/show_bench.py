"""Prints the figures of a bench.py line that one looks at first: python tools/show_bench.py bench.log"""
import json
import sys

for line in open(sys.argv[1]):
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("ms_per_step", d["ms_per_step"], "value", d["value"], d["unit"])
    print("roofline", d["roofline"]["frac"], "host_to_host ms", d.get("host_to_host", {}).get("ms_per_step"))
    print("other_trees", {k: v["ms_per_step"] for k, v in d.get("other_trees", {}).items()})
    print({k: v["ms_per_step"] for k, v in d.get("phases", {}).items() if "ms_per_step" in v})
