"""Sums rocprofv3 counter CSVs per kernel: python benchmarks/summarize_pmc.py gpurun_out/<tag>"""
import csv
import re
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for path in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    sums = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            m = re.search(r"(trace_kernel<[^>]*>|resolve_kernel|wf_[a-z_]+(<[^>]*>)?|closest_hit_kernel)", name)
            k = m.group(1) if m else name[:48]
            sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
    for k, d in sums.items():
        n = max(1, len(calls[k]))
        out.setdefault(k, {}).update({c: v / n for c, v in d.items()})
        out[k]["dispatches_profiled"] = n
print(json.dumps(out, indent=1, sort_keys=True))
