#!/usr/bin/env python3
"""Fold the counter_collection CSVs of tools/pmc_profile.sh into one JSON (per-dispatch means per kernel)."""
import csv
import glob
import json
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
kernel_filter = sys.argv[3] if len(sys.argv) > 3 else "k_"
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(f"{src}/pass*/**/*counter_collection.csv", recursive=True)):
    per_dispatch = defaultdict(lambda: defaultdict(float))
    names = {}
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if kernel_filter not in k:
            continue
        per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = k.split("(")[0]
    for d, counters in per_dispatch.items():
        for c, v in counters.items():
            acc[names[d]][c].append(v)
out = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:36s} {v['mean']:.6g}  (n={v['n']})")
