"""Aggregate rocprofv3 counter_collection CSVs (one per PMC pass) per kernel: mean per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pass*_counters.csv"))):
    with open(f) as fh:
        per_dispatch = defaultdict(float)
        names = {}
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "mr::" not in k:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = k.split("(")[0].replace("void ", "")
        for (disp, cname), v in per_dispatch.items():
            acc[names[disp]][cname].append(v)
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:32s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
