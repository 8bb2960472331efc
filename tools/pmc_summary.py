#!/usr/bin/env python3
"""Aggregates rocprofv3 counter_collection CSVs per kernel: mean counter value per dispatch."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "render_kernel"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    with open(f) as fh:
        per_dispatch = defaultdict(float)
        meta = {}
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if want not in k:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            meta[row["Dispatch_Id"]] = k
        for (disp, cname), v in per_dispatch.items():
            acc[meta[disp]][cname].append(v)
for kname, counters in acc.items():
    print(f"== {kname}")
    for cname in sorted(counters):
        vals = counters[cname]
        print(f"  {cname:42s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
