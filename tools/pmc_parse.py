#!/usr/bin/env python3
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "pmc"
prefix = sys.argv[2] if len(sys.argv) > 2 else "fq_fused"
agg = collections.defaultdict(list)
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof", tag + "_sq*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(prefix) or (prefix in r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:28s} n={len(v)} avg={sum(v)/len(v):.4g}")
