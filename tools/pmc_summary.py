#!/usr/bin/env python3
"""Summarises gpurun_out/pmc/*/pmc_counter_collection.csv: per kernel, counter sums over dispatches."""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
want = sys.argv[2] if len(sys.argv) > 2 else "k_primary"
out = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(root, "*", "pmc_counter_collection.csv"))):
    n = collections.Counter(); disp = set()
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            n[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    for k, v in n.items():
        out[k] = (v, len(disp))
for k, (v, d) in out.items():
    print(f"{k:28s} total {v:16.0f}  per-dispatch {v/max(d,1):16.0f}  ({d} dispatches)")
