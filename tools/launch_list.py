#!/usr/bin/env python3
"""Durations (ms) of every launch of the library's kernels in a rocprofv3 kernel trace, in launch order, per kernel.
usage: launch_list.py <dir with *_kernel_trace.csv>"""
import collections, csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").replace("rt::", "").split("(")[0]
    if k.startswith("k_"):
        by.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in by.items():
    print("%-34s n=%3d sum %8.3f | %s" % (k[:34], len(v), sum(v), " ".join("%.2f" % x for x in v[:24])))
