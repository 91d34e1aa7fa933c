#!/usr/bin/env python3
"""Per-kernel HBM-side traffic of the runs of tools/profile_big.sh: bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB summed over a
kernel's dispatches (FETCH_SIZE doubled: on gfx950 it tallies 128-byte requests at 64 bytes), time from the kernel trace of
the separate --stats run, rate against 8 TB/s.  usage: big_summary.py [dir]"""
import collections, csv, glob, os, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/big"


def short(name):
    name = name.replace("void ", "").replace("rt::", "")
    return name.split("(")[0][:44]


def counter(pass_name):
    tot = collections.defaultdict(float)
    for path in glob.glob(os.path.join(root, pass_name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == pass_name:
                tot[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return tot


fetch, write = counter("FETCH_SIZE"), counter("WRITE_SIZE")
dur, calls = collections.defaultdict(float), collections.defaultdict(int)
for path in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
        calls[k] += 1
print(f"{'kernel':46s} {'calls':>6s} {'ms':>9s} {'HBM-side GB':>12s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
all_b = all_t = 0.0
for k in sorted(dur, key=lambda k: -dur[k]):
    if not k.startswith("k_"):
        continue
    b = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
    all_b += b
    all_t += dur[k]
    print(f"{k:46s} {calls[k]:6d} {dur[k]*1e3:9.2f} {b/1e9:12.2f} {b/dur[k]/1e9:8.0f} {b/dur[k]/8e12:9.3f}")
print(f"{'all library kernels':46s} {'':6s} {all_t*1e3:9.2f} {all_b/1e9:12.2f} {all_b/max(all_t,1e-12)/1e9:8.0f} {all_b/max(all_t,1e-12)/8e12:9.3f}")
