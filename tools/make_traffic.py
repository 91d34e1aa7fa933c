#!/usr/bin/env python3
"""Writes profiles/traffic.json from the FETCH_SIZE and WRITE_SIZE passes of tools/profile_pmc.sh.

HBM-side bytes per k_primary launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, hence the doubling)."""
import csv, json, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/traffic.json"
def per_launch(group, counter, kernel="k_primary"):
    tot = 0.0; disp = set()
    for r in csv.DictReader(open(os.path.join(root, group, "pmc_counter_collection.csv"))):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    return tot / max(len(disp), 1), len(disp)
f, nf = per_launch("fetch", "FETCH_SIZE")
w, nw = per_launch("write", "WRITE_SIZE")
res = {
    "k_primary_hbm_bytes_per_launch": 2.0 * f * 1024 + w * 1024,
    "fetch_size_kib_per_launch_raw": f, "write_size_kib_per_launch_raw": w,
    "launches_profiled": [nf, nw],
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py; FETCH_SIZE doubled for gfx950",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
