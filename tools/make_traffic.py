#!/usr/bin/env python3
"""Writes profiles/traffic.json from the FETCH_SIZE and WRITE_SIZE passes of tools/profile_pmc.sh.

HBM-side bytes per k_primary launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, hence the doubling)."""
import csv, json, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/traffic.json"
TIMED = int(os.environ.get("TIMED_LAUNCHES", "4"))  # the last launches of the run are bench.py's timed steps


def per_launch(group, counter, kernel="k_primary"):
    """Mean over the timed launches only: the warm-up step also holds the short pilot launch."""
    by_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(os.path.join(root, group, "pmc_counter_collection.csv"))):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            by_dispatch[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    last = [by_dispatch[k] for k in sorted(by_dispatch)[-TIMED:]]
    return sum(last) / max(len(last), 1), len(last)
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
