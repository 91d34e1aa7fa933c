#!/usr/bin/env python3
"""Writes profiles/counters.json from the passes of tools/profile_pmc.sh over a bench.py run.

Per launch of the dominant kernel (k_beam) in the timed steps:
  HBM-side bytes = 2 * FETCH_SIZE + WRITE_SIZE   (both in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes)
  VALU busy      = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
  lanes enabled  = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  VALU instructions per wave of 64 rays = SQ_INSTS_VALU / (rays per launch / 64)
usage: make_counters.py [pmc dir] [out json] [rays per launch]"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (source_hash: the device sources these figures belong to; run this on the tree that was profiled)

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/counters.json"
rays = float(sys.argv[3]) if len(sys.argv) > 3 else 8493465600.0
TIMED = int(os.environ.get("TIMED_LAUNCHES", "4"))  # the last launches of the run are bench.py's timed steps
KERNEL = os.environ.get("PMC_KERNEL", "k_beam<")  # k_beam<...>, not k_beam_prepare


def per_launch(group, counter):
    """Mean over the timed launches only (the warm-up also holds pilot launches)."""
    path = os.path.join(root, group, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        return None
    by_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
            by_dispatch[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    last = [by_dispatch[k] for k in sorted(by_dispatch)[-TIMED:]]
    return sum(last) / len(last) if last else None


f, w = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
act, thr = per_launch("sq2", "SQ_ACTIVE_INST_VALU"), per_launch("sq2", "SQ_THREAD_CYCLES_VALU")
insts, gui = per_launch("sq1", "SQ_INSTS_VALU"), per_launch("grbm", "GRBM_GUI_ACTIVE")
res = {
    "csrc_sha16": bench.source_hash(),
    "kernel": KERNEL, "timed_launches_averaged": TIMED, "rays_per_launch": rays,
    "hbm_bytes_per_launch": (2.0 * f + w) * 1024 if f is not None and w is not None else None,
    "fetch_size_kib_per_launch_raw": f, "write_size_kib_per_launch_raw": w,
    "kernel_cycles": gui / 8 if gui else None,
    "valu_busy": 4.0 * act / (1024.0 * gui / 8.0) if act and gui else None,
    "lanes_enabled": thr / (64.0 * act) if act and thr else None,
    "valu_insts_per_ray": insts / rays if insts else None,
    "valu_insts_per_wave_of_64_rays": insts / (rays / 64.0) if insts else None,
    "method": "rocprofv3 --pmc, one counter group per run of bench.py (tools/profile_pmc.sh); FETCH_SIZE doubled for gfx950; "
              "means over the launches of the timed steps",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
