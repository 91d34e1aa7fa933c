#!/usr/bin/env python3
"""Per-kernel figures from the passes of tools/profile_configs_pmc.sh <config>: lanes enabled =
SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); VALU busy = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8);
vector instructions per wave.  usage: cfg_pmc_summary.py gpurun_out/cfgpmc_<config>"""
import collections, csv, glob, os, sys

root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("rt::", "").split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
print(f"{'kernel':40s} {'lanes enabled':>13s} {'VALU busy':>10s} {'VALU/wave':>10s} {'waves':>10s}")
for k, c in sorted(tot.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    if not k.startswith("k_") or not c["SQ_ACTIVE_INST_VALU"]:
        continue
    lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0) if c["GRBM_GUI_ACTIVE"] else 0.0
    print(f"{k:40s} {lanes:13.3f} {busy:10.3f} {c['SQ_INSTS_VALU'] / max(c['SQ_WAVES'], 1):10.0f} {c['SQ_WAVES']:10.0f}")
