"""Prints registers, scratch and occupancy of every kernel in kernels.hip (hipcc resource-usage remarks).

usage: python tools/kernel_resources.py [filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + ROOT + "/include",
           "-I" + ROOT + "/raytrace_amd/csrc", "-c", ROOT + "/raytrace_amd/csrc/kernels.hip", "-o", "/tmp/kernel_resources.o",
           "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("EXTRA_FLAGS", "").split()
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark: (?:\s*)(.*?) \[-Rpass", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in body:
            k, v = body.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    for name, r in rows.items():
        pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        if flt and flt not in pretty:
            continue
        print(f"{pretty:60s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} sgpr {r.get('SGPRs','?'):>4} "
              f"scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")


if __name__ == "__main__":
    main()
