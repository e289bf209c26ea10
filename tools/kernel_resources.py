#!/usr/bin/env python3
"""Compact per-kernel resource table of the product library from the compiler's own report
(-Rpass-analysis=kernel-resource-usage): VGPRs, scratch, LDS, occupancy.  No GPU needed.
    python tools/kernel_resources.py [substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
           "-I" + os.path.join(ROOT, "include"), "-o", "/dev/null",
           os.path.join(ROOT, "qtttgym_amd", "csrc", "qttt_kernels.hip"), "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
            rows.append(cur)
        else:
            cur[k.split(" ")[0]] = v
    print("%-72s %5s %5s %7s %7s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
    for r in rows:
        if pat in r["name"]:
            print("%-72s %5s %5s %7s %7s %4s" % (r["name"][:72], r.get("VGPRs"), r.get("SGPRs"), r.get("ScratchSize"),
                                                  r.get("LDS"), r.get("Occupancy")))


if __name__ == "__main__":
    main()
