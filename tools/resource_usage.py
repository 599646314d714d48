#!/usr/bin/env python3
"""Per-kernel register / spill table from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -c X.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2> X.log
    python tools/resource_usage.py X.log [substring filter]
SGPR spills matter in the persistent kernels: every spilled SGPR costs a v_readlane_b32 + s_nop in the time loop."""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    rows, cur = [], None
    for line in open(sys.argv[1], errors="replace"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-90s %5s %5s %6s %6s %4s" % ("kernel", "SGPR", "VGPR", "Sspill", "scratch", "occ"))
    for r in rows:
        name = demangle(r["name"]).replace("psvo::", "").replace("void ", "").split("(")[0]
        if flt in name:
            print("%-90s %5s %5s %6s %6s %4s" % (name[:90], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("SGPRs Spill"),
                                                r.get("ScratchSize"), r.get("Occupancy")))


if __name__ == "__main__":
    main()
