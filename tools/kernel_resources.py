#!/usr/bin/env python
"""Per-kernel register / spill / LDS table of one csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; no GPU needed).
    python tools/kernel_resources.py conv_sf16.hip [substring]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(REPO, "sound_event_detection_dcase2017_task4_amd", "csrc", sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"),
                          "-DSED_BUILD_FLAGS_HASH=\"x\"", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0].replace("void ", "")
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print("%-60s %5s %5s %6s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "occ", "LDS"))
    for k, v in rows.items():
        if sub in k:
            print("%-60s %5s %5s %6s %6s %4s %7s" % (k[:60], v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill"), v.get("SGPRs Spill"),
                                                    v.get("Occupancy"), v.get("LDS Size")))


if __name__ == "__main__":
    main()
