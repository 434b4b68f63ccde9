#!/bin/bash
# Compile the library for gfx950 and print VGPRs / spills / occupancy / LDS of the kernels matching $1 (default: level_).
# Usage: tools/regs.sh [pattern] [extra hipcc flags...]
cd "$(dirname "$0")/.."
pat=${1:-level_}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -shared -fPIC \
  -fvisibility=hidden -pthread -I include "$@" shinestacker_amd/csrc/capi.hip -o /tmp/regs_$$.so \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re, subprocess
pat = sys.argv[1]
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for key in ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "Occupancy \\[waves/SIMD\\]", "LDS Size \\[bytes/block\\]", "ScratchSize \\[bytes/lane\\]"):
        m = re.search(r"remark:\s+" + key + r": (\d+)", line)
        if m and cur is not None: cur[key.replace("\\", "")] = m.group(1)
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n") if names else []
for r, d in zip(rows, dem):
    if pat in d:
        print("%-90s vgpr %3s spill %3s occ %s scratch %s" % (d.replace("mi::", "").replace("(LevelArgs)", "")[:90], r.get("VGPRs"), r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("ScratchSize [bytes/lane]")))
' "$pat"
rm -f /tmp/regs_$$.so
