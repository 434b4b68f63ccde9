#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from the PMC passes of tools/profile.sh, corrected with the
calibration factors measured by tools/calib_fetch.hip (gpurun_out/calib/).

    python tools/pmc_traffic.py gpurun_out/prof_<tag> gpurun_out/calib [kernel-name-substring] > traffic_<arith>.json

The kernel is the one whose name contains the substring (default "level_fused") with the largest grid; the output is
one entry of profiles/traffic.json ({"separable": {...}, "exact": {...}}, read by bench.py for roofline.traffic).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

prof, calib = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "level_fused"
BYTES = 4 << 30


def counters(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                out[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return out


cal = counters(calib)
fac = {}
for k, v in cal.items():
    for name in ("read8", "read16", "write16", "write4"):
        if name + "(" in k or k.startswith(name):
            ctr = "FETCH_SIZE" if name.startswith("read") else "WRITE_SIZE"
            if ctr in v:
                kb = sum(x[1] for x in v[ctr]) / len(v[ctr])
                fac[name] = BYTES / (kb * 1024.0)  # true bytes per reported byte
pm = counters(prof)
best = None
for k, v in pm.items():
    if want in k and "FETCH_SIZE" in v:
        g = max(x[0] for x in v["FETCH_SIZE"])
        if best is None or g > best[1]:
            best = (k, g)
k, g = best
fetch = [x[1] for x in pm[k]["FETCH_SIZE"] if x[0] == g]
write = [x[1] for x in pm[k]["WRITE_SIZE"] if x[0] == g]
f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
rd = f_kb * 1024 * fac.get("read8", 2.0)
wr = w_kb * 1024 * fac.get("write16", 1.0)
json.dump({"kernel": k, "grid": g, "dispatches": len(fetch), "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
           "calibration_true_bytes_per_reported_byte": fac,
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
           "hbm_bytes_per_launch": rd + wr}, sys.stdout, indent=1)
print()
