#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time stats and PMC sums,
keyed by (kernel, grid size) so the pyramid levels stay apart."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    return name.replace("void mi::", "").replace("mi::", "")[:70]


dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            gs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            key = (short(r["Kernel_Name"]), gs)
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in dur.values()) or 1.0
print(f"{'kernel':72s} {'grid':>10s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>9s} {'pct':>6s}")
for key in sorted(dur, key=lambda k: -sum(dur[k]))[:40]:
    v = dur[key]
    print(f"{key[0]:72s} {key[1]:10d} {len(v):6d} {sum(v)/1e3:9.3f} {sum(v)/len(v):9.1f} {min(v):9.1f} "
          f"{100*sum(v)/tot:6.2f}")

pmc = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            pmc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[key][r["Counter_Name"]].add(r["Dispatch_Id"])
print()
for key in sorted(pmc, key=lambda k: -sum(dur.get(k, [0])))[:4]:
    c = {n: v / max(len(ndisp[key][n]), 1) for n, v in pmc[key].items()}  # per dispatch
    print(f"== {key[0]} grid={key[1]}  (per dispatch)")
    for n in sorted(c):
        print(f"      {n:26s} {c[n]:.4g}")
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8.0  # summed over 8 XCDs
    if g:
        cu_cycles = g * 256
        print(f"      -- kernel cycles ~{g:.4g}; LDS pipe busy {100*c.get('SQ_LDS_IDX_ACTIVE',0)/cu_cycles:.1f}% "
              f"(bank conflicts {100*c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):.1f}% of it)")
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY",
                  "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in c:
                print(f"      -- {n}/WAVE_CYCLES = {100*c[n]/wc:.1f}%")
    if "FETCH_SIZE" in c:
        print(f"      -- FETCH_SIZE {c['FETCH_SIZE']/1e6:.3f} GB(KB units) x2 gfx950 correction = "
              f"{2*c['FETCH_SIZE']/1e6:.3f} GB;  WRITE_SIZE {c.get('WRITE_SIZE',0)/1e6:.3f} GB")
    if "TCC_HIT_sum" in c:
        print(f"      -- L2 hit rate {100*c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum']):.1f}%")
