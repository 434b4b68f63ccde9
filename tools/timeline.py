#!/usr/bin/env python3
"""Print a compact timeline of one bench step from a rocprofv3 kernel trace CSV."""
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"].replace("void mi::", "")
        if "synth" in n:
            continue
        gs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:46], gs, r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
# last step: take the last finalize as the end, and the previous finalize as the start marker
fin = [i for i, r in enumerate(rows) if "finalize" in r[2] or "collapse_final" in r[2] or r[2].startswith("collapse_sep<unsigned")]
lo = fin[-2] + 1 if len(fin) >= 2 else 0
hi = fin[-1] + 1
t0 = rows[lo][0]
for s, e, n, gs, q in rows[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  q={q} grid={gs:9d}  {n}")
