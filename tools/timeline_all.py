#!/usr/bin/env python3
"""Timeline of the LAST `span_ms` of a rocprofv3 kernel trace, by queue:  tools/timeline_all.py trace.csv [span_ms] [--summary]
(config 4 / config 5 runs: which kernels share the GPU, where the gaps are)."""
import csv, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"].replace("void mi::", "").replace("mi::", "")
        if "synth" in n:
            continue
        gs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r.get("Grid_Size_Z", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][:44], gs, r.get("Queue_Id", "?")))
rows.sort()
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
end = max(r[1] for r in rows)
if span:
    rows = [r for r in rows if r[0] >= end - span]
t0 = rows[0][0]
if "--summary" in sys.argv:
    by = defaultdict(lambda: [0, 0.0])
    for s, e, n, gs, q in rows:
        by[(q, n)][0] += 1
        by[(q, n)][1] += (e - s) / 1e3
    # union of busy time
    busy, cur_s, cur_e = 0.0, None, None
    for s, e, *_ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"span {(end - t0) / 1e6:.3f} ms, GPU busy (union) {busy / 1e6:.3f} ms, sum of kernels {sum(e - s for s, e, *_ in rows) / 1e6:.3f} ms")
    for (q, n), (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"q={q:>3} {c:6d} x {us / c:9.1f} us = {us / 1e3:9.3f} ms  {n}")
else:
    for s, e, n, gs, q in rows:
        print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us ({(e - s) / 1e3:8.1f}) q={q} grid={gs:9d} {n}")
