#!/usr/bin/env python3
"""Which kernels hold the GPU, in buckets of `ms` milliseconds, over the last `span_ms` of a rocprofv3 kernel trace:
    tools/timeline_buckets.py trace.csv [span_ms] [bucket_ms]"""
import collections
import csv
import sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"].replace("void mi::", "").replace("mi::", "").split("<")[0].split("(")[0]
        if "synth" in n:
            continue
        rows.append((int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3, n))
rows.sort()
span = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else None
B = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 2e3
end = max(r[1] for r in rows)
if span:
    rows = [r for r in rows if r[0] >= end - span]
t0 = rows[0][0]
buck = collections.defaultdict(collections.Counter)
for s, e, n in rows:
    for b in range(int((s - t0) // B), int((e - t0) // B) + 1):
        lo, hi = max(s - t0, b * B), min(e - t0, (b + 1) * B)
        if hi > lo:
            buck[b][n] += (hi - lo) / 1e3
for b in sorted(buck):
    print("%6.1f ms: " % (b * B / 1e3) + ", ".join("%s %.2f" % kv for kv in buck[b].most_common(6)))
