#!/bin/bash
# Two PMC passes (instruction counts, pipe activity) + a kernel trace of a 32-frame resident 24 MP stack, per-dispatch averages of
# the kernels matching a pattern:  tools/pmc_quick.sh <tag> <dtype> <pattern> [ENV=..]...
cd "$(dirname "$0")/.."
TAG=$1; DT=$2; PAT=$3; shift 3
OUT=gpurun_out/pq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python tools/sep_check.py --skip-check --frames 32 --arith separable --dtype $DT"
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $OUT/pmc_$i -o pmc -- $CMD > /dev/null 2> $OUT/pmc_$i.log
done
python - "$OUT" "$PAT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); nd = defaultdict(lambda: defaultdict(set)); dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = (r["Kernel_Name"].replace("void mi::", "")[:48], int(r["Grid_Size"]))
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for f in glob.glob(os.path.join(out, "pmc_1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            gs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            dur[(r["Kernel_Name"].replace("void mi::", "")[:48], gs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in acc:
    c = {n: v / len(nd[k][n]) for n, v in acc[k].items()}
    d = dur.get(k, [0])
    print(k, "avg_us %.1f" % (sum(d) / max(len(d), 1)), " ".join(f"{n[3:] if n.startswith('SQ_') else n}={v:.4g}" for n, v in sorted(c.items())))
PY
