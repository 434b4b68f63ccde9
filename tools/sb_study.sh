#!/bin/bash
# Super-block shape study (GPU box): for each variant library (tools/variants_build.sh) the level-0 launch time from
# bench.py's own events (interleaved repetitions) and FETCH_SIZE / TCC hits of the level-0 kernel from one PMC pass each.
#   tools/sb_study.sh base w12h8 w8h12 ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
REPS=${REPS:-2} ARGS="--steps 5 --no-other-mode --no-cpu-baseline --no-other-dtypes --no-projection --no-verify" tools/variants_run.sh "$@"
for name in "$@"; do
  lib="$PWD/shinestacker_amd/csrc/variants/libmi355stack_$name.so"
  for ctrs in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=gpurun_out/sb_$name/$(echo $ctrs | cut -c1-5)
    mkdir -p $d
    MI355STACK_LIB="$lib" rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $d -o pmc -- python bench.py --steps 1 --warmup 1 --no-other-mode --no-cpu-baseline --no-other-dtypes --no-projection --no-verify > /dev/null 2>&1
  done
  python - "$name" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(f"gpurun_out/sb_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "level_sep<float, true" in r["Kernel_Name"] and int(r["Grid_Size"]) > 6000000:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in acc.items()}
f = out.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9
h, m = out.get("TCC_HIT_sum", 0), out.get("TCC_MISS_sum", 0)
print(f"{sys.argv[1]:10s} FETCH x 2 = {f:.3f} GB per launch, L2 hit {100 * h / max(h + m, 1):.1f} %")
PY
done
