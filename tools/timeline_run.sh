#!/bin/bash
# kernel timeline of the last bench step:  tools/timeline_run.sh <tag> [bench args]
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/tl_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --no-cpu-baseline --no-other-mode --no-other-dtypes --no-projection --no-verify --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/log
python tools/timeline.py $OUT/t_kernel_trace.csv > $OUT/timeline.txt
