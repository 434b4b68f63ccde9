#!/bin/bash
# Per-kernel STANDALONE times of the default bench: the -DMI_STUDY library with MI_SERIAL=1 puts every kernel of the job
# on one stream, rocprofv3 --kernel-trace --stats lists them.  tools/serial_prof.sh <tag> [bench args]
set -u
cd "$(dirname "$0")/.."
TAG=${1:-serial}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_study.so"
MI_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python bench.py --no-cpu-baseline --no-other-mode --no-verify --steps 2 --warmup 1 "$@" > $OUT/bench_stats.json 2> $OUT/stats.log
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
