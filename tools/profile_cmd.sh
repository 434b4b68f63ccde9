#!/bin/bash
# rocprofv3 --kernel-trace --stats of any command:  tools/profile_cmd.sh <tag> <command...>
# -> gpurun_out/prof_<tag>/{stats/, summary.txt, cmd.out}
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- "$@" > $OUT/cmd.out 2> $OUT/stats.log
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
