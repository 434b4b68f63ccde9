#!/bin/bash
# Interleaved A/B/... of bench.py under different environments (library builds via MI355STACK_LIB, the pair switch via
# SHINESTACKER_AMD_PAIR_LEVELS, ...):  tools/ab.sh <reps> "<env of variant 1>" "<env of variant 2>" ... -- [bench flags]
cd "$(dirname "$0")/.."
R=$1; shift
V=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do V+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in $(seq $R); do
  for v in "${V[@]}"; do
    echo -n "[$v] "
    env $v python tools/bench_line.py "$@"
  done
done
