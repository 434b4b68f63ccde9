#!/bin/bash
# bench.py A/B of two library builds, interleaved:  bash tools/bench_ab.sh "<flags A>" "<flags B>" [reps] -- [bench flags]
cd "$(dirname "$0")/.."
A="$1"; B="$2"; R=${3:-2}; shift 3; [ "$1" = "--" ] && shift
MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_A.so" MI_EXTRA_FLAGS="$A" python -m shinestacker_amd.build --force >/dev/null || exit 1
MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_B.so" MI_EXTRA_FLAGS="$B" python -m shinestacker_amd.build --force >/dev/null || exit 1
for i in $(seq $R); do
  for v in A B; do
    echo -n "$v: "
    MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_$v.so" python tools/bench_line.py "$@"
  done
done
