#!/bin/bash
# A/B on one box: the default bench (no CPU leg, no verify) with every variant library named, REPS times, interleaved.
cd "$(dirname "$0")/.."
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_$v.so" python bench.py --no-cpu-baseline --no-other-mode --no-verify $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['breakdown_ms_per_step'].items()}, round(d['roofline']['avg_launch_ms'],3))"
  done
done
