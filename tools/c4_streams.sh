#!/bin/bash
# config 4 and the headline against the number of streams / hardware queues (study build)
cd "$(dirname "$0")/.."
L=$PWD/shinestacker_amd/csrc/libmi355stack_study.so
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.4f s" % d["seconds"])'
for i in 1 2; do
  for v in "MI_PAYLOAD_STREAM=1" "MI_PAYLOAD_STREAM=1 GPU_MAX_HW_QUEUES=8" "MI_PAYLOAD_STREAM=2" "MI_PAYLOAD_STREAM=0"; do
    echo -n "config 4 [$v]: "; env MI355STACK_LIB=$L $v python tools/config4.py --frames 128 --resident --reuse-handles --arith separable 2>/dev/null | python -c "$P"
  done
done
tools/ab.sh 2 "MI355STACK_LIB=$L MI_PAYLOAD_STREAM=1" "MI355STACK_LIB=$L MI_PAYLOAD_STREAM=1 GPU_MAX_HW_QUEUES=8" "MI355STACK_LIB=$L MI_PAYLOAD_STREAM=2" "MI355STACK_LIB=$L MI_PAYLOAD_STREAM=0"
