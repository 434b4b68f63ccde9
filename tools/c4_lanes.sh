#!/bin/bash
# config 4 against the number of warp lanes (libmi355stack_A.so = -DMI_WARP_LANES=1, _B = 2, release = 3), interleaved
cd "$(dirname "$0")/.."
C=$PWD/shinestacker_amd/csrc
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.4f s  shift %.4f px" % (d["seconds"], d["worst_error"]["shift_px"]))'
for i in 1 2 3; do
  for v in A B ""; do
    L=$C/libmi355stack${v:+_$v}.so
    echo -n "lanes ${v:-3 (release)}: "; MI355STACK_LIB=$L python tools/config4.py --frames 128 --resident --reuse-handles --arith separable $C4_FLAGS 2>/dev/null | python -c "$P"
  done
done
