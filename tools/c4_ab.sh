#!/bin/bash
# config 4 (128 x 24 MP, handles reused), A/B of library builds:  bash tools/c4_ab.sh "<flags A>" "<flags B>" [repetitions]
cd "$(dirname "$0")/.."
A="$1"; B="$2"; R=${3:-3}
MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_A.so" MI_EXTRA_FLAGS="$A" python -m shinestacker_amd.build --force >/dev/null || exit 1
MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_B.so" MI_EXTRA_FLAGS="$B" python -m shinestacker_amd.build --force >/dev/null || exit 1
for i in $(seq $R); do
  for v in A B; do
    echo -n "$v: "
    MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_$v.so" python tools/config4.py --frames 128 --resident --reuse-handles --arith separable ${C4_FLAGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f s  shift %.4f px' % (d['seconds'], d['worst_error']['shift_px']))"
  done
done
