#!/bin/bash
# warp kernel variants alone on the GPU:  bash tools/warp_ab.sh "<flags A>" "<flags B>" ...
cd "$(dirname "$0")/.."
i=0
for fl in "$@"; do
  i=$((i+1))
  MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_w$i.so" MI_EXTRA_FLAGS="$fl" python -m shinestacker_amd.build --force >/dev/null || exit 1
done
for rep in 1 2; do
  i=0
  for fl in "$@"; do
    i=$((i+1))
    echo "== [$fl]"
    MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_w$i.so" python tools/warp_time.py --only "0.2 deg" | sed 's/^/   /'
  done
done
