#!/bin/bash
# build study variants of the separable kernel and time the level-0 interior launch alone (32 fp32 frames)
cd "$(dirname "$0")/.."
for v in "${@:-"-DMI_SEP_TH=28"}"; do
  export MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_study.so"
  MI_EXTRA_FLAGS="-DMI_STUDY $v" python -m shinestacker_amd.build --force > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  for rep in 1 2; do
    echo -n "[$v] "
    MI_ONLY_L0=1 MI_ABLATE=256 python tools/sep_check.py --skip-check --frames 32 --arith separable | tail -1 | sed 's/separable  32 x 6000x4000 float32://'
  done
done
