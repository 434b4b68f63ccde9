#!/bin/bash
# Phase ablation of the level-0 launch (study build): MF against the VALU form.  bash tools/mf_ablate.sh [u8|u16]
# bits: 1 P1 v-reduce, 2 P2 / MF reduce, 4 P3 lapq, 8 P4 select, 16 prefetch after frame 0, 32 G_{l+1} store
cd "$(dirname "$0")/.."
export MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_study.so"
MI_EXTRA_FLAGS="-DMI_STUDY $MI_EXTRA_FLAGS" python -m shinestacker_amd.build --force >/dev/null || exit 1
DT=${1:-u8}
for nomf in 0 1; do
  for ab in ${ABLATES:-0 1 2 3 4 8 12 15 16 32 48 63}; do
    echo -n "dtype=$DT nomf=$nomf ablate=$ab: "
    MI_NO_MFMA=$nomf MI_ONLY_L0=1 MI_ABLATE=$((ab + 256)) python tools/sep_check.py --skip-check --frames 32 --arith separable --dtype $DT | tail -1 | sed 's/.*level0 \([0-9.]*\) ms.*/\1 ms\/launch/'
  done
done
