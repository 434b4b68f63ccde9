#!/bin/bash
# Level-0 interior launch of the separable kernel ALONE on the GPU (no border tiles, no coarser levels), 32 fp32
# frames, with phases ablated.  Needs the study build (tools/study_build.sh).
# bits: 1 P1 v-reduce, 2 P2 h-reduce/expand, 4 P3 lapq, 8 P4 select, 16 prefetch after frame 0, 32 G_{l+1} store
cd "$(dirname "$0")/.."
export MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_study.so"
for ab in ${ABLATES:-0 1 2 4 8 15 16 31 47}; do
  echo -n "arith=${ARITH:-separable} ablate=$ab: "
  MI_ONLY_L0=1 MI_ABLATE=$((ab + 256)) python tools/sep_check.py --skip-check --frames ${FRAMES:-32} --arith ${ARITH:-separable} --dtype ${DTYPE:-f32} | tail -1
done
