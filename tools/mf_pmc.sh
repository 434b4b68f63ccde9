#!/bin/bash
# PMC of the level-0 launch of an 8 / 16-bit stack: the matrix-pipe reduce (default build) against the VALU form (a build with
# -DMI_SEP_MFMA=0 next to it).  Run on the GPU box:  bash tools/mf_pmc.sh [u8|u16]
cd "$(dirname "$0")/.."
DT=${1:-u8}
MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_nomf.so" MI_EXTRA_FLAGS="-DMI_SEP_MFMA=0" python -m shinestacker_amd.build --force >/dev/null || exit 1
CMD="python tools/sep_check.py --skip-check --frames 32 --arith separable --dtype $DT"
bash tools/pmc.sh mf_$DT $CMD 
MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_nomf.so" bash tools/pmc.sh nomf_$DT $CMD 
