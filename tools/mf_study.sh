#!/bin/bash
# Level-0 interior launch of the 8 / 16-bit separable kernel, matrix-pipe reduce (MF) against the VALU form, with the
# per-phase clocks of a -DMI_PHASE_CLOCK study build.  Run on the GPU box:  bash tools/mf_study.sh [u8|u16]
cd "$(dirname "$0")/.."
export MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_study.so"
MI_EXTRA_FLAGS="-DMI_STUDY -DMI_PHASE_CLOCK $MI_EXTRA_FLAGS" python -m shinestacker_amd.build --force >/dev/null || exit 1
for dt in ${@:-u8}; do
  for nomf in 0 1; do
    echo "== dtype=$dt MI_NO_MFMA=$nomf"
    MI_ONLY_L0=1 MI_NO_MFMA=$nomf python tools/sep_check.py --skip-check --frames 32 --arith separable --dtype $dt 2>&1 | tail -12
  done
done
