#!/bin/bash
# tile height of the separable kernels (-DMI_SEP_TH=44 / 56 builds at libmi355stack_th<TH>.so): parity, then interleaved A/B
cd "$(dirname "$0")/.."
C=$PWD/shinestacker_amd/csrc
for th in 44 56; do
  MI355STACK_LIB=$C/libmi355stack_th$th.so SHINESTACKER_AMD_PAIR_LEVELS=2 timeout 600 python -m pytest tests/test_gpu_separable.py -m gpu -x -q 2>&1 | tail -2
done
for d in u8 u16 f32; do
  tools/ab.sh 2 "MI355STACK_LIB=$C/libmi355stack.so SHINESTACKER_AMD_PAIR_LEVELS=2" "MI355STACK_LIB=$C/libmi355stack_th44.so SHINESTACKER_AMD_PAIR_LEVELS=2" "MI355STACK_LIB=$C/libmi355stack_th56.so SHINESTACKER_AMD_PAIR_LEVELS=2" -- --dtype $d
done
