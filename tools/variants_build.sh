#!/bin/bash
# Build named variants of the library HERE (hipcc cross-compiles; the .so files travel to the GPU box with gpurun):
#   tools/variants_build.sh base "" dma "-DMI_SEP_DMA=1" th44 "-DMI_SEP_TH=44"
# -> shinestacker_amd/csrc/variants/libmi355stack_<name>.so ; run them with tools/variants_run.sh
cd "$(dirname "$0")/.."
mkdir -p shinestacker_amd/csrc/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( MI355STACK_LIB="$PWD/shinestacker_amd/csrc/variants/libmi355stack_$name.so" MI_EXTRA_FLAGS="$flags" \
      python -m shinestacker_amd.build --force > /dev/null 2>&1 && echo "built $name [$flags]" || echo "FAILED $name [$flags]" ) &
done
wait
