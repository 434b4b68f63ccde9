#!/bin/bash
# Build variants of the library locally (one .so per flag set) for A/B runs on ONE GPU box:
#   tools/variants_build.sh name1 "-DFLAG.." name2 "-DFLAG.."   ->  shinestacker_amd/csrc/libmi355stack_<name>.so
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  export MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_$1.so"
  MI_EXTRA_FLAGS="-DMI_STUDY $2" python -m shinestacker_amd.build --force > /dev/null 2>&1 && echo "$MI355STACK_LIB" || echo "build failed: $1"
  shift 2
done
