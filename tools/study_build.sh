#!/bin/bash
# Build the timing-study variant of the library (-DMI_STUDY: phase ablation and launch-mix knobs read from the
# environment) next to the release one and print its path; use it with MI355STACK_LIB=<path>.
cd "$(dirname "$0")/.."
export MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_study.so"
MI_EXTRA_FLAGS="-DMI_STUDY $MI_EXTRA_FLAGS" python -m shinestacker_amd.build --force >/dev/null && echo "$MI355STACK_LIB"
