#!/bin/bash
# rocprofv3 passes for any command:  tools/pmc.sh <tag> <command...>
# kernel trace + stats first, then the PMC groups, each in its own run (counters are never combined with other traces).
# Output: gpurun_out/prof_<tag>/{stats,pmc_*}; summary.txt lists per-kernel times and per-dispatch counter averages.
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- "$@" > $OUT/cmd_stats.out 2> $OUT/stats.log
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $OUT/pmc_$i -o pmc -- "$@" > /dev/null 2> $OUT/pmc_$i.log
done
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
