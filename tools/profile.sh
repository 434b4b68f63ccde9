#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun).
#   tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}/ ; copy the summaries you keep into profiles/.
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-other-mode --no-other-dtypes --no-projection --steps 2 --warmup 1 $*"
# 1) kernel trace + stats (no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.log
# 2) PMC passes, each in its own run (kernel-trace only)
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $OUT/pmc_$i -o pmc -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_$i.log
done
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
