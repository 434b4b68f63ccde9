#!/bin/bash
# LDS bank-conflict cycles of the level-0 interior kernel per ablated phase (MI_ABLATE bits:
# 1 reduce, 4 laplacian/Q, 8 energy; 256 = no border kernels)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for ab in 0 1 4 8 13; do
  out=gpurun_out/lds_$ab
  MI_ABLATE=$((ab+256)) rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $out -o pmc -- python bench.py --frames 32 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - "$out" "$ab" <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'level_fused<float, true, true, 32, 64' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
print('ablate',sys.argv[2],{k:'%.3g'%(sum(v)/len(v)) for k,v in d.items()})
PY
done
