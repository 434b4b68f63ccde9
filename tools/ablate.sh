#!/bin/bash
# timing study of the fused level kernel: skip phases via MI_ABLATE bits
# 1 reduce, 2 gnext store, 4 lapq, 8 energy, 16 prefetch(global loads after first)
cd "$(dirname "$0")/.."
for ab in 0 1 4 8 5 13 15 31; do
  echo -n "ablate=$ab : "
  MI_ABLATE=$ab python bench.py --frames 32 --steps 3 --warmup 1 --no-cpu-baseline "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f Gpx/s  %.2f ms/step' % (d['value']/1e3, d['ms_per_step']), {k: round(v,2) for k,v in d.get('breakdown_ms_per_step',{}).items()})"
done
