#!/bin/bash
# build + bench tile-configuration variants on the GPU box:
#   tools/tune.sh "32,64,512,32,32,256,1" ...   (TH0,TW0,NT0 for level 0; TH,TW,NT others; PAD)
cd "$(dirname "$0")/.."
FR=${FRAMES:-64}
for cfg in "$@"; do
  MI_TILE_CFG=$cfg python -m shinestacker_amd.build --force > /dev/null 2>&1 || { echo "$cfg: build failed"; continue; }
  echo -n "cfg=$cfg : "
  python bench.py --frames $FR --steps 3 --warmup 1 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f Gpx/s  %.2f ms/step' % (d['value']/1e3, d['ms_per_step']), {k: round(v,2) for k,v in d.get('breakdown_ms_per_step',{}).items()})"
done
python -m shinestacker_amd.build --force > /dev/null 2>&1
