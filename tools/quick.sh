#!/bin/bash
# quick GPU check: parity subset + bench breakdown
cd "$(dirname "$0")/.."
export MI_EXPECT_GPU=1
python -m pytest tests -q -m gpu -x -k "golden_fusion or seeded or large" 2>&1 | tail -3
for f in 32 128; do
python bench.py --frames $f --steps 3 --warmup 1 --no-cpu-baseline "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f Gpx/s  %.2f ms/step' % (d['value']/1e3, d['ms_per_step']), {k: round(v,2) for k,v in d.get('breakdown_ms_per_step',{}).items()}, 'roofline', round(d['roofline']['frac'],3))"
done
