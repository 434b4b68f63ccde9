#!/bin/bash
# bench the streaming impl over a few segment heights (and dtypes)
cd "$(dirname "$0")/.."
for args in "$@"; do
for seg in ${SEGS:-64}; do
MI_STREAM_SEG=$seg python bench.py --impl stream --frames ${FRAMES:-64} --steps 3 --warmup 1 --no-cpu-baseline $args 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('seg=$seg $args', d['config'].get('impl'), '%.1f Gpx/s  %.2f ms/step' % (d['value']/1e3, d['ms_per_step']), {k: round(v,2) for k,v in d.get('breakdown_ms_per_step',{}).items()}, 'roofline', round(d['roofline']['frac'],3))"
done; done
