#!/usr/bin/env python3
"""one line per bench.py JSON file: value, ms, roofline, verified, breakdown"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        v = d.get('verified')
        v = v.get('ok') if isinstance(v, dict) else v
        print('%-40s %6.1f Gpx/s %6.2f ms roof %.3f job %.3f ver %s' % (f.split('/')[-1], d['value'] / 1e3, d['ms_per_step'], d['roofline']['frac'],
              d.get('job_roofline_frac', 0), v), {k: round(x, 2) for k, x in d.get('breakdown_ms_per_step', {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
