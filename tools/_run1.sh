set -x
export MI_EXPECT_GPU=1
timeout 900 python -m pytest tests/test_gpu_separable.py -x -q 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-other-mode --steps 5 --warmup 2"
STUDY=$PWD/shinestacker_amd/csrc/libmi355stack_study.so
for i in 1 2; do
MI355STACK_LIB=$STUDY MI_EDGE_FOLD=0 $B > gpurun_out/ab_fold0_$i.json 2>/dev/null
MI355STACK_LIB=$STUDY MI_EDGE_FOLD=1 $B > gpurun_out/ab_fold1_$i.json 2>/dev/null
done
$B > gpurun_out/rel_f32.json 2>/dev/null
$B --dtype u8 > gpurun_out/rel_u8.json 2>/dev/null
$B --dtype u16 > gpurun_out/rel_u16.json 2>/dev/null
for f in gpurun_out/ab_*.json gpurun_out/rel_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], '%.1f Gpx/s %.2f ms' % (d['value']/1e3, d['ms_per_step']), 'roof %.3f' % d['roofline']['frac'], 'verified', d.get('verified',{}).get('ok') if isinstance(d.get('verified'),dict) else d.get('verified'), {k: round(v,2) for k,v in d.get('breakdown_ms_per_step',{}).items()})
PY
done
tools/timeline_run.sh r03_fold >/dev/null 2>&1; tail -40 gpurun_out/tl_r03_fold/timeline.txt
