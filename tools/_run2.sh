export MI_EXPECT_GPU=1
timeout 900 python -m pytest tests/test_gpu_separable.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-other-mode --steps 5 --warmup 2"
STUDY=$PWD/shinestacker_amd/csrc/libmi355stack_study.so
show() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, '%.1f Gpx/s %.2f ms' % (d['value']/1e3, d['ms_per_step']), 'roof %.3f' % d['roofline']['frac'], 'ok' if d.get('verified',{}).get('ok') else d.get('verified'), {k: round(v,2) for k,v in d.get('breakdown_ms_per_step',{}).items()})
    except Exception as e: print(f, 'ERR', e)
PY
}
for dt in f32 u8 u16; do $B --dtype $dt > gpurun_out/r2_rel_$dt.json 2>gpurun_out/r2_rel_$dt.err; done
show gpurun_out/r2_rel_*.json
for pt in 3072 4000 100000; do
MI355STACK_LIB=$STUDY MI_PAR_TILES=$pt $B > gpurun_out/r2_par$pt.json 2>/dev/null
done
MI355STACK_LIB=$STUDY MI_LAUNCH_FRAMES=32 $B > gpurun_out/r2_lf32.json 2>/dev/null
MI355STACK_LIB=$STUDY MI_PAR_TILES=100000 $B --dtype u8 > gpurun_out/r2_par100000_u8.json 2>/dev/null
show gpurun_out/r2_par*.json gpurun_out/r2_lf32.json
