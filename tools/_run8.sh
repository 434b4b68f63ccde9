export MI_EXPECT_GPU=1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
( time python bench.py > gpurun_out/r8_bench_default.json 2> gpurun_out/r8_bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r8_bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','job_roofline_frac','verified')}, d['roofline']['frac'])
o=d['other_mode']; print({k:o[k] for k in o if k!='parity'})
p=o['parity']; print({k:p[k] for k in p if k!='levels'}); print(d['cpu_baseline'])
PY
