export MI_EXPECT_GPU=1
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4
( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | tail -3
python tools/show.py gpurun_out/bench_default.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(d['roofline']); print({k:v for k,v in d['other_mode'].items() if k!='parity'}); print(d['cpu_baseline'])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
