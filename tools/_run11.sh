export MI_EXPECT_GPU=1
python bench.py --force-combine --scaling strong --no-cpu-baseline --no-other-mode --steps 4 --warmup 1 > gpurun_out/r11_force.json 2> gpurun_out/r11_force.err || tail -20 gpurun_out/r11_force.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r11_force.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','combine_ms','combine_note','verified')}); print(d['breakdown_ms_per_step'])
PY
