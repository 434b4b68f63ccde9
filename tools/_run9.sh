export MI_EXPECT_GPU=1
python tools/parity_report.py > gpurun_out/parity_full.json 2>gpurun_out/parity_full.err || tail -5 gpurun_out/parity_full.err; python - <<'PY'
import json
p=json.load(open('gpurun_out/parity_full.json')); print({k:p[k] for k in p if k!='levels'})
PY
python tools/parity_report.py --dtype u8 --frames 64 > gpurun_out/parity_u8.json 2>/dev/null; python - <<'PY'
import json
p=json.load(open('gpurun_out/parity_u8.json')); print({k:p[k] for k in p if k!='levels'})
PY
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
