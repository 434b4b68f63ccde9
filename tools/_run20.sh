export MI_EXPECT_GPU=1
timeout 600 python -m pytest tests/test_gpu_ecc.py -x -q 2>&1 | tail -5
for i in 1 2; do
for a in exact separable; do python tools/config4.py --resident --frames 128 --arith $a 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('native $a', round(d['seconds'],4), d['worst_error']['shift_px'])"; done
python tools/config4.py --resident --frames 128 --python-loop 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('python loop exact', round(d['seconds'],4))"
done
