export MI_EXPECT_GPU=1
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_ecc.py -x -q -k "step_process" 2>&1 | tail -3; done
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
