export MI_EXPECT_GPU=1
timeout 900 python -m pytest tests/test_gpu_combine.py -x -q 2>&1 | tail -15
