export MI_EXPECT_GPU=1
timeout 900 python -m pytest tests/test_gpu_align.py tests/test_gpu_ecc.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
python tools/config4.py --resident 2>&1 | tail -3
python tools/warp_time.py 2>&1 | tail -6
