export MI_EXPECT_GPU=1
( time timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "two_stage" 2>&1 | tail -5 ) 2>&1 | tail -9
python tools/config5.py --two-stage --frames 130 --height 5760 2>&1 | tail -2
python tools/config5.py --frames 64 2>&1 | tail -1
