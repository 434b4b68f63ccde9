export MI_EXPECT_GPU=1
python tools/config5.py --two-stage --frames 130 --height 5760 2>&1 | tail -1 | tee gpurun_out/config5_two_stage.json
bash tools/profile_r03.sh
