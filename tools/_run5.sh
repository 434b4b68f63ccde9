export MI_EXPECT_GPU=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-other-mode --steps 4 --warmup 1 --arith exact"
STUDY=$PWD/shinestacker_amd/csrc/libmi355stack_study.so
$B > gpurun_out/r5_exact_rel.json 2>/dev/null
MI355STACK_LIB=$STUDY $B > gpurun_out/r5_exact_study.json 2>/dev/null
MI355STACK_LIB=$STUDY MI_ABLATE=256 $B --no-verify > gpurun_out/r5_exact_nobd.json 2>/dev/null
MI355STACK_LIB=$STUDY MI_LAUNCH_FRAMES=32 $B > gpurun_out/r5_exact_lf32.json 2>/dev/null
MI355STACK_LIB=$STUDY MI_LAUNCH_FRAMES=8 $B > gpurun_out/r5_exact_lf8.json 2>/dev/null
MI355STACK_LIB=$STUDY MI_WIDE_LEVELS=2 $B > gpurun_out/r5_exact_wide2.json 2>/dev/null
MI355STACK_LIB=$STUDY MI_WIDE_LEVELS=3 $B > gpurun_out/r5_exact_wide3.json 2>/dev/null
MI355STACK_LIB=$STUDY MI_BD_PRIO=2 $B > gpurun_out/r5_exact_bdlow.json 2>/dev/null
python tools/show.py gpurun_out/r5_*.json
