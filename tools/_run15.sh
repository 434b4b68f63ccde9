export MI_EXPECT_GPU=1
B="python bench.py --no-cpu-baseline --no-other-mode --steps 5 --warmup 2"
V=$PWD/shinestacker_amd/csrc/libmi355stack_v576.so
MI355STACK_LIB=$V timeout 600 python -m pytest tests/test_gpu_separable.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
$B > gpurun_out/r15_nt512_$i.json 2>/dev/null
MI355STACK_LIB=$V $B > gpurun_out/r15_nt576_$i.json 2>/dev/null
done
MI355STACK_LIB=$V $B --dtype u8 > gpurun_out/r15_nt576_u8.json 2>/dev/null
$B --dtype u8 > gpurun_out/r15_nt512_u8.json 2>/dev/null
python tools/show.py gpurun_out/r15_*.json
