B="python bench.py --no-cpu-baseline --no-other-mode --steps 5 --warmup 2"
for i in 1 2; do
$B > gpurun_out/r17_pf0_$i.json 2>/dev/null
for v in 1 2; do MI355STACK_LIB=$PWD/shinestacker_amd/csrc/libmi355stack_pf$v.so $B > gpurun_out/r17_pf${v}_$i.json 2>/dev/null; done
done
for v in 1 2; do MI355STACK_LIB=$PWD/shinestacker_amd/csrc/libmi355stack_pf$v.so $B --dtype u8 > gpurun_out/r17_pf${v}_u8.json 2>/dev/null; done
$B --dtype u8 > gpurun_out/r17_pf0_u8.json 2>/dev/null
for v in 1 2; do MI355STACK_LIB=$PWD/shinestacker_amd/csrc/libmi355stack_pf$v.so $B --dtype u16 > gpurun_out/r17_pf${v}_u16.json 2>/dev/null; done
$B --dtype u16 > gpurun_out/r17_pf0_u16.json 2>/dev/null
python tools/show.py gpurun_out/r17_*.json
