mkdir -p gpurun_out/r06
L=$PWD/shinestacker_amd/csrc/libmi355stack_study.so
for a in "--arith exact" "--dtype u8" "--dtype u16" "--frames 32 --shards interleaved"; do
tools/ab.sh 2 "MI355STACK_LIB=$L MI_PAYLOAD_STREAM=1" "MI355STACK_LIB=$L MI_PAYLOAD_STREAM=2" -- $a
done
python tools/shard_step.py 8
python tools/config4.py --frames 128 --resident --reuse-handles --arith separable 2>/dev/null | cut -c1-160
python tools/config4.py --frames 128 --resident --reuse-handles --arith separable --step-process 2>/dev/null | cut -c1-200
