export TMPDIR=/tmp
for a in exact separable; do python tools/config4.py --resident --frames 128 --arith $a 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['seconds'], d['worst_error'])"; done
python tools/config4.py --resident --frames 128 --arith separable --batch 32 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sep batch32', d['seconds'])"
OUT=gpurun_out/prof_c4; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python tools/config4.py --resident --frames 128 --arith separable > $OUT/out.json 2> $OUT/log
python tools/summarize_prof.py $OUT 2>/dev/null | head -40
