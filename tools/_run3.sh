export MI_EXPECT_GPU=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-other-mode --steps 4 --warmup 1"
for dt in f32 u8; do $B --arith exact --dtype $dt > gpurun_out/r3_exact_$dt.json 2>gpurun_out/r3_exact_$dt.err; done
python tools/show.py gpurun_out/r3_exact_*.json
tools/timeline_run.sh r03_exact --arith exact >/dev/null 2>&1; python - <<'PY'
import re,collections
agg=collections.OrderedDict()
for l in open('gpurun_out/tl_r03_exact/timeline.txt'):
    m=re.search(r'\(\s*([\d.]+)\)\s+q=\d+ grid=\s*(\d+)\s+(.*)',l)
    if m:
        k=(m.group(3)[:60],m.group(2)); agg.setdefault(k,[0,0.0]); agg[k][0]+=1; agg[k][1]+=float(m.group(1))
for k,v in agg.items(): print('%5d x %9.1f us total  grid %9s  %s'%(v[0],v[1],k[1],k[0]))
PY
