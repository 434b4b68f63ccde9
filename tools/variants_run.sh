#!/bin/bash
# A/B on ONE box: bench.py with each named variant library (tools/variants_build.sh), interleaved REPS times.
#   REPS=3 ARGS="--steps 10" tools/variants_run.sh base dma th44
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/variants
for rep in $(seq 1 ${REPS:-2}); do
  for name in "$@"; do
    lib="$PWD/shinestacker_amd/csrc/variants/libmi355stack_$name.so"
    touch "$lib"   # newer than the sources: bench.py must not rebuild it without its flags
    MI355STACK_LIB="$lib" python bench.py ${ARGS:---steps 10 --no-other-mode --no-cpu-baseline} > gpurun_out/variants/${name}_$rep.json 2> gpurun_out/variants/${name}_$rep.err
    python - "$name" "$rep" gpurun_out/variants/${name}_$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    r = d.get("roofline", {})
    print("%-12s rep %s: %8.0f Mpx/s  %.2f ms/step  level0 launch %.4f ms  frac %.3f  job %.3f  verified %s  %s" % (
        sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_launch_ms", 0), r.get("frac", 0),
        d.get("job_roofline_frac", 0), d.get("verified"), d.get("breakdown_ms_per_step")))
except Exception as e:
    print(sys.argv[1], "rep", sys.argv[2], "FAILED", e)
PY
  done
done
