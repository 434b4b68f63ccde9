#!/bin/bash
# Does level 1 read G_1 out of the Infinity Cache when it runs right behind the level-0 launch that wrote it?  Study build,
# level 1 interleaved with level 0 (MI_INTERLEAVE01) in frame groups small enough for the 256 MB cache (MI_LAUNCH_FRAMES):
# per-frame duration of the level-1 launches (rocprofv3 kernel trace), against the shipped 16-frame launches.
cd "$(dirname "$0")/.."
export MI355STACK_LIB="$PWD/shinestacker_amd/csrc/libmi355stack_study.so" TMPDIR=/tmp
MI_EXTRA_FLAGS="-DMI_STUDY $MI_EXTRA_FLAGS" python -m shinestacker_amd.build --force >/dev/null || exit 1
run() {   # tag, env...
  tag=$1; shift
  rm -rf gpurun_out/mall_$tag; mkdir -p gpurun_out/mall_$tag
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mall_$tag -o t -- python tools/sep_check.py --skip-check --frames 64 --arith separable --dtype f32 > gpurun_out/mall_$tag/out.txt 2>&1
  tail -1 gpurun_out/mall_$tag/out.txt
  python - "$tag" <<'PY'
import csv, sys, glob
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob(f"gpurun_out/mall_{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "level_sep" in n:
            d[(("coarse" if "coarse" in n else "level0"), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:3]:
    print("   ", k, "n=%d avg %.1f us  total %.2f ms" % (len(v), sum(v) / len(v), sum(v) / 1e3))
PY
}
run base MI_X=0
run il16 MI_INTERLEAVE01=1
run il4 MI_INTERLEAVE01=1 MI_LAUNCH_FRAMES=4
run il2 MI_INTERLEAVE01=1 MI_LAUNCH_FRAMES=2
