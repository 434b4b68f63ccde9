#!/bin/bash
# Round-4 rocprofv3 evidence (run on the GPU box):  tools/profile_r04.sh
#   1. FETCH_SIZE / WRITE_SIZE calibration on a known 4 GiB stream (tools/calib_fetch.hip), each counter in its own pass;
#   2. tools/profile.sh r04      : default bench (separable, 256 x 24 MP fp32): kernel trace + stats, then the PMC groups;
#   3. tools/profile.sh r04_exact: the same for --arith exact (all PMC groups);
#   4. profiles/traffic.json with provenance (kernel sources' sha256, dtype, frames per launch), read by bench.py.
# Everything lands under gpurun_out/; copy what is kept into profiles/r04/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/calib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib_fetch 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/calib/$c -o calib -- /tmp/calib_fetch > /dev/null 2>&1
done
tools/profile.sh r04 > /dev/null 2>&1
tools/profile.sh r04_exact --arith exact > /dev/null 2>&1      # every PMC group for the exact kernel too (round 3 had FETCH / WRITE only)
python tools/pmc_traffic.py gpurun_out/prof_r04 gpurun_out/calib "level_sep<float, true" > gpurun_out/traffic_sep.json
python tools/pmc_traffic.py gpurun_out/prof_r04_exact gpurun_out/calib "level_fused<float, true, true, 32, 64" > gpurun_out/traffic_exact.json
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
out = {}
for k, f in (("separable", "gpurun_out/traffic_sep.json"), ("exact", "gpurun_out/traffic_exact.json")):
    e = json.load(open(f))
    e.update(source_sha=bench.kernel_source_sha(), dtype="f32", frames_per_launch=16,
             note="r04: one launch = 16 frames of a 256-frame resident push of 4000x6000x3 fp32 frames (tools/profile_r04.sh); "
                  "average over the profiled launches; FETCH_SIZE doubled per the gfx950 calibration, WRITE_SIZE as reported")
    out[k] = e
json.dump(out, open("gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps({k: (v["kernel"], v["hbm_bytes_per_launch"]) for k, v in out.items()}))
PY
tools/timeline_run.sh r04
cat gpurun_out/prof_r04/summary.txt | head -60
cat gpurun_out/prof_r04_exact/summary.txt | head -40
