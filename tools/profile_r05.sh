#!/bin/bash
# Round-5 rocprofv3 evidence (run on the GPU box):  tools/profile_r05.sh
#   1. FETCH_SIZE / WRITE_SIZE calibration on a known 4 GiB stream (tools/calib_fetch.hip), each counter in its own pass;
#   2. tools/profile.sh r05       : default bench (separable, 256 x 24 MP fp32): kernel trace + stats, then the PMC groups;
#   3. tools/profile.sh r05_u8/u16: the same stack held as 8- / 16-bit frames (the reference's input types), all PMC groups;
#   4. tools/profile.sh r05_exact : --arith exact, kernel trace + stats + PMC;
#   5. profiles/traffic.json with provenance (kernel sources' sha256, dtype, frames per launch), read by bench.py;
#   6. kernel timelines of the 256-frame and of the 32-frame step.
# Everything lands under gpurun_out/; copy what is kept into profiles/r05/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/calib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib_fetch 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/calib/$c -o calib -- /tmp/calib_fetch > /dev/null 2>&1
done
tools/profile.sh r05 > /dev/null 2>&1
tools/profile.sh r05_u8 --dtype u8 > /dev/null 2>&1
tools/profile.sh r05_u16 --dtype u16 > /dev/null 2>&1
tools/profile.sh r05_exact --arith exact > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/prof_r05 gpurun_out/calib "level_sep<float, true" > gpurun_out/traffic_sep.json
python tools/pmc_traffic.py gpurun_out/prof_r05_exact gpurun_out/calib "level_fused<float, true, true, 32, 64" > gpurun_out/traffic_exact.json
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
out = {}
for k, f in (("separable", "gpurun_out/traffic_sep.json"), ("exact", "gpurun_out/traffic_exact.json")):
    e = json.load(open(f))
    e.update(source_sha=bench.kernel_source_sha(), dtype="f32", frames_per_launch=16,
             note="r05: one launch = 16 frames of a 256-frame resident push of 4000x6000x3 fp32 frames (tools/profile_r05.sh); "
                  "average over the profiled launches; FETCH_SIZE doubled per the gfx950 calibration, WRITE_SIZE as reported")
    out[k] = e
json.dump(out, open("gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps({k: (v["kernel"], v["hbm_bytes_per_launch"]) for k, v in out.items()}))
PY
tools/timeline_run.sh r05
tools/timeline_run.sh r05_32 --frames 32
for t in r05 r05_u8 r05_u16 r05_exact; do echo "=== $t"; head -45 gpurun_out/prof_$t/summary.txt; done
