#!/bin/bash
# Round-6 rocprofv3 evidence (run on the GPU box):  tools/profile_r06.sh
#   1. FETCH_SIZE / WRITE_SIZE calibration on a known 4 GiB stream (tools/calib_fetch.hip), each counter in its own pass;
#   2. tools/profile.sh r06        : default bench (separable, 256 x 24 MP fp32, levels 0 / 1 as a PAIR): kernel trace + stats, PMC groups;
#      tools/profile.sh r06_nopair : the same with the pair switched off (SHINESTACKER_AMD_PAIR_LEVELS=2): the level-by-level kernels;
#   3. tools/profile.sh r06_u8/u16 : the same stack held as 8- / 16-bit frames (the reference's input types), all PMC groups;
#   4. tools/profile.sh r06_exact  : --arith exact, kernel trace + stats + PMC;
#   5. gpurun_out/traffic.json with provenance (kernel sources' sha256, dtype, frames per launch) -- tools/promote_r06.sh copies it to
#      profiles/traffic.json, which bench.py reads (tests/test_records.py fails while its sha is not the sources');
#   6. kernel timelines of the 256-frame step and of rank 0's interleaved 32-frame shard;
#   7. interleaved A/B of the shipped pair plan against the level-by-level kernels (fp32), config 4 (non-chained, chained) + its kernel summary.
# Everything lands under gpurun_out/; tools/promote_r06.sh (run where the repository is) copies what is kept into profiles/r06/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/calib gpurun_out/r06
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib_fetch 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/calib/$c -o calib -- /tmp/calib_fetch > /dev/null 2>&1
done
tools/profile.sh r06 > /dev/null 2>&1
SHINESTACKER_AMD_PAIR_LEVELS=2 tools/profile.sh r06_nopair > /dev/null 2>&1
tools/profile.sh r06_u8 --dtype u8 > /dev/null 2>&1
tools/profile.sh r06_u16 --dtype u16 > /dev/null 2>&1
tools/profile.sh r06_exact --arith exact > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/prof_r06 gpurun_out/calib "level_sep_pair<float, true" > gpurun_out/traffic_sep.json
python tools/pmc_traffic.py gpurun_out/prof_r06_exact gpurun_out/calib "level_fused<float, true, true, 32, 64" > gpurun_out/traffic_exact.json
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
out = {}
for k, f in (("separable", "gpurun_out/traffic_sep.json"), ("exact", "gpurun_out/traffic_exact.json")):
    e = json.load(open(f))
    e.update(source_sha=bench.kernel_source_sha(), dtype="f32", frames_per_launch=16,
             note="r06: one launch = 16 frames of a 256-frame resident push of 4000x6000x3 fp32 frames (tools/profile_r06.sh); "
                  "average over the profiled launches; FETCH_SIZE doubled per the gfx950 calibration, WRITE_SIZE as reported")
    out[k] = e
json.dump(out, open("gpurun_out/traffic.json", "w"), indent=1)
print(json.dumps({k: (v["kernel"], v["hbm_bytes_per_launch"]) for k, v in out.items()}))
PY
tools/timeline_run.sh r06
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_r06_shard -o t -- python tools/shard_step.py 8 256 --trace > gpurun_out/r06/shard_step.txt 2>&1
python tools/timeline_all.py gpurun_out/tl_r06_shard/t_kernel_trace.csv 4.9 > gpurun_out/tl_r06_shard/timeline.txt
{
  echo "# interleaved A/B on ONE box: the shipped plan (SHINESTACKER_AMD_PAIR_LEVELS=0: float-32, >= 192 frames: levels 0 / 1 as a pair) against the level-by-level kernels (=2); 256 x 24 MP fp32"
  tools/ab.sh 3 "SHINESTACKER_AMD_PAIR_LEVELS=2" "SHINESTACKER_AMD_PAIR_LEVELS=0" -- --steps 5 --warmup 1 --dtype f32
} > gpurun_out/r06/pair_ab_final.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_r06_c4 -o t -- python tools/config4.py --frames 128 --resident --arith separable --reuse-handles > /dev/null 2>&1
python tools/timeline_all.py gpurun_out/tl_r06_c4/t_kernel_trace.csv 45 --summary > gpurun_out/r06/config4_timeline_summary.txt 2>&1
python tools/config4.py --frames 128 --resident --arith separable --reuse-handles > gpurun_out/r06/config4_resident_separable.json 2> /dev/null
python tools/config4.py --frames 128 --resident --arith separable --step-process --reuse-handles > gpurun_out/r06/config4_resident_step_refined.json 2> /dev/null
python tools/config4.py --frames 128 --resident --arith separable --step-process --chain-serial > gpurun_out/r06/config4_resident_step_serial.json 2> /dev/null
python bench.py --traffic-json gpurun_out/traffic.json > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err   # (the file promote_r06.sh copies to profiles/traffic.json)
for t in r06 r06_nopair r06_u8 r06_u16 r06_exact; do echo "=== $t"; head -45 gpurun_out/prof_$t/summary.txt; done
