#!/bin/bash
# After tools/profile_r06.sh ran on the GPU box (gpurun merges gpurun_out/ back): copy what is kept into the tracked tree.
# LAST step of a profiling session -- any later edit of the kernel sources voids profiles/traffic.json (tests/test_records.py).
cd "$(dirname "$0")/.."
mkdir -p profiles/r06
cp gpurun_out/traffic.json profiles/traffic.json
for t in r06 r06_nopair r06_u8 r06_u16 r06_exact; do
  s=${t#r06}; s=${s#_}; s=${s:+_$s}
  cp gpurun_out/prof_$t/summary.txt profiles/r06/rocprofv3_summary$s.txt
  f=$(ls gpurun_out/prof_$t/stats/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/r06/kernel_stats$s.csv
done
cp gpurun_out/calib/FETCH_SIZE/*counter_collection.csv profiles/r06/calib_FETCH_SIZE.csv 2>/dev/null
cp gpurun_out/calib/WRITE_SIZE/*counter_collection.csv profiles/r06/calib_WRITE_SIZE.csv 2>/dev/null
cp gpurun_out/tl_r06/timeline.txt profiles/r06/timeline.txt
cp gpurun_out/tl_r06_shard/timeline.txt profiles/r06/timeline_32_frames_interleaved.txt
for f in pair_ab_final.txt config4_resident_separable.json config4_resident_step_refined.json config4_resident_step_serial.json \
         config4_timeline_summary.txt bench_default.json; do
  cp gpurun_out/r06/$f profiles/r06/$f
done
grep -v "^[EWI][0-9]\{8\}" gpurun_out/r06/shard_step.txt > profiles/r06/shard_step.txt   # (without rocprofv3's own log lines)
ls profiles/r06
