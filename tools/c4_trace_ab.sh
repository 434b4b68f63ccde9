mkdir -p gpurun_out/r06
export TMPDIR=/tmp
for t in r05 now; do
  D=$GRAFT_REPO_ROOT; [ $t = r05 ] && D=$GRAFT_REPO_ROOT/_r05tree
  (cd $D && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$t -o t -- python tools/config4.py --frames 128 --resident --reuse-handles --arith separable > /dev/null 2>&1)
  F=$(find /tmp/tr_$t -name "*kernel_trace.csv" | head -1)
  python tools/timeline_buckets.py $F 62 2 > gpurun_out/r06/c4_buckets_$t.txt
  python tools/timeline_all.py $F 62 --summary > gpurun_out/r06/c4_summary_$t.txt
  python tools/timeline_all.py $F 62 > gpurun_out/r06/c4_timeline_$t.txt
done
