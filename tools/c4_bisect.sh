#!/bin/bash
# config 4 across worktrees of this repository (_r05tree, _wt_<commit>, built there), two rounds
cd "$(dirname "$0")/.."
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.4f s" % d["seconds"])'
for i in 1 2; do
  for t in _r05tree _wt_bb8ecfa _wt_91bae35 _wt_715e59f _wt_13e3147 .; do
    [ -d $t ] || continue
    echo -n "$t: "; (cd $t && python tools/config4.py --frames 128 --resident --reuse-handles --arith separable 2>/dev/null | python -c "$P")
  done
done
