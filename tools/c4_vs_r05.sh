#!/bin/bash
# same-box A/B of config 4: this tree against a worktree of the round-5 head at _r05tree (git worktree add _r05tree cde44e5; build there)
cd "$(dirname "$0")/.."
P='import json,sys; d=json.loads(sys.stdin.read()); print("%.4f s  shift %.4f px" % (d["seconds"], d["worst_error"]["shift_px"]))'
for i in 1 2 3; do
  echo -n "r05: "; (cd _r05tree && python tools/config4.py --frames 128 --resident --reuse-handles --arith separable 2>/dev/null | python -c "$P")
  echo -n "now: "; python tools/config4.py --frames 128 --resident --reuse-handles --arith separable 2>/dev/null | python -c "$P"
done
