#!/bin/bash
# Phase ablation of the border kernels alone (MI_ABLATE bit 512 skips every interior launch).
for ab in 512 513 516 520 528 525 541; do
  echo -n "ablate $ab: "
  MI_ABLATE=$ab python bench.py --no-cpu-baseline --frames 32 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['breakdown_ms_per_step'])"
done
