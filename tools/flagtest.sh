cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
for fl in "" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy" "-mllvm -amdgpu-early-inline-all=true"; do
  MI_EXTRA_FLAGS="$fl" python -m shinestacker_amd.build --force > /dev/null 2>&1 || { echo "[$fl] build failed"; continue; }
  echo -n "[$fl] "
  python bench.py --frames 128 --steps 3 --warmup 1 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f Gpx/s  %.2f ms/step' % (d['value']/1e3, d['ms_per_step']))"
done
python -m shinestacker_amd.build --force > /dev/null 2>&1
