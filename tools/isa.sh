#!/bin/bash
# Device ISA of the library -> /tmp/capi.s ; then: tools/isa.sh <mangled-or-demangled substring> prints scratch traffic,
# barriers and loop structure of the first kernel whose demangled name contains the substring.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC \
  -fvisibility=hidden -I include -S --cuda-device-only "${@:2}" shinestacker_amd/csrc/capi.hip -o /tmp/capi.s 2>/dev/null
python3 - "$1" <<'PY'
import re, subprocess, sys
s = open('/tmp/capi.s').read()
names = re.findall(r'^(_Z\w+):\s*; @', s, re.M)
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
for n, d in zip(names, dem):
    if sys.argv[1] in d:
        i = s.index(n + ':'); j = s.index('s_endpgm', i)
        body = s[i:j].split('\n')
        print(d, len(body), 'lines')
        inloop = False
        for k, l in enumerate(body):
            if re.match(r'\.LBB\d+_\d+:', l):
                inloop = 'Loop' in l
            if 'scratch_' in l:
                print(k, 'LOOP' if inloop else '    ', l.strip())
        print('barriers:', sum('s_barrier' in l for l in body), ' valu(v_):', sum(l.strip().startswith('v_') for l in body),
              ' ds:', sum(l.strip().startswith('ds_') for l in body))
        break
PY
