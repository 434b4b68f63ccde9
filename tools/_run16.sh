B="python bench.py --no-cpu-baseline --no-other-mode --no-verify --steps 2 --warmup 1"
V=$PWD/shinestacker_amd/csrc/libmi355stack_pc.so
for dt in f32 u8 u16; do echo "== $dt"; MI355STACK_LIB=$V $B --dtype $dt 2>&1 >/dev/null | grep "^wave" | tail -8; done
