"""Why does a 32-frame step cost 4.7 ms alone and 5.8 ms inside bench.py's projection?  Same handle, same launches:
(a) a 32-frame buffer holding a 32-frame synthetic stack, (b) the first 32 frames of a 256-frame stack in a 256-frame buffer,
(c) a 32-frame stack written into the 256-frame buffer.   python tools/shard_step.py [frames] [total]"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from shinestacker_amd import _lib as L

H, W = 4000, 6000
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
per = H * W * 3 * 4


def run(buf, label, steps=40):
    st = L.Stack(H, W, in_dtype=np.float32, out_dtype=np.uint8, arith="separable")
    def step():
        st.reset()
        st.push_frames_device(buf.ptr, F)
        st.finish_device()
    for _ in range(3):
        step()
    st.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    st.sync()
    print(f"{label}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step", flush=True)
    st.close()


small = L.DeviceBuffer(per * F)
L.synth_frames_device(small.ptr, np.float32, H, W, 0, F, F)
run(small, f"(a) {F}-frame buffer, {F}-frame stack")
big = L.DeviceBuffer(per * T)
L.synth_frames_device(big.ptr, np.float32, H, W, 0, T, T)
run(big, f"(b) {T}-frame buffer, first {F} frames of a {T}-frame stack")
run(small, f"(a) again, with the {T}-frame buffer allocated")
L.synth_frames_device(big.ptr, np.float32, H, W, 0, F, F)
run(big, f"(c) {T}-frame buffer, {F}-frame stack")
