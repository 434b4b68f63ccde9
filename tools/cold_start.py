"""Cold vs warm: the first resident push on a fresh handle allocates (and first touches) its per-batch buffers."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from shinestacker_amd import _lib as L
L.require_device()
H, W, N = 4000, 6000, int(sys.argv[1]) if len(sys.argv) > 1 else 256
fb = H * W * 3 * 4
buf = L.DeviceBuffer(N * fb)
L.synth_frames_device(buf.ptr, np.float32, H, W, 0, N, N)
out = L.DeviceBuffer(H * W * 3)
lib = L.load()
lib.mi_device_synchronize(0)
for arith in ("separable", "exact"):
    t0 = time.perf_counter()
    st = L.Stack(H, W, in_dtype=np.float32, out_dtype=np.uint8, arith=arith)
    t1 = time.perf_counter()
    times = []
    for rep in range(3):
        st.reset()
        a = time.perf_counter()
        st.push_frames_device(buf.ptr, N, fb)
        st.finish_device(out.ptr)
        st.sync()
        times.append(time.perf_counter() - a)
    st.close()
    print(f"{arith}: create {1e3*(t1-t0):.1f} ms; stack 1 / 2 / 3 on the handle: " + " / ".join(f"{1e3*t:.1f} ms" for t in times))
