import sys, time
sys.path.insert(0, '.')
from shinestacker_amd import _lib as L
L.require_device()
for rep in range(2):
    for gb in (0.1, 0.5, 1, 2, 4, 8, 16):
        n = int(gb * (1 << 30))
        t0 = time.perf_counter(); b = L.DeviceBuffer(n); t1 = time.perf_counter(); b.free(); t2 = time.perf_counter()
        print(f"{gb:5.1f} GB: malloc {1e3*(t1-t0):8.2f} ms  free {1e3*(t2-t1):8.2f} ms")
import numpy as np
for bf in (16, 32):
    t0 = time.perf_counter(); st = L.Stack(4000, 6000, in_dtype=np.uint8, arith="separable", batch_frames=bf); t1 = time.perf_counter(); st.close(); t2 = time.perf_counter()
    print(f"Stack(24MP u8, batch {bf}): create {1e3*(t1-t0):.1f} ms, close {1e3*(t2-t1):.1f} ms")
