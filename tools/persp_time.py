import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from shinestacker_amd import _lib as L
L.require_device()
H, W = 4000, 6000
fb = H * W * 3
src, dst, tmp, mask = L.DeviceBuffer(fb), L.DeviceBuffer(fb), L.DeviceBuffer(fb), L.DeviceBuffer(H * W)
L.synth_frames_device(src.ptr, np.uint8, H, W, 0, 1, 4)
lib = L.load()
bv = (C.c_double * 4)(0, 0, 0, 0)
M6 = (C.c_double * 6)(1.0001, -0.0003, 3.4, 0.0003, 1.0001, -2.2)
M9 = (C.c_double * 9)(1.0001, -0.0003, 3.4, 0.0003, 1.0001, -2.2, 0, 0, 1)
for name, fn, M in (("affine", lib.mi_warp_affine_device, M6), ("persp", lib.mi_warp_perspective_device, M9)):
    for mode in (1, 2):
        fn(0, None, src.ptr, dst.ptr, tmp.ptr, mask.ptr, H, W, 0, M, mode, bv, 21, C.c_double(50.0))
        lib.mi_device_synchronize(0)
        t0 = time.perf_counter()
        for _ in range(50):
            L.check(fn(0, None, src.ptr, dst.ptr, tmp.ptr, mask.ptr, H, W, 0, M, mode, bv, 21, C.c_double(50.0)))
        t1 = time.perf_counter()
        lib.mi_device_synchronize(0)
        t2 = time.perf_counter()
        print(f"{name} mode {mode}: host enqueue {1e6*(t1-t0)/50:.1f} us per call, total {1e6*(t2-t0)/50:.1f} us per call")
