"""Time mi_warp_affine_device's kernels for a few transforms (run under rocprofv3 --kernel-trace --stats)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shinestacker_amd import _lib as L  # noqa: E402

H, W = 4000, 6000
lib = L.load()
nb = H * W * 3
src, dst, tmp, mask = (L.DeviceBuffer(nb) for _ in range(3)), None, None, None
src, dst, tmp = src
mask = L.DeviceBuffer(H * W)
L.synth_frames_device(src.ptr, np.uint8, H, W, 0, 1, 4, 7)
cases = {"identity": (1.0, 0.0, 0.0, 0.0), "xshift3": (1.0, 0.0, 3.3, 0.0), "yshift2": (1.0, 0.0, 0.0, -2.2),
         "shift3": (1.0, 0.0, 3.3, -2.2), "config4-like": (1.002, 0.0015, 9.0, -7.0),
         "rot0.5deg": (1.0, 0.0087, 0.0, 0.0)}
for name, (a, b, tx, ty) in cases.items():
    M = (C.c_double * 6)(a, -b, tx, b, a, ty)
    bv = (C.c_double * 4)(0, 0, 0, 0)
    for _ in range(3):
        L.check(lib.mi_warp_affine_device(0, None, src.ptr, dst.ptr, tmp.ptr, mask.ptr, H, W, L.MI_U8, M, 2, bv, 21, 50.0))
    m = mask.download((H, W), np.uint8)
    print(name, "masked pixels:", int((m == 0).sum()))
