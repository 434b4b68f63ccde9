"""Alone on the GPU: mi_warp_affine_device on a 24 MP frame for a few transforms, with and without the blurred border.
   python tools/warp_time.py [--dtype u8|u16]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shinestacker_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="u8")
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--only", default="", help="run only the transform whose name contains this")
ap.add_argument("--no-blur", action="store_true")
a = ap.parse_args()
dt = np.uint8 if a.dtype == "u8" else np.uint16
H, W = 4000, 6000
fb = H * W * 3 * np.dtype(dt).itemsize
src, dst, tmp, mask = L.DeviceBuffer(fb), L.DeviceBuffer(fb), L.DeviceBuffer(fb), L.DeviceBuffer(H * W)
L.synth_frames_device(src.ptr, dt, H, W, 0, 1, 4)
lib = L.load()
bv = (C.c_double * 4)(0, 0, 0, 0)
cx, cy = (W - 1) / 2, (H - 1) / 2
for name, (deg, s, tx, ty) in {"shift 3.4/-2.2": (0, 1, 3.4, -2.2), "0.2 deg": (0.2, 1.001, 5, -3), "1.3 deg": (1.3, 1.006, 24, -13),
                               "5 deg": (5, 1.0, 0, 0), "30 deg": (30, 1.0, 0, 0),
                               # no tile's source window leaves the frame with one of these two (no per-pixel ring tiles):
                               "zoom 1.02": (0, 1.02, 0, 0), "zoom 0.98": (0, 0.98, 0, 0)}.items():
    if a.only and a.only not in name:
        continue
    t = np.deg2rad(deg)
    ca, sa = s * np.cos(t), s * np.sin(t)
    M = (C.c_double * 6)(ca, -sa, cx - ca * cx + sa * cy + tx, sa, ca, cy - sa * cx - ca * cy + ty)
    for mode, mname in ((1, "replicate"), (2, "replicate+blur"))[:1 if a.no_blur else 2]:
        def run():
            L.check(lib.mi_warp_affine_device(0, None, src.ptr, dst.ptr, tmp.ptr, mask.ptr, H, W, L.DTYPE_CODE[np.dtype(dt)], M, mode,
                                              bv, 21, 50.0))
        run()
        lib.mi_device_synchronize(0)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            run()
        lib.mi_device_synchronize(0)
        us = (time.perf_counter() - t0) / a.reps * 1e6
        print(f"{a.dtype} {name:16s} {mname:15s} {us:8.1f} us   {7 * H * W * np.dtype(dt).itemsize / us / 1e6:6.2f} TB/s (src + dst + mask)", flush=True)
