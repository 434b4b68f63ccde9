"""Time DepthMapStack on the device: N synthetic frames resident in HBM, push (energy) and finish (smoothing,
weights, blend) timed separately.  python tools/depthmap_time.py [--frames 32] [--dtype u8] [--map max] ..."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shinestacker_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--height", type=int, default=4000)
ap.add_argument("--width", type=int, default=6000)
ap.add_argument("--dtype", default="u8")
ap.add_argument("--map", default="average")
ap.add_argument("--energy", default="laplacian")
ap.add_argument("--smooth", type=int, default=15)
ap.add_argument("--levels", type=int, default=3)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dt = np.uint8 if a.dtype == "u8" else np.uint16
H, W, N = a.height, a.width, a.frames
fb = H * W * 3 * np.dtype(dt).itemsize
buf = L.DeviceBuffer(fb * (N + 1))
L.synth_frames_device(buf.ptr, dt, H, W, 0, N, N, 20250824)
dm = L.DepthMap(H, W, dtype=dt, map_type={"average": 0, "max": 1}[a.map], energy={"laplacian": 0, "sobel": 1}[a.energy],
                smooth_size=a.smooth, levels=a.levels)
best = None
for rep in range(a.reps + 1):
    dm.reset()
    t0 = time.perf_counter()
    for i in range(N):
        dm.push_frame_device(buf.ptr + i * fb)
    t1 = time.perf_counter()
    dm.finish_device(buf.ptr + N * fb)
    t2 = time.perf_counter()
    if rep and (best is None or t2 - t0 < best[0]):
        best = (t2 - t0, t1 - t0, t2 - t1)
print(json.dumps({"config": f"{N}x{W}x{H} {a.dtype} depth map ({a.energy}, {a.map}, smooth {a.smooth}, levels {a.levels})",
                  "seconds": best[0], "push_s": best[1], "finish_s": best[2],
                  "Mpixels_per_s": N * H * W / best[0] / 1e6, "ms_per_frame": best[0] / N * 1e3}))
