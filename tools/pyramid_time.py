"""ecc_pyramid2 alone (the estimator's gray + two finest levels of one 24 MP frame):  rocprofv3 --kernel-trace --stats -- python tools/pyramid_time.py
(mi_aligner_set_reference builds exactly one pyramid)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from shinestacker_amd import _lib as L
H, W = 4000, 6000
for dt in (np.uint8, np.uint16):
    fb = H * W * 3 * np.dtype(dt).itemsize
    buf = L.DeviceBuffer(4 * fb)
    L.synth_frames_device(buf.ptr, dt, H, W, 0, 4, 4)
    for fast in (False, True):
        al = L.Aligner(H, W, dt, subsample=2, fast=fast)
        for k in range(40):
            al.set_reference(buf.ptr + (k % 4) * fb)
        al.close()
    buf.free()
