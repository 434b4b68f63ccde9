import sys, time
import numpy as np
sys.path.insert(0, ".")
from shinestacker_amd import _lib as L
H, W, N = 4000, 6000, 16
fb = H * W * 3
buf = L.DeviceBuffer(fb * N)
L.synth_frames_device(buf.ptr, np.uint8, H, W, 0, N, N)
def T(label, f, n=1):
    L.load().mi_device_synchronize(0)
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    L.load().mi_device_synchronize(0)
    print(f"{label}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms", flush=True)
    return r
al = T("Aligner create (sub 2)", lambda: L.Aligner(H, W, np.uint8, subsample=2))
al2 = T("Aligner create again", lambda: L.Aligner(H, W, np.uint8, subsample=2))
T("set_reference", lambda: al.set_reference(buf.ptr), 5)
T("estimate (1 frame)", lambda: al.estimate(buf.ptr + fb), 5)
ms, cs, _ = T("estimate_batch (15 frames)", lambda: al.estimate_batch([buf.ptr + i * fb for i in range(1, N)]))
T("refine_batch (15 frames, 2 levels)", lambda: al.refine_batch([buf.ptr + i * fb for i in range(1, N)], ms, levels=2), 3)
T("Aligner close", lambda: al2.close())
