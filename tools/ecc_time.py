import sys, time, numpy as np
sys.path.insert(0, '.')
from shinestacker_amd import _lib as L
H, W, N = 4000, 6000, 33
fb = H*W*3
buf = L.DeviceBuffer(N*fb)
L.synth_frames_device(buf.ptr, np.uint8, H, W, 0, N, N)
# smooth-ish content: reuse frame 0 for all (identity transforms) - timing only
al = L.Aligner(H, W, np.uint8, subsample=2, fast=("--area" not in sys.argv))
al.set_reference(buf.ptr)
ptrs = [buf.ptr + (k+1)*fb for k in range(32)]
for nb in (1, 4, 16):
    al.estimate_batch(ptrs[:nb])  # warm (allocation)
    t0 = time.perf_counter()
    tot_it = 0
    for rep in range(0, 32, nb):
        ms, cc, it = al.estimate_batch(ptrs[rep:rep+nb])
        tot_it += int(it.sum())
    dt = time.perf_counter() - t0
    print(f"batch {nb:2d}: {dt/32*1e3:.3f} ms per frame, {tot_it/32:.1f} iterations per frame, cc {cc[0]:.3f}")
for nb in (1, 4, 16):
    t0 = time.perf_counter()
    a2 = L.Aligner(H, W, np.uint8, subsample=2)
    a2.set_reference(buf.ptr)
    t1 = time.perf_counter()
    a2.estimate_batch(ptrs[:nb])
    t2 = time.perf_counter()
    a2.estimate_batch(ptrs[:nb])
    t3 = time.perf_counter()
    a2.close()
    t4 = time.perf_counter()
    print(f"batch {nb:2d}: create+ref {1e3*(t1-t0):.1f} ms, first batch {1e3*(t2-t1):.1f} ms, warm batch {1e3*(t3-t2):.1f} ms, close {1e3*(t4-t3):.1f} ms")
