"""The step of ONE rank of an N-GPU job on its shard of the 256-frame stack, alone on this GPU: interleaved shard (frames 0, N,
2N, ...: what bench.py --shards interleaved deals) against the contiguous block (frames 0 .. 256/N - 1, rounds 1-5).
    python tools/shard_step.py [ranks] [total] [--trace]      (--trace: one extra step of the interleaved shard, for rocprofv3)"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from shinestacker_amd import _lib as L

H, W = 4000, 6000
N = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 256
F = T // N
per = H * W * 3 * 4


def run(buf, label, stride, steps=40):
    st = L.Stack(H, W, in_dtype=np.float32, out_dtype=np.uint8, arith="separable")

    def step():
        st.reset()
        st.set_first_index(0, stride)
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
    if "--trace" in sys.argv:
        step()
        st.sync()
    st.close()


buf = L.DeviceBuffer(per * F)
L.synth_frames_device(buf.ptr, np.float32, H, W, 0, F, T)
run(buf, f"contiguous : frames 0 .. {F - 1} of {T}", 1)
L.synth_frames_device(buf.ptr, np.float32, H, W, 0, F, T, frame_step=N)
run(buf, f"interleaved: frames 0, {N}, {2 * N}, ... of {T}", N)
