#!/usr/bin/env python3
"""End-to-end timing of PyramidStack.focus_stack on image FILES (decode + upload + fuse), with the
sequential decode of the reference's loop and with the decode-ahead thread pool."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from shinestacker_amd import PyramidStack
    from shinestacker_amd import _lib as L
    from shinestacker_amd.imageio import write_img
    n, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 12, 4000, 6000
    buf = L.DeviceBuffer(H * W * 3)
    with tempfile.TemporaryDirectory() as d:
        names = []
        for f in range(n):
            L.synth_frames_device(buf.ptr, np.uint8, H, W, f, 1, n)
            names.append(os.path.join(d, f"f{f:03d}.jpg"))
            write_img(names[-1], buf.download((H, W, 3), np.uint8))

        class Proc:
            id, name = 0, "e2e"

            def callback(self, *_a):
                return True

            def sub_message_r(self, *_a, **_k):
                pass
        for threads in (1, 4, 8, 16):
            algo = PyramidStack(decode_threads=threads)
            algo.process = Proc()
            algo.focus_stack(names[:2])   # warm-up (handle, codec)
            t0 = time.perf_counter()
            algo.focus_stack(names)
            dt = time.perf_counter() - t0
            print(f"decode_threads={threads:2d}: {n} x 24 MP JPEG files fused in {dt:.2f} s = {n * H * W / dt / 1e6:.0f} Mpixels/s")
            algo.close()


if __name__ == "__main__":
    main()
