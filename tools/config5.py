#!/usr/bin/env python3
"""BASELINE config 5 on one GPU at reduced length: 16-bit 50 MP frames in HOST memory, stacked in
bunches (FocusStackBunch geometry: frames=10, overlap=2, stack.py:61-64) through the pinned asynchronous
upload path of mi_stack_push_frame -- PCIe, the bounce copy and the kernels overlap.  One stacker handle
serves every bunch (reset between bunches), as FocusStackBunch does."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--height", type=int, default=5792)
    ap.add_argument("--width", type=int, default=8640)
    args = ap.parse_args()
    from shinestacker_amd import _lib as L
    from shinestacker_amd.actions import get_bunches
    N, H, W = args.frames, args.height, args.width
    per = H * W * 3 * 2
    ndist = 8   # distinct host frames, cycled
    buf = L.DeviceBuffer(per * ndist)
    L.synth_frames_device(buf.ptr, np.uint16, H, W, 0, ndist, ndist)
    host = [buf.download((H, W, 3), np.uint16, offset=i * per) for i in range(ndist)]
    bunches = get_bunches(list(range(N)), 10, 2)
    st = L.Stack(H, W, in_dtype=np.uint16, out_dtype=np.uint16)
    out = L.DeviceBuffer(per)

    def run():
        pushed = 0
        for b in bunches:
            st.reset()
            for f in b:
                st.push_frame(host[f % ndist])
                pushed += 1
            st.finish_device(out.ptr)
        st.sync()
        return pushed
    run()
    t0 = time.perf_counter()
    pushed = run()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": f"{N} x {W}x{H} u16 frames from host memory, {len(bunches)} bunches of <= 10 (overlap 2)",
                      "frames_pushed": pushed, "seconds": dt, "Mpixels_per_s": pushed * H * W / dt / 1e6,
                      "host_to_device_GB_per_s": pushed * per / dt / 1e9}))


if __name__ == "__main__":
    main()
