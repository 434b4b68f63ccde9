"""Separable mode on the GPU: tiled kernel vs the one-thread-per-output kernels (all taps, bit for bit), then
timings of both arithmetic modes on a resident stack.   python tools/sep_check.py [--frames 64] [--skip-check]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shinestacker_amd import _lib as L  # noqa: E402
from shinestacker_amd import build  # noqa: E402


def taps(st):
    out = {}
    for l in range(st.levels):
        out[f"lap{l}"] = st.tap(L.TAP_FUSED_LAP, l)
        out[f"e{l}"] = st.tap(L.TAP_ENERGY, l)
        out[f"i{l}"] = st.tap(L.TAP_INDEX, l)
    for l in range(1, st.levels + 1):
        out[f"g{l}"] = st.tap(L.TAP_GAUSS, l)
    return out


def check(H, W, N, dt, seed=0, min_size=32, batch=0):
    rng = np.random.default_rng(seed)
    hi = 65535 if dt == np.uint16 else 255
    frames = [rng.integers(0, hi + 1, (H, W, 3)).astype(dt) for _ in range(N)]   # float32: integer values
    if N > 2:
        frames[2] = frames[0].copy()   # a tie
    res = []
    for impl in (L.IMPL_SIMPLE, L.IMPL_TILED):
        st = L.Stack(H, W, in_dtype=dt, out_dtype=np.uint16 if dt == np.uint16 else np.uint8, impl=impl,
                     arith="separable", min_size=min_size, batch_frames=batch)
        for f in frames:
            st.push_frame(f)
        img = st.finish()
        t = taps(st)
        t["img"] = img
        res.append(t)
        st.close()
    bad = 0
    for k in res[0]:
        a, b = res[0][k], res[1][k]
        if not np.array_equal(a, b):
            d = np.argwhere(a != b)
            bad += 1
            print(f"  MISMATCH {k}: {len(d)} of {a.size} differ, first at {d[0]}, rows {d[:,0].min()}..{d[:,0].max()} "
                  f"cols {d[:,1].min()}..{d[:,1].max()}  ({a[tuple(d[0])]} vs {b[tuple(d[0])]})")
    print(f"check {H}x{W} N={N} {np.dtype(dt).name} min_size={min_size}: {'OK' if not bad else 'FAILED'}", flush=True)
    return bad == 0


def timing(frames, H, W, dt, arith, steps=3):
    per = H * W * 3 * np.dtype(dt).itemsize
    buf = L.DeviceBuffer(per * frames)
    L.synth_frames_device(buf.ptr, dt, H, W, 0, frames, frames)
    st = L.Stack(H, W, in_dtype=dt, out_dtype=np.uint8, arith=arith)

    def step():
        st.reset()
        st.push_frames_device(buf.ptr, frames)
        st.finish_device()
    step()
    st.sync()
    st.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    st.sync()
    dt_s = (time.perf_counter() - t0) / steps
    ms0, n0, b0 = st.profile_get(L.PROF_LEVEL0)
    msl, nl, bl = st.profile_get(L.PROF_LEVEL)
    print(f"{arith:10s} {frames} x {W}x{H} {np.dtype(dt).name}: {dt_s*1e3:8.2f} ms/stack = {frames*H*W/dt_s/1e9:7.1f} Gpx/s; "
          f"level0 {ms0/max(n0,1):.3f} ms/launch ({n0//steps} launches/stack, {b0/max(ms0,1e-9)/1e6:.0f} GB/s algorithmic), "
          f"other levels {msl/steps:.2f} ms/stack", flush=True)
    st.close()
    buf.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--skip-timing", action="store_true")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--arith", default="exact,separable")
    a = ap.parse_args()
    build.build_extension()
    print(L.device_name(0))
    ok = True
    if not a.skip_check:
        for (H, W, N, dt, ms) in [(300, 452, 4, np.uint8, 32), (133, 201, 4, np.uint8, 8), (257, 130, 3, np.uint16, 16),
                                  (64, 64, 3, np.float32, 8), (500, 750, 5, np.float32, 32), (97, 1031, 3, np.uint8, 8),
                                  (1000, 1500, 34, np.uint8, 32)]:
            ok &= check(H, W, N, dt, min_size=ms)
    if not a.skip_timing:
        dt = {"u8": np.uint8, "u16": np.uint16, "f32": np.float32}[a.dtype]
        for arith in a.arith.split(","):
            timing(a.frames, 4000, 6000, dt, arith)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
