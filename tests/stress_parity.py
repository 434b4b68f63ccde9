#!/usr/bin/env python3
"""One-off randomised differential stress: many random shapes / dtypes / batch sizes / options through
the tiled (default) path against the streaming oracle.  Not part of the test suite (minutes on a GPU box):
    python tests/stress_parity.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    from oracle import oracle
    from shinestacker_amd import _lib as L
    from test_gpu_fuzz import make_frames
    oracle.build()
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
    bad = 0
    for case in range(n_cases):
        h = int(rng.integers(40, 700))
        w = int(rng.integers(40, 900))
        if case % 7 == 0:   # widths around the tile / border arithmetic
            w = int(rng.choice([64 + 6 + 32, 32 + 64 + 5, 32 + 64 + 6, 32 + 64 + 7, 32 + 128 + 6, 134, 166, 198, 230]))
        dt = [np.uint8, np.uint16][int(rng.integers(0, 2))]
        n = int(rng.integers(1, 9))
        kw = {"min_size": int(rng.choice([8, 16, 32])), "use_fma": bool(rng.integers(0, 2)),
              "kernel_size": int(rng.choice([3, 5, 7])),
              # generating kernels with integer reduce taps (0.4, 0.3, 0.5: 20 k integral) and without (separable spec v2)
              # ... a negative integer outer tap (0.7: -2 5 14), taps too large for the integer form (6.7: -62 5 134: integral,
              # over the bound)
              "gen_kernel": float(rng.choice([0.4, 0.4, 0.3, 0.5, 0.35, 0.375, 0.7, 6.7]))}
        batch = int(rng.integers(0, 5))
        frames = make_frames(rng, (h, w), dt, n)
        arith = ["exact", "separable"][case % 2]
        if case % 5 == 0:   # long resident pushes: frame chunks + merges (duplicates: ties across chunks)
            frames = [frames[int(k)] for k in rng.integers(0, len(frames), int(rng.integers(33, 80)))]
        so = oracle.StreamingOracle(h, w, dt, keep_gauss=False, arith=arith, **kw)
        for f in frames:
            so.push_frame(f)
        want = so.finish()
        for impl in (L.IMPL_TILED,):
            st = L.Stack(h, w, in_dtype=dt, impl=impl, batch_frames=batch, arith=arith, **kw)
            if case % 5 == 0:
                fbytes = frames[0].nbytes
                buf = L.DeviceBuffer(fbytes * len(frames))
                for i, f in enumerate(frames):
                    buf.upload(f, i * fbytes)
                st.push_frames_device(buf.ptr, len(frames), fbytes)
                st.sync()
                buf.free()
            else:
                for f in frames:
                    st.push_frame(f)
            ok = all(np.array_equal(st.tap(L.TAP_ENERGY, lv), so.best_e[lv]) and
                     np.array_equal(st.tap(L.TAP_FUSED_LAP, lv), so.best_lap[lv]) for lv in range(st.levels))
            got = st.finish()
            ok = ok and np.array_equal(got, want)
            st.close()
            if not ok:
                bad += 1
                print("MISMATCH", case, impl, arith, h, w, dt.__name__, len(frames), kw, batch)
    print(f"{n_cases} cases (exact / separable alternating, every fifth a long resident push): {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
