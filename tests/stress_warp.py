#!/usr/bin/env python3
"""One-off randomised differential stress of the warp kernel (not part of the suite): random sizes (one to many tiles each
way, widths that are / are not multiples of four), dtypes, border modes and similarity transforms from sub-pixel shifts to
large rotations, against oracle/align_oracle.c -- bit-exact, mask included.
    python tests/stress_warp.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from oracle import oracle
    from shinestacker_amd import _lib as L
    oracle.build()
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
    bad = 0
    for case in range(n_cases):
        h = int(rng.integers(8, 400))
        w = int(rng.integers(8, 1400))
        if case % 3 == 0:
            w = (w + 3) // 4 * 4
        dt = [np.uint8, np.uint16][int(rng.integers(0, 2))]
        mode = int(rng.integers(0, 3))
        kind = case % 4
        deg = [0.0, float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-3, 3)), float(rng.uniform(-40, 40))][kind]
        s = float(rng.uniform(0.97, 1.03)) if kind else 1.0
        tx, ty = (float(rng.uniform(-30, 30)), float(rng.uniform(-20, 20))) if case % 5 else (float(rng.integers(-9, 9)), 0.5)
        t = np.deg2rad(deg)
        a, b = s * np.cos(t), s * np.sin(t)
        cx, cy = (w - 1) / 2, (h - 1) / 2
        M = np.array([[a, b, (1 - a) * cx - b * cy + tx], [-b, a, b * cx + (1 - a) * cy + ty]], dtype=np.float64)
        hi = 256 if dt == np.uint8 else 65536
        img = rng.integers(0, hi, (h, w, 3)).astype(dt)
        bv = (int(rng.integers(0, hi)), 3, int(rng.integers(0, hi)), 0)
        want, wmask = oracle.warp_affine(img, M, border_mode=mode, border_value=bv, want_mask=True)
        got, gmask = L.warp_affine(img, M, border_mode=mode, border_value=bv, want_mask=True)
        if not (np.array_equal(got, want) and np.array_equal(gmask, wmask)):
            bad += 1
            print(f"MISMATCH case {case}: {h}x{w} {np.dtype(dt).name} mode {mode} deg {deg:.3f} s {s:.4f} t ({tx:.2f}, {ty:.2f}): "
                  f"{int((got != want).sum())} values, {int((gmask != wmask).sum())} mask pixels")
    print(f"{n_cases} cases: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
