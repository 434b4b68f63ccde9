"""Chained alignment on a simulated FOCUS stack (tools/parity_report.py::defocus_frames: every frame blurred by its distance from
the focal plane, so the global reference frame and a far frame look different) with a known similarity per frame: worst
centre shift error of (a) the fixed-reference order, (b) the plain chain, (c) the chain refined against the global reference
frame.   python tools/chain_defocus.py [frames] [height] [width]"""
import sys
import numpy as np
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import parity_report as pr
from shinestacker_amd import _lib as L
from shinestacker_amd.pipeline import align_and_stack_device

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
frames = pr.defocus_frames(H, W, N, np.uint8)
ref = N // 2
cx, cy = (W - 1) / 2, (H - 1) / 2
truth, moved = [], []
for f in range(N):
    d = f - ref
    t = np.deg2rad(0.03 * d)
    s = 1 + 4e-4 * d          # focus breathing: the magnification changes with the focus position
    a, b = s * np.cos(t), s * np.sin(t)
    T = np.array([[a, -b, cx - a * cx + b * cy + 0.8 * d], [b, a, cy - b * cx - a * cy - 0.5 * d]])
    truth.append(T)
    moved.append(frames[f] if d == 0 else L.warp_affine(frames[f], T, border_mode=L.BORDER_REPLICATE))
fb = H * W * 3
buf = L.DeviceBuffer(N * fb)
for f, fr in enumerate(moved):
    buf.upload(fr, f * fb)


def worst(tr):
    w = 0.0
    for f in range(N):
        if f == ref:
            continue
        A = truth[f][:, :2]
        Ai = np.linalg.inv(A)
        want = np.hstack([Ai, -Ai @ truth[f][:, 2:3]])
        c = np.array([cx, cy, 1.0])
        corners = np.array([[0, 0, 1.0], [W - 1, 0, 1], [0, H - 1, 1], [W - 1, H - 1, 1]]).T
        w = max(w, np.abs(np.asarray(tr[f])[:2] @ corners - want @ corners).max())
    return w


for name, kw in (("fixed reference", dict(step_process=False)), ("chain, plain", dict(step_process=True, chain_refine=False)),
                 ("chain, refined", dict(step_process=True, chain_refine=True))):
    _, tr, ccs = align_and_stack_device(buf.ptr, N, H, W, np.uint8, ref_idx=ref, alignment_config={'subsample': 2}, **kw)
    print(f"{name:16s}: worst corner error {worst(tr):.3f} px, lowest correlation {min(ccs):.3f}", flush=True)
