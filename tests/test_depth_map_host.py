"""CPU: the DepthMapStack oracle (oracle/depth_map_oracle.py) against the recordings of the reference's own
DepthMapStack.focus_stack (tests/golden/depth_map.npz, made by oracle/gen_golden.py --only-depth-map over the
cv2 shim), its primitives' invariants, and the host-side mirror's option handling."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import depth_map_oracle as dmo


def golden_cases():
    with open(os.path.join(GOLDEN, "depth_map.json")) as fh:
        meta = json.load(fh)
    g = load_golden("depth_map")
    for name, m in meta.items():
        yield name, m, list(g[name + "_frames"]), g[name + "_out"], g.get(name + "_undefined")


def test_oracle_reproduces_reference_recordings():
    n = 0
    for name, m, frames, out, undefined in golden_cases():
        mine = dmo.depth_map_stack(frames, **m["kwargs"])
        assert mine.dtype == out.dtype and mine.shape == out.shape
        keep = slice(None) if undefined is None else ~undefined
        assert np.array_equal(mine[keep], out[keep]), name
        assert m["trace_len"] == 4 * len(frames)          # after_step + check_running, two loops
        assert m["trace_head"][0][0] == "after_step" and m["trace_head"][1][0] == "check_running"
        n += 1
    assert n >= 9


def test_derivative_kernels():
    assert np.array_equal(dmo.sobel_kernels(1, 0, 3)[0], [-1, 0, 1])
    assert np.array_equal(dmo.sobel_kernels(1, 0, 3)[1], [1, 2, 1])
    assert np.array_equal(dmo.sobel_kernels(2, 0, 5)[0], [1, 0, -2, 0, 1])
    assert np.array_equal(dmo.sobel_kernels(2, 0, 5)[1], [1, 4, 6, 4, 1])
    for k in (1, 3, 5, 7, 9):
        assert dmo.laplacian_kernel2d(k).sum() == 0          # a constant image has no Laplacian
    for k in (1, 3, 5, 7, 9, 15):
        g = dmo.gaussian_kernel_f32(k)
        assert g.dtype == np.float32 and abs(float(g.sum()) - 1) < 1e-6 and np.array_equal(g, g[::-1])


def test_default_energy_is_exact_in_any_order():
    """Default parameters: every intermediate of |Laplacian(GaussianBlur(gray))| is exactly representable, so the
    float32 separable blur equals the float64 2-D one -- the energies do not depend on OpenCV's operation order."""
    rng = np.random.default_rng(3)
    for hi in (256, 65536):
        gray = rng.integers(0, hi, (40, 57)).astype(np.float32)
        sep = dmo.gaussian_blur_f32(gray, 5).astype(np.float64)
        k = np.array([1, 4, 6, 4, 1], np.float64) / 16
        full = dmo.filter2d_f64(gray, np.outer(k, k))
        assert np.array_equal(sep, full)


def test_pyramid_primitives_shapes_and_constants():
    rng = np.random.default_rng(4)
    for h, w in ((45, 70), (1, 9), (8, 1), (2, 2), (33, 33)):
        a = rng.random((h, w, 3)).astype(np.float32)
        d = dmo.pyr_down(a)
        assert d.shape == ((h + 1) // 2, (w + 1) // 2, 3) and d.dtype == np.float32
        u = dmo.pyr_up(d, (w, h))
        assert u.shape == a.shape
        c = np.full((h, w), 3.5, np.float32)                  # constants survive both directions
        assert np.array_equal(dmo.pyr_down(c), np.full(d.shape[:2], 3.5, np.float32))
        assert np.array_equal(dmo.pyr_up(dmo.pyr_down(c), (w, h)), c)
    with pytest.raises(AssertionError):
        dmo.pyr_up(np.zeros((4, 4), np.float32), (11, 8))


def test_bilateral_constant_and_bounds():
    c = np.full((20, 30), 0.25, np.float32)
    assert np.array_equal(dmo.bilateral_f32(c, 15), c)
    rng = np.random.default_rng(5)
    e = rng.random((31, 44)).astype(np.float32)
    s = dmo.bilateral_f32(e, 15)
    assert s.dtype == np.float32 and s.min() >= e.min() and s.max() <= e.max()   # a weighted mean
    assert s.std() < 0.5 * e.std()


def test_identical_frames_fuse_to_the_frame():
    rng = np.random.default_rng(6)
    f = rng.integers(0, 256, (36, 52, 3)).astype(np.uint8)
    for kw in ({}, {"map_type": "max"}, {"energy": "sobel"}):
        out = dmo.depth_map_stack([f, f.copy(), f.copy()], **kw)
        assert np.abs(out.astype(int) - f.astype(int)).max() <= 1, kw


def test_mirror_options():
    from shinestacker_amd import DepthMapStack, InvalidOptionError, constants
    d = DepthMapStack()
    assert d.map_type == constants.DEFAULT_DM_MAP == "average" and d.energy == constants.DEFAULT_DM_ENERGY == "laplacian"
    assert (d.kernel_size, d.blur_size, d.smooth_size, d.temperature, d.levels) == (5, 5, 15, 0.1, 3)
    assert d.name() == "depth map" and d.steps_per_frame() == 2
    with pytest.raises(InvalidOptionError):
        DepthMapStack(float_type="float-16")
    assert DepthMapStack(float_type=constants.FLOAT_64).float_type is np.float64
