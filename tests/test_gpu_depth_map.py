"""GPU: DepthMapStack (mi_dmap_*, csrc/kernels_depthmap.hpp) against the reference recordings
(tests/golden/depth_map.npz), against oracle/depth_map_oracle.py on random stacks, and the plug-in protocol.

Tolerance (floating point, stated): the kernels keep the oracle's operation order, so results are expected to
be identical; the test allows the output to differ by at most 1 count on at most 0.1 % of the values -- the
double-precision exp behind the bilateral range table and the softmax may differ in its last bit between the
device library and the host's libm, which can move a value across a truncation boundary."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import depth_map_oracle as dmo

pytestmark = pytest.mark.gpu

MAP = {"average": 0, "max": 1}
ENERGY = {"laplacian": 0, "sobel": 1}


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


def close_enough(got, want, where=None):
    assert got.dtype == want.dtype and got.shape == want.shape
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    if where is not None:
        d = d[where]
    return d.max() <= 1 and (d != 0).mean() <= 1e-3, (int(d.max()), float((d != 0).mean()))


def run_gpu(L, frames, **kw):
    kw = dict(kw)
    h, w = frames[0].shape[:2]
    ft = L.MI_F64 if kw.pop("float_type", "float-32") == "float-64" else L.MI_F32
    with L.DepthMap(h, w, dtype=frames[0].dtype, map_type=MAP[kw.pop("map_type", "average")],
                    energy=ENERGY[kw.pop("energy", "laplacian")], float_type=ft, **kw) as dm:
        for f in frames:
            dm.push_frame(f)
        assert dm.frames_pushed == len(frames)
        return dm.finish()


def test_golden_recordings_of_the_reference(L):
    with open(os.path.join(GOLDEN, "depth_map.json")) as fh:
        meta = json.load(fh)
    g = load_golden("depth_map")
    exact = 0
    for name, m in meta.items():
        frames, want = list(g[name + "_frames"]), g[name + "_out"]
        got = run_gpu(L, frames, **m["kwargs"])
        und = g.get(name + "_undefined")
        ok, info = close_enough(got, want, None if und is None else ~und)
        assert ok, (name, info)
        exact += info[0] == 0
    assert exact >= len(meta) - 1   # bit-identical is the rule, the tolerance the exception


def scene(rng, n, h, w, dtype):
    top = 255 if dtype == np.uint8 else 65535
    base = rng.random((h, w, 3))
    out = []
    for i in range(n):
        k = 1 + 2 * abs(i - n // 2)
        sm = base.copy()
        for _ in range(k - 1):
            sm = (sm + np.roll(sm, 1, 0) + np.roll(sm, 1, 1)) / 3
        mix = np.linspace(0, 1, w)[None, :, None] if i % 2 else np.linspace(1, 0, h)[:, None, None]
        out.append(np.clip((mix * base + (1 - mix) * sm) * top, 0, top).astype(dtype))
    return out


CASES = [
    (np.uint8, 4, 61, 83, {}),
    (np.uint16, 3, 50, 77, {}),
    (np.uint8, 3, 64, 64, {"map_type": "max"}),
    (np.uint16, 3, 47, 35, {"map_type": "max", "temperature": 0.02, "levels": 2}),
    (np.uint8, 5, 39, 101, {"energy": "sobel", "levels": 4}),
    (np.uint16, 2, 33, 33, {"energy": "sobel", "smooth_size": 0, "levels": 1}),
    (np.uint8, 3, 40, 56, {"kernel_size": 1, "blur_size": 1, "smooth_size": 3}),
    (np.uint8, 3, 40, 56, {"kernel_size": 3, "blur_size": 7, "smooth_size": 31, "levels": 5}),
    (np.uint8, 2, 70, 45, {"kernel_size": 9, "blur_size": 11, "smooth_size": 7}),
    (np.uint8, 2, 5, 7, {"levels": 3}),            # tiny: every stencil reflects more than once
    (np.uint8, 2, 1, 9, {"levels": 2, "smooth_size": 5}),
    (np.uint16, 2, 130, 259, {"levels": 6}),
    (np.uint8, 2, 141, 270, {"levels": 4}),        # 8-bit tiles of the one-pass energy kernel's interior path (and its rim)
    (np.uint8, 2, 141, 270, {"map_type": "max", "smooth_size": 0, "levels": 2}),
    (np.uint8, 2, 134, 530, {"levels": 3, "smooth_size": 5}),   # level 1 wide enough for dm_pyrdown_tile's interior staging (3-channel float)
    (np.uint8, 4, 61, 83, {"float_type": "float-64"}),
    (np.uint16, 3, 50, 77, {"float_type": "float-64", "map_type": "max"}),
    (np.uint16, 3, 47, 66, {"float_type": "float-64", "smooth_size": 0, "levels": 4}),
    (np.uint8, 3, 39, 58, {"float_type": "float-64", "smooth_size": 0, "map_type": "max", "energy": "sobel"}),
    (np.uint8, 2, 40, 40, {"float_type": "float-64", "kernel_size": 9, "blur_size": 11, "smooth_size": 7, "levels": 2}),
    (np.uint8, 2, 3, 5, {"float_type": "float-64"}),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_random_stacks_vs_oracle(L, case):
    dtype, n, h, w, kw = CASES[case]
    frames = scene(np.random.default_rng(100 + case), n, h, w, dtype)
    want = dmo.depth_map_stack(frames, **kw)
    got = run_gpu(L, frames, **kw)
    ok, info = close_enough(got, want)
    assert ok, (CASES[case], info)


def test_handle_reuse_device_push_and_order(L):
    rng = np.random.default_rng(7)
    a, b = scene(rng, 3, 48, 64, np.uint8), scene(rng, 5, 48, 64, np.uint8)
    with L.DepthMap(48, 64) as dm:
        outs = []
        for frames in (a, b, a):
            for f in frames:
                dm.push_frame(f)
            outs.append(dm.finish())
            with pytest.raises(RuntimeError):
                dm.push_frame(frames[0])          # push after finish
            dm.reset()
        assert np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[0], outs[1])
        assert close_enough(outs[1], dmo.depth_map_stack(b))[0]
        # frames already resident in device memory, result to device memory
        buf = L.DeviceBuffer(a[0].nbytes * (len(a) + 1))
        for i, f in enumerate(a):
            buf.upload(f, i * f.nbytes)
            dm.push_frame_device(buf.ptr + i * f.nbytes)
        dm.finish_device(buf.ptr + len(a) * a[0].nbytes)
        assert np.array_equal(buf.download(a[0].shape, np.uint8, len(a) * a[0].nbytes), outs[0])
        buf.free()
        dm.reset()
        with pytest.raises(RuntimeError):
            dm.finish()                           # nothing pushed
    # frame order matters only through float summation: a permutation stays within the tolerance
    assert close_enough(run_gpu(L, a[::-1]), outs[0])[0]


def test_argument_errors(L):
    from shinestacker_amd import InvalidOptionError
    for kw in ({"kernel_size": 4}, {"kernel_size": 17}, {"blur_size": 6}, {"smooth_size": 33}, {"levels": 0},
               {"map_type": 2}, {"energy": 5}, {"map_type": 1, "temperature": 0.0}, {"dtype": np.float32}):
        with pytest.raises((ValueError, RuntimeError, KeyError, InvalidOptionError)):
            L.DepthMap(32, 32, **kw)
    with L.DepthMap(32, 32) as dm:
        with pytest.raises(ValueError):
            dm.push_frame(np.zeros((32, 33, 3), np.uint8))
        with pytest.raises(ValueError):
            dm.push_frame(np.zeros((32, 32, 3), np.uint16))


class Proc:
    id, name = 3, "dm"

    def __init__(self, stop_at=None):
        self.trace, self.stop_at = [], stop_at

    def callback(self, key, *a):
        self.trace.append((key,) + a)
        if key == "check_running" and self.stop_at is not None:
            return sum(t[0] == "check_running" for t in self.trace) <= self.stop_at
        return True

    def sub_message_r(self, *_a, **_k):
        pass


def test_plugin_protocol_on_files(L, tmp_path):
    from shinestacker_amd import DepthMapStack, InvalidOptionError, RunStopException
    from shinestacker_amd.errors import ShapeError
    from shinestacker_amd.imageio import write_img
    frames = scene(np.random.default_rng(11), 6, 72, 96, np.uint8)
    names = []
    for i, f in enumerate(frames):
        names.append(str(tmp_path / f"f{i:02d}.png"))
        write_img(names[-1], f)
    want = dmo.depth_map_stack(frames)
    for threads in (1, 4):
        algo = DepthMapStack(decode_threads=threads)
        algo.process = Proc()
        out = algo.focus_stack(names)
        assert close_enough(out, want)[0]
        tr = algo.process.trace
        assert [t[0] for t in tr] == ["after_step", "check_running"] * 12
        assert [t[3] for t in tr if t[0] == "after_step"] == list(range(12))
    # the same stacker object again (FocusStackBunch reuses it), other options
    algo.map_type, algo.energy = "max", "sobel"
    assert close_enough(algo.focus_stack(names[:4]), dmo.depth_map_stack(frames[:4], map_type="max", energy="sobel"))[0]
    algo64 = DepthMapStack(float_type="float-64", levels=2)
    algo64.process = Proc()
    assert close_enough(algo64.focus_stack(names[:3]), dmo.depth_map_stack(frames[:3], float_type="float-64", levels=2))[0]
    # stop request during the first loop
    algo = DepthMapStack()
    algo.process = Proc(stop_at=2)
    with pytest.raises(RunStopException):
        algo.focus_stack(names)
    assert sum(t[0] == "after_step" for t in algo.process.trace) == 3
    # unknown options surface after the first loop, as in the reference (depth_map.py:83-87, :62-63)
    for kw in ({"energy": "variance"}, {"map_type": "median"}):
        algo = DepthMapStack(**kw)
        algo.process = Proc()
        with pytest.raises(InvalidOptionError):
            algo.focus_stack(names)
        assert sum(t[0] == "after_step" for t in algo.process.trace) == 6
    write_img(names[3], frames[3][:, :80])
    algo = DepthMapStack()
    algo.process = Proc()
    with pytest.raises(ShapeError):
        algo.focus_stack(names)


def test_focus_stack_action_with_depth_map(L, tmp_path):
    """tests/test_0060_stack.py:27-34 of the reference: StackJob + FocusStack('...', DepthMapStack())."""
    from shinestacker_amd import DepthMapStack, FocusStack, StackJob
    from shinestacker_amd.imageio import read_img, write_img
    frames = scene(np.random.default_rng(12), 4, 64, 80, np.uint16)
    work = tmp_path / "proj"
    os.makedirs(work / "input")
    for i, f in enumerate(frames):
        write_img(str(work / "input" / f"im{i}.tif"), f)
    job = StackJob("job", str(work), input_path="input")
    job.add_action(FocusStack("stack-depthmap", DepthMapStack(), output_path="out", prefix="dm_"))
    job.run()
    outs = sorted(os.listdir(work / "out"))
    assert len(outs) == 1 and outs[0].startswith("dm_")
    got = read_img(str(work / "out" / outs[0]))
    assert close_enough(got, dmo.depth_map_stack(frames))[0]


def test_full_size_properties(L):
    """24 MP frames (BASELINE config size), properties that need no CPU oracle: identical frames fuse to the frame
    (weights 1/N each), and of a sharp and a defocused version of one scene the result follows the sharp one."""
    H, W = 4000, 6000
    rng = np.random.default_rng(13)
    tile = rng.integers(0, 256, (250, 375, 3)).astype(np.uint8)
    sharp = np.tile(tile, (16, 16, 1))
    out = run_gpu(L, [sharp, sharp, sharp, sharp])
    assert np.abs(out.astype(np.int16) - sharp.astype(np.int16)).max() <= 1
    soft = ((sharp.astype(np.uint16) + np.roll(sharp, 1, 0) + np.roll(sharp, 1, 1) + np.roll(sharp, (1, 1), (0, 1))) // 4).astype(np.uint8)
    left_sharp = soft.copy()
    left_sharp[:, : W // 2] = sharp[:, : W // 2]
    right_sharp = soft.copy()
    right_sharp[:, W // 2:] = sharp[:, W // 2:]
    out = run_gpu(L, [left_sharp, right_sharp], map_type="max", temperature=0.01)
    inner = np.s_[64:-64, 64:-64]
    err_sharp = np.abs(out.astype(np.int16) - sharp.astype(np.int16))[inner].mean()
    err_soft = np.abs(out.astype(np.int16) - soft.astype(np.int16))[inner].mean()
    assert err_sharp < 0.35 * err_soft, (err_sharp, err_soft)


def test_step_methods_equal_the_reference_methods_of_the_same_names(L):
    """DepthMapStack.get_sobel_map / get_laplacian_map / smooth_energy / get_focus_map (depth_map.py:28-62: public methods
    the reference's own tests call) against recordings of the REFERENCE's methods run over the cv2 shim
    (oracle/gen_golden.py::depth_map_steps_case): float-32 and float-64, both map types, several kernel sizes.  Bit-equal
    for the float-32 planes; float-64: the device library's exp / the fma contraction of a 25-tap float-64 chain may differ
    in the last place (the tolerance the fused path's tests use)."""
    import json
    from shinestacker_amd import DepthMapStack
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "depth_map_steps.npz"))
    for e in json.loads(str(z["meta"])):
        t = e["tag"]
        dms = DepthMapStack(float_type=e["float_type"], **e["kwargs"])
        f64 = e["float_type"] == "float-64"

        def same(got, want, what):
            assert got.dtype == want.dtype and got.shape == want.shape, (t, what, got.dtype, want.dtype)
            if f64 and want.dtype == np.float64:
                assert np.allclose(got, want, rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(want).max()))), (t, what)
            else:
                assert np.array_equal(got, want), (t, what, float(np.abs(got.astype(np.float64) - want).max()))
        same(dms.get_sobel_map(z[f"{t}_gray"]), z[f"{t}_sobel"], "sobel")
        same(dms.get_laplacian_map(z[f"{t}_gray"]), z[f"{t}_laplacian"], "laplacian")
        sm = dms.smooth_energy(z[f"{t}_energy"])
        same(sm, z[f"{t}_smoothed"], "smoothed")
        fm = dms.get_focus_map(z[f"{t}_smoothed"])
        want = z[f"{t}_focus"]
        if e["kwargs"].get("map_type", "average") == "average":
            ok = z[f"{t}_smoothed"].sum(axis=0) != 0          # the reference leaves zero-total pixels uninitialised
            assert fm.dtype == want.dtype and np.array_equal(fm[:, ok], want[:, ok]) and np.all(fm[:, ~ok] == 0), t
        else:
            same(fm, want, "focus")
        dms.close()
