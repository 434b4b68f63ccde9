"""GPU: MI_ARITH_SEPARABLE (csrc/kernels_sep.hpp) through the C ABI against its CPU restatement
(oracle/separable_oracle.c), bit for bit: every tap of both implementations (LDS-tiled and one-thread-per-output),
interior and border tiles, odd sizes at every level, ties, u8 / u16 / f32 input, host and device pushes, batch
boundaries.  The tolerance of this arithmetic against float64 is tests/test_sep_tolerance.py (CPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


def run_oracle(oracle, frames, **kw):
    h, w = frames[0].shape[:2]
    dt = frames[0].dtype if frames[0].dtype != np.float32 else np.uint8
    so = oracle.StreamingOracle(h, w, dt, arith="separable", **kw)
    gs = [so.push_frame(f) for f in frames]
    return so, gs


def compare(L, st, so, last_gauss=None):
    for lv in range(st.levels):
        assert np.array_equal(st.tap(L.TAP_ENERGY, lv), so.best_e[lv]), f"energy {lv}"
        assert np.array_equal(st.tap(L.TAP_INDEX, lv), so.best_idx[lv]), f"index {lv}"
        assert np.array_equal(st.tap(L.TAP_FUSED_LAP, lv), so.best_lap[lv]), f"lap {lv}"
    if last_gauss is not None:
        for lv in range(1, st.levels + 1):
            assert np.array_equal(st.tap(L.TAP_GAUSS, lv), last_gauss[lv]), f"gauss {lv}"
    want = so.finish()
    got = st.finish()
    assert np.array_equal(st.tap(L.TAP_COLLAPSED), np.clip(np.abs(so.collapse()), 0, 255 if got.dtype == np.uint8 else 65535))
    assert got.dtype == want.dtype and np.array_equal(got, want)


CASES = [
    # (h, w, n, dtype, min_size, gen_kernel, batch)
    (133, 201, 4, np.uint8, 8, 0.4, 0),       # odd sizes at every level, all tiles are border tiles
    (300, 452, 5, np.uint8, 32, 0.4, 2),      # interior + border tiles, batches of 2 (state reloaded between launches)
    (257, 130, 3, np.uint16, 16, 0.35, 0),
    (96, 64, 3, np.float32, 8, 0.4, 0),
    (500, 750, 6, np.float32, 32, 0.5, 4),
    (97, 1031, 3, np.uint8, 8, 0.3, 0),       # one tile row, many tile columns
    (640, 90, 3, np.uint16, 8, 0.4, 0),       # many tile rows, two tile columns
    # edge tiles (staged through the mirror, interior code): even sizes with whole and partial tiles at the far edges,
    # a far edge exactly on a tile boundary, images narrower than one tile + halo, the smallest level that folds (16)
    (112, 224, 4, np.uint8, 8, 0.4, 0),
    (284, 458, 4, np.float32, 16, 0.4, 3),
    (30, 58, 3, np.uint16, 8, 0.4, 0),
    (256, 64, 5, np.uint8, 8, 0.35, 0),
    (230, 342, 4, np.uint8, 8, 0.4, 0),       # even / odd alternate down the levels: edge and border tiles side by side
    # reduce taps (red_taps): a negative integer outer tap (a = 0.7: -2 5 14), integer taps over the exactness bound
    # (a = 6.7: -62 5 134 -> the float taps), both on 16-bit input where the bound matters
    (150, 226, 3, np.uint16, 8, 0.7, 0),
    (150, 226, 3, np.uint16, 8, 6.7, 0),
    (133, 201, 3, np.uint8, 8, 0.7, 0),
]


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("h,w,n,dt,min_size,a,batch", CASES)
def test_separable_equals_oracle(L, oracle, impl, h, w, n, dt, min_size, a, batch):
    rng = np.random.default_rng(h * 7 + w)
    hi = 65536 if dt == np.uint16 else 256
    frames = [rng.integers(0, hi, (h, w, 3)).astype(dt) for _ in range(n)]
    if n > 2:
        frames[2] = frames[0].copy()     # exact ties: the first maximum must win
    so, gs = run_oracle(oracle, frames, min_size=min_size, gen_kernel=a)
    st = L.Stack(h, w, in_dtype=dt, out_dtype=np.uint16 if dt == np.uint16 else np.uint8, impl=impl, arith="separable",
                 min_size=min_size, gen_kernel=a, batch_frames=batch)
    assert st.levels == so.levels
    for f in frames:
        st.push_frame(f)
    compare(L, st, so, gs[-1])
    st.close()


def test_separable_device_push_many_frames(L, oracle):
    """34 frames resident in HBM: one full batch of 32 plus a tail, tiled kernel, synthetic generator."""
    h, w, n = 420, 620, 34
    frames = [oracle.synth_frame_numpy(h, w, f, n) for f in range(n)]
    so, _ = run_oracle(oracle, frames)
    buf = L.DeviceBuffer(h * w * 3 * n)
    for i, f in enumerate(frames):
        buf.upload(f, i * h * w * 3)
    st = L.Stack(h, w, in_dtype=np.uint8, arith="separable")
    st.push_frames_device(buf.ptr, n)
    compare(L, st, so)
    # the handle is reusable: a second, shorter stack after reset
    st.reset()
    st.push_frames_device(buf.ptr, 3)
    so2, _ = run_oracle(oracle, frames[:3])
    compare(L, st, so2)
    st.close()
    buf.free()


@pytest.mark.parametrize("dt", [np.uint8, np.float32])
def test_separable_frame_chunks(L, oracle, dt):
    """Long resident pushes of small frames: every level runs in frame chunks (blockIdx.y) whose partial maxima are merged
    in chunk order.  Duplicate frames sit in different chunks and in different pushes -- the earliest must stay the
    winner -- and the second push continues the state the first one left."""
    h, w, n = 133, 201, 70
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(n)]
    for dup, of in ((17, 3), (29, 3), (44, 30), (58, 3), (69, 44), (52, 51)):
        frames[dup] = frames[of].copy()
    so, _ = run_oracle(oracle, frames, min_size=8)
    fr = [f.astype(dt) for f in frames]
    fb = h * w * 3 * np.dtype(dt).itemsize
    buf = L.DeviceBuffer(fb * n)
    for i, f in enumerate(fr):
        buf.upload(f, i * fb)
    st = L.Stack(h, w, in_dtype=dt, out_dtype=np.uint8, arith="separable", min_size=8)
    st.push_frames_device(buf.ptr, 50)
    st.push_frames_device(buf.ptr + 50 * fb, 20)
    compare(L, st, so)
    st.close()
    # the same with a named batch size: 70 frames > 48 -> two equal batches (36 + 34 frames), chunks of 16
    st = L.Stack(h, w, in_dtype=dt, out_dtype=np.uint8, arith="separable", min_size=8, batch_frames=48)
    st.push_frames_device(buf.ptr, n)
    compare(L, st, so)
    st.close()
    buf.free()


def test_separable_close_to_exact_mode(L, oracle):
    """Same stack in both arithmetic modes: fused images differ by at most 1 count, on few pixels."""
    h, w, n = 300, 452, 5
    frames = [oracle.synth_frame_numpy(h, w, f, n) for f in range(n)]
    outs = []
    for arith in ("exact", "separable"):
        st = L.Stack(h, w, in_dtype=np.uint8, arith=arith)
        for f in frames:
            st.push_frame(f)
        outs.append(st.finish().astype(np.int32))
        st.close()
    d = np.abs(outs[0] - outs[1])
    assert d.max() <= 1 and (d != 0).mean() < 1e-3


def test_separable_mixed_pushes_grow_the_batch_buffers(L, oracle):
    """Host frames (32-frame ring), then resident pushes of growing length on the same handle: the per-batch buffers and
    the chunk partials are re-allocated on demand, the running state carries over, reset() starts a new stack."""
    h, w = 210, 340
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(3 + 40 + 7 + 66)]
    frames[50] = frames[2].copy()
    frames[100] = frames[2].copy()
    fb = h * w * 3
    buf = L.DeviceBuffer(fb * len(frames))
    for i, f in enumerate(frames):
        buf.upload(f, i * fb)
    st = L.Stack(h, w, in_dtype=np.uint8, arith="separable", min_size=16)
    for f in frames[:3]:
        st.push_frame(f)
    st.push_frames_device(buf.ptr + 3 * fb, 40)
    st.push_frames_device(buf.ptr + 43 * fb, 7)
    st.push_frames_device(buf.ptr + 50 * fb, 66)
    so, _ = run_oracle(oracle, frames, min_size=16)
    compare(L, st, so)
    st.reset()
    st.push_frames_device(buf.ptr + 20 * fb, 35)
    so, _ = run_oracle(oracle, frames[20:55], min_size=16)
    compare(L, st, so)
    st.close()
    buf.free()


@pytest.mark.parametrize("seed", range(12))
def test_separable_tiled_equals_simple_on_random_shapes(L, seed):
    """Randomised cross-check of the two implementations of the separable arithmetic (LDS-tiled with frame chunks /
    consecutive launches / merges vs one thread per output, one frame at a time): random sizes (odd, narrow, a single
    tile row or column), frame counts that straddle the chunking rules, input types, kernel parameters; resident push
    in one or several pieces.  Every level's energy / arg-max / Laplacian and the fused image must be identical."""
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(33, 420)), int(rng.integers(33, 640))
    n = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 47, 64, 65]))
    dt = [np.uint8, np.uint16, np.float32][seed % 3]
    hi = 65536 if dt == np.uint16 else 256
    frames = [rng.integers(0, hi, (h, w, 3)).astype(dt) for _ in range(min(n, 9))]
    frames = [frames[int(k)] for k in rng.integers(0, len(frames), n)]           # many exact duplicates: ties across chunks
    kw = dict(in_dtype=dt, out_dtype=np.uint16 if dt == np.uint16 else np.uint8, arith="separable",
              min_size=int(rng.choice([8, 16, 32])), gen_kernel=float(rng.choice([0.3, 0.4, 0.5])))
    a = L.Stack(h, w, impl=1, **kw)
    for f in frames:
        a.push_frame(f)
    fb = frames[0].nbytes
    buf = L.DeviceBuffer(fb * n)
    for i, f in enumerate(frames):
        buf.upload(f, i * fb)
    b = L.Stack(h, w, impl=2, **kw)
    cut = int(rng.integers(0, n + 1))
    if cut:
        b.push_frames_device(buf.ptr, cut, fb)
    if n - cut:
        b.push_frames_device(buf.ptr + cut * fb, n - cut, fb)
    assert a.levels == b.levels
    for lv in range(a.levels):
        for tap in (L.TAP_ENERGY, L.TAP_INDEX, L.TAP_FUSED_LAP):
            assert np.array_equal(a.tap(tap, lv), b.tap(tap, lv)), (seed, h, w, n, dt, lv, tap)
    assert np.array_equal(a.finish(), b.finish()), (seed, h, w, n, dt)
    a.close()
    b.close()
    buf.free()


def test_pyramid_stack_arith_option(L, oracle):
    """PyramidStack(arith="separable") -- the keyword the drop-in class adds -- fuses with the separable arithmetic
    (== its oracle) and is what PyramidStack() gives (constants.DEFAULT_PY_ARITH, round 4); arith="exact" is the reference's
    evaluation order (== the exact oracle); the environment cannot change it; float-64 stacks stay exact; bad
    combinations are refused."""
    from shinestacker_amd.errors import InvalidOptionError
    from shinestacker_amd.pyramid import PyramidStack
    h, w, n = 300, 452, 5
    frames = [oracle.synth_frame_numpy(h, w, f, n) for f in range(n)]
    so, _ = run_oracle(oracle, frames)
    assert np.array_equal(PyramidStack(arith="separable").focus_stack_arrays(frames), so.finish())
    se = oracle.StreamingOracle(h, w, np.uint8)
    for f in frames:
        se.push_frame(f)
    want_exact = se.finish()
    assert np.array_equal(PyramidStack(arith="exact").focus_stack_arrays(frames), want_exact)
    assert PyramidStack().arith == "separable" and PyramidStack(float_type="float-64").arith == "exact"
    assert np.array_equal(PyramidStack().focus_stack_arrays(frames), so.finish())
    # one default behind every high-level entry point, and no environment override (round 5)
    from shinestacker_amd.defaults import resolve_arith
    import os
    os.environ["SHINESTACKER_AMD_ARITH"] = "exact"
    try:
        assert PyramidStack().arith == "separable" == resolve_arith() and resolve_arith(None, "float-64") == "exact"
    finally:
        del os.environ["SHINESTACKER_AMD_ARITH"]
    assert "arith=separable" in PyramidStack().describe()
    with pytest.raises(InvalidOptionError):
        PyramidStack(arith="fast")
    with pytest.raises(InvalidOptionError):
        PyramidStack(arith="separable", float_type="float-64")


def test_separable_rejects_float64(L):
    with pytest.raises(ValueError):
        L.Stack(64, 64, arith="separable", float_type=L.MI_F64)
