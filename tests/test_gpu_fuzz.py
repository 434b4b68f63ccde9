"""GPU: randomised differential test of the tiled path against the streaming oracle -- shapes that
exercise every code path of the fused kernel (odd widths, images narrower than a tile, widths not
divisible by 4 at some level, single frames, batch boundaries, host vs device frames, all dtypes,
both arithmetic modes, several pyramid depths)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


def make_frames(rng, shape, dtype, n):
    hi = 256 if dtype == np.uint8 else 65536
    out = []
    for f in range(n):
        coarse = rng.integers(0, hi, (shape[0] // 6 + 2, shape[1] // 6 + 2, 3))
        img = np.kron(coarse, np.ones((6, 6, 1)))[:shape[0], :shape[1]]
        img = img + rng.integers(-hi // 8, hi // 8 + 1, shape + (3,)) * ((f % 3) + 1) // 3
        out.append(np.clip(img, 0, hi - 1).astype(dtype))
    return out


def check(L, oracle, frames, in_dtype=None, batch=0, device_frames=False, impl=None, **kw):
    h, w = frames[0].shape[:2]
    dt = frames[0].dtype
    so = oracle.StreamingOracle(h, w, dt, keep_gauss=False, **kw)
    for f in frames:
        so.push_frame(f)
    want = so.finish()
    src_dt = in_dtype or dt
    st = L.Stack(h, w, in_dtype=src_dt, out_dtype=dt, impl=L.IMPL_TILED if impl is None else impl, batch_frames=batch, **kw)
    assert st.levels == so.levels
    if device_frames:
        per = h * w * 3 * np.dtype(src_dt).itemsize
        buf = L.DeviceBuffer(per * len(frames))
        for i, f in enumerate(frames):
            buf.upload(f.astype(src_dt), i * per)
        st.push_frames_device(buf.ptr, len(frames))
    else:
        for f in frames:
            st.push_frame(f.astype(src_dt))
    for lv in range(st.levels):
        assert np.array_equal(st.tap(L.TAP_INDEX, lv), so.best_idx[lv]), f"index level {lv}"
        assert np.array_equal(st.tap(L.TAP_ENERGY, lv), so.best_e[lv]), f"energy level {lv}"
        assert np.array_equal(st.tap(L.TAP_FUSED_LAP, lv), so.best_lap[lv]), f"lap level {lv}"
    got = st.finish()
    assert np.array_equal(st.tap(L.TAP_FUSED_BASE), so.fused_base())
    assert np.array_equal(got, want)
    st.close()


CASES = [
    # (h, w, dtype, n, kwargs, batch, device_frames, in_dtype)
    (64, 64, np.uint8, 1, {}, 0, False, None),                       # one level, single frame
    (65, 67, np.uint8, 2, {}, 0, False, None),                       # odd sizes, one level
    (131, 259, np.uint8, 3, {}, 0, False, None),                     # odd width at level 0 (scalar staging)
    (200, 999, np.uint16, 2, {}, 0, False, None),                    # wn % 4 != 0 at several levels
    (40, 700, np.uint8, 3, {"min_size": 8}, 0, False, None),         # shorter than a tile, wide
    (700, 40, np.uint8, 3, {"min_size": 8}, 0, False, None),         # narrower than a tile
    (257, 385, np.uint8, 35, {"min_size": 64}, 0, True, None),       # > one batch, tapered tail, device frames
    (192, 320, np.uint8, 9, {}, 4, False, None),                     # host ring wraps (batch 4)
    (192, 320, np.uint16, 7, {}, 3, True, None),                     # device frames, odd batch
    (300, 300, np.uint8, 4, {"use_fma": False}, 0, False, None),
    (300, 300, np.uint8, 4, {}, 0, True, np.float32),                # f32 input holding integers
    (513, 771, np.uint8, 3, {"min_size": 16, "gen_kernel": 0.35, "kernel_size": 3}, 0, False, None),
    (96, 128, np.uint16, 5, {"min_size": 4}, 2, False, None),        # deepest pyramid: 4 levels, tiny base
    (1000, 70, np.uint8, 2, {"min_size": 16}, 0, True, None),
    (40, 52, np.uint8, 4, {}, 0, False, None),                       # smaller than 2*min_size: base only, no levels
    (33, 47, np.uint16, 3, {}, 2, True, None),                       # same, 16-bit, device frames
    # a generating kernel with NEGATIVE outer taps (a > 0.5): the blurred energies can be negative, the first frame must win all
    # the same (round 6: the running maximum started at -1, "every energy is >= 0")
    (150, 226, np.uint8, 4, {"min_size": 8, "gen_kernel": 0.7}, 0, False, None),
    (150, 226, np.uint16, 3, {"min_size": 8, "gen_kernel": 0.7}, 0, True, None),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_fuzz_case(L, oracle, case):
    h, w, dt, n, kw, batch, dev, in_dt = CASES[case]
    rng = np.random.default_rng(1000 + case)
    frames = make_frames(rng, (h, w), dt, n)
    check(L, oracle, frames, in_dtype=in_dt, batch=batch, device_frames=dev, **kw)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_random_shapes(L, oracle, seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(64, 420)), int(rng.integers(64, 520))
    dt = [np.uint8, np.uint16][int(rng.integers(0, 2))]
    n = int(rng.integers(1, 7))
    kw = {"min_size": int(rng.choice([8, 16, 32])), "use_fma": bool(rng.integers(0, 2))}
    frames = make_frames(rng, (h, w), dt, n)
    check(L, oracle, frames, batch=int(rng.integers(0, 4)), device_frames=bool(rng.integers(0, 2)), **kw)


def test_u16_full_range_values(L, oracle):
    frames = [np.full((96, 96, 3), 65535, np.uint16), np.zeros((96, 96, 3), np.uint16)]
    frames[1][::2, ::2] = 65535
    check(L, oracle, frames)


@pytest.mark.parametrize("h,w,off,pad", [(211, 333, 4, 12), (256, 512, 8, 0), (130, 262, 12, 36), (300, 1000, 0, 4000)])
@pytest.mark.parametrize("src_dt", [np.float32, np.uint8, np.uint16])
def test_device_frames_with_odd_alignment_and_stride(L, oracle, h, w, off, pad, src_dt):
    """Frames resident on the device at a base address that is only 4-byte aligned and with a frame
    stride larger than a frame: the vector staging paths (16-byte loads, dword loads of packed 8/16-bit
    rows) must not assume more alignment than they check."""
    rng = np.random.default_rng(h + w + off)
    dt = np.uint8 if src_dt == np.float32 else src_dt
    frames = make_frames(rng, (h, w), dt, 5)
    so = oracle.StreamingOracle(h, w, dt, keep_gauss=False)
    for f in frames:
        so.push_frame(f)
    want = so.finish()
    per = h * w * 3 * np.dtype(src_dt).itemsize
    stride = per + pad
    buf = L.DeviceBuffer(off + stride * len(frames) + 64)
    for i, f in enumerate(frames):
        buf.upload(f.astype(src_dt), off + i * stride)
    for impl in (L.IMPL_TILED, L.IMPL_SIMPLE):
        st = L.Stack(h, w, in_dtype=src_dt, out_dtype=dt, impl=impl, batch_frames=3)
        st.push_frames_device(buf.ptr + off, len(frames), stride)
        assert np.array_equal(st.finish(), want), (impl, off, pad)
        st.close()
