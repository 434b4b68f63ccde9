"""GPU: the register-streaming interior kernel (MI_IMPL_STREAM, csrc/kernels_stream.hpp) against
the oracle and against the LDS-tiled kernel -- bit-exact state (index, energy, fused Laplacian)
on every level, fused base and output.  Sizes are chosen so that the streaming interior
([32, h-8) x [32, w-8) rounded to 32) has several 240-pixel strips, several row segments,
partial last strips/segments, and both the vector (w % 4 == 0) and the scalar load paths."""
import numpy as np
import pytest

from test_gpu_fuzz import check, make_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    import os
    hiplib.require_device()
    # production keeps small levels (too few waves to fill the chip) on the tiled kernel; the tests
    # push every level with a non-empty interior through the streaming kernel
    os.environ["MI_STREAM_MIN_WAVES"] = "1"
    yield hiplib
    del os.environ["MI_STREAM_MIN_WAVES"]


CASES = [
    # (h, w, dtype, n, kwargs, batch, in_dtype)
    (200, 360, np.uint8, 3, {}, 0, None),                   # one strip, two segments at level 0
    (330, 1000, np.uint8, 4, {}, 2, None),                  # four strips (last partial), batches of 2
    (264, 808, np.uint16, 3, {}, 0, None),                  # 16-bit
    (203, 413, np.uint8, 3, {}, 0, None),                   # odd sizes: scalar load path everywhere
    (250, 750, np.uint8, 5, {"use_fma": False}, 3, None),   # mul+add arithmetic, w % 4 == 2
    (256, 512, np.uint8, 4, {}, 4, np.float32),             # float32 frames of 8-bit values
    (520, 640, np.uint16, 2, {"min_size": 16}, 0, None),    # deeper pyramid: small levels through the same kernel
]


@pytest.mark.parametrize("h,w,dtype,n,kw,batch,in_dtype", CASES)
def test_stream_kernel_vs_oracle(L, oracle, h, w, dtype, n, kw, batch, in_dtype):
    frames = make_frames(np.random.default_rng(h * 7 + w), (h, w), dtype, n)
    check(L, oracle, frames, in_dtype=in_dtype, batch=batch, device_frames=True, impl=L.IMPL_STREAM, **kw)


def test_stream_equals_tiled_on_a_large_stack(L):
    """2000 x 3000 x 24 frames of the bench generator: every state array identical to the tiled path."""
    H, W, N = 2000, 3000, 24
    per = H * W * 3
    buf = L.DeviceBuffer(per * N)
    L.synth_frames_device(buf.ptr, np.uint8, H, W, 0, N, N)
    res = {}
    for impl in (L.IMPL_TILED, L.IMPL_STREAM):
        st = L.Stack(H, W, impl=impl, batch_frames=8)
        st.push_frames_device(buf.ptr, N)
        taps = [(st.tap(L.TAP_INDEX, lv), st.tap(L.TAP_ENERGY, lv), st.tap(L.TAP_FUSED_LAP, lv))
                for lv in range(st.levels)]
        res[impl] = (taps, st.finish())
        st.close()
    a, b = res[L.IMPL_TILED], res[L.IMPL_STREAM]
    for lv, (ta, tb) in enumerate(zip(a[0], b[0])):
        for name, x, y in zip(("index", "energy", "lap"), ta, tb):
            assert np.array_equal(x, y), f"{name} differs at level {lv}: {(x != y).sum()} of {x.size}"
    assert np.array_equal(a[1], b[1])
