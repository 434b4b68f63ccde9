"""GPU: MI_ARITH_SEPARABLE with pyramid levels run as PAIRS (csrc/kernels_sep.hpp: the first level's kernel -- level_sep_pair --
hands the second gray(G_{l+1}) and G_{l+2}, level_sep_e is the second level's energy pass, level_sep_pl / sep_payload_pair0 / 1
recompute the winners' G_{l+1}; the three-channel G_{l+1} of the batch never reaches HBM -- reference: the reduce -> expand
dependency of algorithms/pyramid.py:27-46, :125-139).  A pair must give the SAME BITS as the level-by-level kernels, i.e. equal
oracle/separable_oracle.c on every tap: the cases of tests/test_gpu_separable.py with the pair plans forced (pair_levels = 1:
(0, 1), (2, 3), ...; 3: (1, 2), (3, 4), ... -- small test stacks stay below the automatic plan's threshold), frame chunks, batch
boundaries, the automatic choice, the kept-frame tap, and the tile-by-tile payload pass."""
import numpy as np
import pytest

from test_gpu_separable import CASES, compare, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


@pytest.mark.parametrize("pl", [1, 3])       # pairs (0, 1), (2, 3), ... / (1, 2), (3, 4), ...
@pytest.mark.parametrize("h,w,n,dt,min_size,a,batch", CASES)
def test_pair_equals_oracle(L, oracle, h, w, n, dt, min_size, a, batch, pl):
    rng = np.random.default_rng(h * 7 + w)
    hi = 65536 if dt == np.uint16 else 256
    frames = [rng.integers(0, hi, (h, w, 3)).astype(dt) for _ in range(n)]
    if n > 2:
        frames[2] = frames[0].copy()     # exact ties: the first maximum must win
    so, gs = run_oracle(oracle, frames, min_size=min_size, gen_kernel=a)
    st = L.Stack(h, w, in_dtype=dt, out_dtype=np.uint16 if dt == np.uint16 else np.uint8, impl=2, arith="separable",
                 min_size=min_size, gen_kernel=a, batch_frames=batch, pair_levels=pl)
    for f in frames:
        st.push_frame(f)
    compare(L, st, so, gs[-1])      # (gs[-1][l + 1]: the three-channel G_{l+1} of the last frame -- a pair keeps exactly that one)
    st.close()


@pytest.mark.parametrize("pl", [1, 3])
@pytest.mark.parametrize("dt", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("h,w", [(420, 620), (421, 619), (338, 458), (1000, 1500)])
def test_pair_sizes_and_types(L, oracle, dt, h, w, pl):
    """natural-ish content (the generator), sizes whose levels 0 / 1 / 2 are even / odd in every combination, a size with many
    interior tiles; host pushes in batches of 3 (state reloaded, the kept frame changes) and one resident push"""
    n = 7
    hi = 257 if dt == np.uint16 else 1
    frames = [(oracle.synth_frame_numpy(h, w, f, n).astype(np.uint16) * hi).astype(dt) for f in range(n)]
    so, gs = run_oracle(oracle, frames, min_size=16)     # (one level more than the default: pairs at two depths)
    od = np.uint16 if dt == np.uint16 else np.uint8
    st = L.Stack(h, w, in_dtype=dt, out_dtype=od, arith="separable", batch_frames=3, pair_levels=pl, min_size=16)
    for f in frames:
        st.push_frame(f)
    compare(L, st, so, gs[-1])
    st.close()
    fb = h * w * 3 * np.dtype(dt).itemsize
    buf = L.DeviceBuffer(fb * n)
    for i, f in enumerate(frames):
        buf.upload(f, i * fb)
    so, gs = run_oracle(oracle, frames, min_size=16)
    st = L.Stack(h, w, in_dtype=dt, out_dtype=od, arith="separable", pair_levels=pl, min_size=16)
    st.push_frames_device(buf.ptr, n)
    compare(L, st, so, gs[-1])
    st.close()
    buf.free()


@pytest.mark.parametrize("dt", [np.uint8, np.float32])
def test_pair_frame_chunks(L, oracle, dt):
    """small frames, long resident pushes: both levels of the pair run in frame chunks whose partial maxima the pair's payload
    passes fold; duplicates in different chunks and pushes; the automatic choice (50 and 20 frames: pair, then not)"""
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
    for pl in (1, 0):
        so, gs = run_oracle(oracle, frames, min_size=8)
        st = L.Stack(h, w, in_dtype=dt, out_dtype=np.uint8, arith="separable", min_size=8, pair_levels=pl)
        st.push_frames_device(buf.ptr, 50)
        st.push_frames_device(buf.ptr + 50 * fb, 20)
        compare(L, st, so, gs[-1])
        st.close()
    so, _ = run_oracle(oracle, frames, min_size=8)
    st = L.Stack(h, w, in_dtype=dt, out_dtype=np.uint8, arith="separable", min_size=8, batch_frames=48, pair_levels=1)
    st.push_frames_device(buf.ptr, n)
    compare(L, st, so)
    st.close()
    buf.free()


def test_pair_is_bit_identical_to_the_unpaired_kernels(L, oracle):
    """the same resident stack with pair_levels 1 and 2: every tap of every level and the fused image are equal"""
    h, w, n = 676, 1012, 36
    frames = [oracle.synth_frame_numpy(h, w, f, n) for f in range(n)]
    fb = h * w * 3
    buf = L.DeviceBuffer(fb * n)
    for i, f in enumerate(frames):
        buf.upload(f, i * fb)
    taps = []
    for pl in (1, 2, 0, 3):
        st = L.Stack(h, w, in_dtype=np.uint8, arith="separable", pair_levels=pl)
        st.push_frames_device(buf.ptr, n)
        t = [st.tap(k, lv) for lv in range(st.levels) for k in (L.TAP_ENERGY, L.TAP_INDEX, L.TAP_FUSED_LAP)]
        t += [st.tap(L.TAP_GAUSS, lv) for lv in range(1, st.levels + 1)]
        t.append(st.finish())
        taps.append(t)
        st.close()
    buf.free()
    for other in taps[1:]:
        assert len(other) == len(taps[0])
        for x, y in zip(taps[0], other):
            assert np.array_equal(x, y)


def test_pair_needs_two_levels(L, oracle):
    """a one-level pyramid has no pair: pair_levels = 1 is ignored, results as ever"""
    h, w, n = 40, 70, 3
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(n)]
    so, gs = run_oracle(oracle, frames, min_size=16)
    st = L.Stack(h, w, in_dtype=np.uint8, arith="separable", min_size=16, pair_levels=1)
    assert st.levels == 1
    for f in frames:
        st.push_frame(f)
    compare(L, st, so, gs[-1])
    st.close()
    with pytest.raises(Exception):
        L.Stack(h, w, in_dtype=np.uint8, arith="separable", pair_levels=4)


@pytest.mark.parametrize("pl", [1, 3])
@pytest.mark.parametrize("kind", ["coherent", "noise", "mixed"])
def test_pair_tile_payload(L, oracle, kind, pl):
    """frames large enough for levels 0 AND 1 to run unchunked (> 1536 tiles each): the pair's payload is the tile-by-tile pass
    (level_sep_pl) -- `coherent`: few winners per tile, no tile is flagged; `noise`: 36 frames of noise, every tile has more
    than 32 winners and goes to the per-quad kernels; `mixed`: both kinds of tile in one image"""
    h, w, n = 2912, 3472, 36
    rng = np.random.default_rng(11)
    if kind == "noise":
        frames = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(n)]
    else:
        frames = [oracle.synth_frame_numpy(h, w, f, n) for f in range(n)]
        if kind == "mixed":
            for f in frames:
                f[:, : w // 2] = rng.integers(0, 256, (h, w // 2, 3)).astype(np.uint8)
    so, gs = run_oracle(oracle, frames)
    fb = h * w * 3
    buf = L.DeviceBuffer(fb * n)
    for i, f in enumerate(frames):
        buf.upload(f, i * fb)
    st = L.Stack(h, w, in_dtype=np.uint8, arith="separable", pair_levels=pl)
    st.push_frames_device(buf.ptr, n)
    compare(L, st, so, gs[-1])
    st.close()
    buf.free()
