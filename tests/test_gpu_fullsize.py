"""GPU: BASELINE.json configs 2, 4 and 5 at FULL size, self-verifying.  Sizes the oracle cannot
run whole are checked through size-independent properties plus the oracle on a corner: the level-0 selection state of
a pixel depends on a bounded neighbourhood of every frame, so a crop reproduces it exactly away from the crop's own
borders.  Reference paths: algorithms/pyramid.py:150-179 (config 2), stack.py:61-97 (config 5),
align.py:154-252 + tests/test_0031_align_precision.py:62-65 (config 4)."""
import ctypes as C
import os
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


_FUSED = {}   # arith -> fused image of config 2 (kept for interactive comparisons; the parity test below runs its own stacks)


@pytest.mark.parametrize("arith", ["exact", "separable"])
def test_config2_256_frames_24mp_fp32(L, oracle, arith):
    """256 x 4000x6000x3 fp32 frames resident in HBM (73.7 GB), 6 levels + 63x94 base: the benchmarked
    combination -- exact: 8 batches with double-buffered Gaussians and the tapered tail; separable: one batch, level
    after level, launches of 16 frames and frame chunks -- verified exactly as bench.py verifies its last step (band
    structure of the level-0 arg-max; level-0 energy / arg-max / fused Laplacian of the top-left corner == oracle on
    the cropped frames)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    H, W, N = 4000, 6000, 256
    per = H * W * 3 * 4
    buf = L.DeviceBuffer(per * N)
    L.synth_frames_device(buf.ptr, np.float32, H, W, 0, N, N)
    st = L.Stack(H, W, in_dtype=np.float32, out_dtype=np.uint8, arith=arith)
    assert st.levels == 6 and st.shapes[-1] == (63, 94)
    st.push_frames_device(buf.ptr, N)
    out = st.finish()
    args = types.SimpleNamespace(height=H, width=W, dtype="f32", arith=arith)
    v = bench.verify(L, st, args, N, 1)
    # six 512 x 512 windows (image corners, centre, a super-block seam), levels 0-2: energy / arg-max / fused Laplacian ==
    # the oracle fed the cropped frames (tests/test_verify_crops.py proves the windows on the CPU)
    assert v["band_match"] > 0.9 and v["crops_equal"] and len(v["crops"]) == 6, v
    assert all(sorted(c["levels"]) == ["0", "1", "2"] for c in v["crops"]), v
    # the fused image: every frame is sharp in its own band around the same 64..191 ramp, so the result stays in range
    assert out.shape == (H, W, 3) and out.dtype == np.uint8 and 40 < out.mean() < 215
    _FUSED[arith] = out
    st.close()
    buf.free()


def test_config2_parity_report_between_the_arithmetics_at_full_size(L):
    """SURVEY 8(d) "parity reporting" at the benchmark's own size (tools/parity_report.py): MI_ARITH_SEPARABLE against the
    reference-order arithmetic on the SAME 256 x 24 MP stack.  Per level the Gaussian and running-max-energy differences
    stay inside the stated forward-error bounds; the arg-max differs on a few dozen of 32 M pyramid pixels, and for every
    one of them both candidate frames -- pushed alone through both arithmetics -- are a near tie; the fused image
    differs on a fraction of a percent of its values, almost all by one count (truncating cast at an integer boundary),
    and every value off by two or more counts lies in the collapse footprint of a flipped selection.
    Measured (round 3): no flip at level 0, 13 / 12 / 4 / 4 / 3 at levels 1-5, largest gap 0.2 % of the bound; final
    image 88 170 values off by one count, 169 by two, 78 by more (max 10) of 72 M."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_report
    _FUSED.clear()
    H, W, N = 4000, 6000, 256
    buf = L.DeviceBuffer(H * W * 3 * 4 * N)
    L.synth_frames_device(buf.ptr, np.float32, H, W, 0, N, N)
    rep = parity_report.report(L, buf.ptr, N, H, W, np.float32)
    buf.free()
    assert rep["ok"], rep
    for row in rep["levels"]:
        assert row["energy_diff_over_bound_max"] <= 1.0 and row["selection_mismatch_rate"] < 1e-3, row
        if row["level"] >= 1:
            assert row["gauss_abs_diff_max_lsb"] <= row["gauss_bound_lsb"], row
    nt = rep["near_tie"]
    assert nt["checked"] == nt["pixels"] and nt.get("not_a_near_tie", 0) == 0 and nt.get("winner_energy_reproduced", True), nt
    hist = np.array(rep["final_abs_diff_counts_0_1_2_3plus"], float) / (H * W * 3)
    assert hist[1] < 5e-3 and hist[2] < 1e-5 and hist[3] < 1e-5, hist
    assert rep["final_pixels_off_by_2plus_outside_a_flip_footprint"] == 0 and rep["flip_footprint_fraction_of_image"] < 0.01, rep
    assert rep["base"]["fused_base_abs_diff_max_lsb"] < 0.9, rep["base"]


def _parity_tools():
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import parity_report
    return parity_report


def _assert_parity(rep, npx3, base_lsb_max=None):
    """`base_lsb_max`: None = the report's own verdict (`ok`: fused bases within 0.9 LSB of each other); a number = every
    other statement of the report individually and that bound on the fused-base difference instead"""
    if base_lsb_max is None:
        assert rep["ok"], rep
    else:
        assert rep["base"]["fused_base_abs_diff_max_lsb"] < base_lsb_max, rep["base"]
        assert rep["base"]["gauss_abs_diff_max_lsb"] <= rep["base"]["gauss_bound_lsb"], rep["base"]
    for row in rep["levels"]:
        assert row["energy_diff_over_bound_max"] <= 1.0, row
        if row["level"] >= 1:
            assert row["gauss_abs_diff_max_lsb"] <= row["gauss_bound_lsb"], row
    nt = rep["near_tie"]
    assert nt["checked"] == nt["pixels"] and nt.get("not_a_near_tie", 0) == 0 and nt.get("winner_energy_reproduced", True), nt
    assert rep["final_pixels_off_by_2plus_outside_a_flip_footprint"] == 0, rep
    hist = np.array(rep["final_abs_diff_counts_0_1_2_3plus"], float) / npx3
    return hist


def test_parity_report_on_real_frames(L):
    """The same report on NON-synthetic content: the six frames of tests/golden/img_jpg_crop (crops of the reference's own
    example stack, 256 x 384, three levels).  Real frames differ in their low-pass content, so here the base-level rule
    (pyramid.py:95-111) decides visible values: the report's base rows and the final-image histogram are what to read."""
    pr = _parity_tools()
    frames = pr.real_crop_frames()
    rep = pr.report_host_frames(L, frames)
    print("\n[parity, real frames]", {k: rep[k] for k in ("base", "final_abs_diff_counts_0_1_2_3plus", "final_max_abs_diff", "near_tie")},
          [(r["level"], r["selection_mismatches"]) for r in rep["levels"]])
    hist = _assert_parity(rep, frames[0].size)
    assert hist[1] < 5e-3 and hist[2:].sum() < 1e-4, hist


def test_parity_report_on_frames_with_an_exposure_ramp(L, oracle):
    """... and on the generator with a per-frame exposure ramp (gain 0.70 .. 1.30): the frames' base images differ by tens of
    counts, so a base-selection flip between the arithmetics would show in the fused image."""
    pr = _parity_tools()
    H, W, N = 2000, 3000, 24
    rep = pr.report_host_frames(L, pr.ramp_frames(H, W, N, oracle))
    print("\n[parity, exposure ramp]", {k: rep[k] for k in ("base", "final_abs_diff_counts_0_1_2_3plus", "final_max_abs_diff", "near_tie")},
          [(r["level"], r["selection_mismatches"]) for r in rep["levels"]])
    hist = _assert_parity(rep, H * W * 3)
    assert hist[1] < 5e-3 and hist[2:].sum() < 1e-4, hist


@pytest.mark.parametrize("dtype,n", [(np.uint8, 16), (np.uint16, 12)])
def test_parity_report_on_a_simulated_focus_stack(L, dtype, n):
    """... and on a simulated focus stack of a natural-looking scene (tools/parity_report.py::defocus_frames: 1 / f texture,
    hard edges, a clipped highlight, a nearly black corner, depth-dependent defocus, exposure drift, sensor noise): the
    frames' energies fall off smoothly around the focal plane, so neighbouring frames ARE close calls at many pixels -- the
    case where the two arithmetics could pick different frames.  Same gates as above: every flipped arg-max a proven near
    tie, every final value off by two or more (8-bit-equivalent) counts inside the footprint of a flip.  What this content
    adds (round 5): its frames' LOW-PASS images are nearly alike, so the base-level rule (pyramid.py:95-111: entropy of
    the gray image TRUNCATED to integers) flips on 1-2 % of the base pixels between the two arithmetics -- a discontinuity of
    the reference's own rule, which its float-32 and float-64 modes show against each other as well -- and the fused base
    images differ by up to 1.4 counts (16-bit frames, in 8-bit-equivalent counts; 8-bit frames: below 0.9): a smooth
    offset of a few 1e-5 of full scale over the collapse footprint of those pixels, recorded here rather than hidden
    behind the synthetic generator, whose frames' base images never disagree."""
    pr = _parity_tools()
    H, W = 2000, 3000
    rep = pr.report_host_frames(L, pr.defocus_frames(H, W, n, dtype))
    print("\n[parity, simulated focus stack]", np.dtype(dtype).name,
          {k: rep.get(k) for k in ("base", "final_abs_diff_counts_0_1_2_3plus", "final_abs_diff_counts_0_1_2_3plus_lsb8",
                                   "final_max_abs_diff", "near_tie")},
          [(r["level"], r["selection_mismatches"]) for r in rep["levels"]])
    _assert_parity(rep, H * W * 3, base_lsb_max=2.0)
    hist = np.array(rep.get("final_abs_diff_counts_0_1_2_3plus_lsb8", rep["final_abs_diff_counts_0_1_2_3plus"]), float) / (H * W * 3)
    assert hist[1] < 5e-3 and hist[2:].sum() < 1e-4, hist   # (16-bit frames: in 8-bit-equivalent counts)
    assert rep["final_max_abs_diff"] <= 5 * (257 if dtype == np.uint16 else 1), rep["final_max_abs_diff"]


def test_config5_two_bunches_50mp_u16_from_host(L, oracle):
    """config 5's shape on one GPU: bunches of 10 x 5760x8640 uint16 frames with overlap 2, pushed from HOST memory
    through the pinned upload path, one handle reused (stack.py:61-97: a bunch's output is truncated to the input
    dtype).  Per bunch: output corner == oracle on the cropped frames is not available for the collapsed image (the
    base level is global), so the check is on the level-0 state of the corner plus determinism of the reused handle."""
    H, W, NB, OV = 5760, 8640, 10, 2
    c, good = 128, 120
    st = L.Stack(H, W, in_dtype=np.uint16, out_dtype=np.uint16)
    assert st.levels == 7
    n_total = 2 * NB - OV
    frames = [oracle.synth_frame_u8(H, W, f, n_total).astype(np.uint16) * 257 for f in range(n_total)]
    outs = []
    for b0 in (0, NB - OV):
        st.reset()
        for f in range(b0, b0 + NB):
            st.push_frame(frames[f])
        so = oracle.StreamingOracle(c, c, np.uint16, levels=1)
        for f in range(b0, b0 + NB):
            so.push_frame(np.ascontiguousarray(frames[f][:c, :c]))
        assert np.array_equal(st.tap(L.TAP_INDEX, 0)[:good, :good], so.best_idx[0][:good, :good])
        assert np.array_equal(st.tap(L.TAP_ENERGY, 0)[:good, :good], so.best_e[0][:good, :good])
        assert np.array_equal(st.tap(L.TAP_FUSED_LAP, 0)[:good, :good], so.best_lap[0][:good, :good])
        out = st.finish()
        assert out.dtype == np.uint16 and out.shape == (H, W, 3)
        outs.append(out[::97, ::89].copy())
    # the second bunch again on the same handle: identical
    st.reset()
    for f in range(NB - OV, 2 * NB - OV):
        st.push_frame(frames[f])
    assert np.array_equal(st.finish()[::97, ::89], outs[1])
    st.close()


def _scene(H, W, seed=4):
    """broadband scene (octaves of smooth random fields + pixel noise), as tools/config4.py"""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    field = np.zeros((H, W), np.float32)
    for k in range(3, 9):
        g = rng.standard_normal((H // 2 ** k + 2, W // 2 ** k + 2)).astype(np.float32)
        field += ndimage.zoom(g, 2 ** k, order=1)[:H, :W] * (2.0 ** (k - 5))
    field = (field - field.min()) / (field.max() - field.min())
    noise = rng.integers(-6, 7, (H, W)).astype(np.float32)
    return np.stack([np.clip(30 + 190 * field + noise + 5 * c, 0, 255).astype(np.uint8) for c in range(3)], axis=-1)


def test_config4_128_frames_24mp_align_and_stack(L, oracle):
    """128 x 24 MP uint8 frames resident in HBM, each the same scene under a known similarity (focus-breathing like:
    0.02 deg, 1e-4 scale, (0.37, -0.21) px per frame of distance from the reference), registered on the device (ECC),
    warped with the blurred replicate border and fused (pipeline.align_and_stack_device).  Accuracy gate = the
    reference's own precision test (tests/test_0031_align_precision.py:62-65: 0.005 deg, 0.2 px, 1e-4 scale); the
    apply step of two frames is checked against the CPU restatement of align.py:238-251 on the full frame."""
    from shinestacker_amd.pipeline import align_and_stack_device
    H, W, N = 4000, 6000, 128
    ref = N // 2
    cx, cy = (W - 1) / 2, (H - 1) / 2
    scene = _scene(H, W)
    fb = H * W * 3
    buf = L.DeviceBuffer(N * fb)
    src = L.DeviceBuffer(fb)
    src.upload(scene)
    lib = L.load()
    truth = []
    bv = (C.c_double * 4)(0, 0, 0, 0)
    for f in range(N):
        d = f - ref
        t, s = np.deg2rad(0.02 * d), 1 + 1e-4 * d
        a, b = s * np.cos(t), s * np.sin(t)
        T = np.array([[a, -b, cx - a * cx + b * cy + 0.37 * d], [b, a, cy - b * cx - a * cy - 0.21 * d]])
        truth.append(T)
        if d == 0:
            L.check(lib.mi_memcpy_d2d(0, buf.ptr + f * fb, src.ptr, fb))
        else:
            mm = (C.c_double * 6)(*T.reshape(6))
            L.check(lib.mi_warp_affine_device(0, None, src.ptr, buf.ptr + f * fb, None, None, H, W, L.MI_U8, mm,
                                              L.BORDER_REPLICATE, bv, 21, 50.0))
    L.check(lib.mi_device_synchronize(0))
    fused, tr, ccs = align_and_stack_device(buf.ptr, N, H, W, np.uint8, ref_idx=ref)
    assert fused.shape == (H, W, 3) and fused.dtype == np.uint8 and tr[ref] is None
    for f in range(N):
        if f == ref:
            continue
        A = truth[f][:, :2]
        Ai = np.linalg.inv(A)
        want = np.hstack([Ai, -Ai @ truth[f][:, 2:3]])       # moving -> reference
        m = tr[f]
        ang = np.rad2deg(np.arctan2(m[1, 0], m[0, 0]) - np.arctan2(want[1, 0], want[0, 0]))
        sc = np.hypot(m[0, 0], m[1, 0]) - np.hypot(want[0, 0], want[1, 0])
        ctr = np.array([cx, cy, 1.0])
        sh = np.abs(m @ ctr - want @ ctr).max()
        assert abs(ang) < 0.005 and abs(sc) < 1e-4 and sh < 0.2, (f, ang, sc, sh)
        assert ccs[f] > 0.9
    # all frames show the same scene once aligned: the fused image is the scene up to interpolation blur
    inner = (slice(200, H - 200), slice(200, W - 200))
    assert np.abs(fused[inner].astype(np.int16) - scene[inner].astype(np.int16)).mean() < 4.0
    # the apply step at full size against the CPU restatement (warp + mask + blurred border), two frames
    for f in (0, N - 1):
        mov = buf.download((H, W, 3), np.uint8, offset=f * fb)
        got = L.warp_affine(mov, tr[f])
        want = oracle.warp_affine(mov, tr[f])
        assert np.array_equal(got, want), f
    buf.free()
    src.free()


def test_config5_two_stage_16_bunches_50mp_u16(L, oracle):
    """config 5 end to end on one GPU at a meaningful length (pipeline.bunches_then_stack, stack.py:61-113): 130 frames
    of 5760 x 8640 uint16 from HOST memory -> 16 bunches of 10 with overlap 2 (asynchronous pinned upload, one handle
    reused) -> the 16 bunch results, kept on the device as uint16 (what the reference's intermediate files hold), fused
    once more.  Checked: every bunch's level-0 state in a corner == the oracle fed that bunch's frames cropped; the final
    stack's level-0 state in the corner == the oracle fed the ACTUAL bunch results cropped (they went through the
    truncating cast); bunch geometry == get_bunches."""
    from shinestacker_amd.actions import get_bunches
    from shinestacker_amd.pipeline import bunches_then_stack
    H, W, N = 5760, 8640, 130
    c, good = 128, 120
    per = H * W * 3 * 2
    gen = L.DeviceBuffer(per)

    def get_frame(i):            # generated on the device, brought to the host, pushed from there like a decoded file
        L.synth_frames_device(gen.ptr, np.uint16, H, W, i, 1, N)
        return gen.download((H, W, 3), np.uint16)

    def crop(i):
        return oracle.synth_crop_u8(H, W, i, N, 0, 0, c, c).astype(np.uint16) * 257
    want = get_bunches(list(range(N)), 10, 2)
    assert len(want) == 16 and want[0] == list(range(10)) and want[-1][0] == 120
    checked = []

    def on_bunch(k, st):
        so = oracle.StreamingOracle(c, c, np.uint16, levels=1, arith="separable")   # bunches_then_stack's default
        for i in want[k]:
            so.push_frame(crop(i))
        assert np.array_equal(st.tap(L.TAP_FUSED_LAP, 0)[:good, :good], so.best_lap[0][:good, :good]), k
        assert np.array_equal(st.tap(L.TAP_INDEX, 0)[:good, :good], so.best_idx[0][:good, :good]), k
        checked.append(k)

    final = {}

    def on_final(st2, results):
        so = oracle.StreamingOracle(c, c, np.uint16, levels=1, arith="separable")
        for k in range(len(want)):
            # rows 0..c-1 of bunch result k, then the corner of them
            rows = results.download((c, W, 3), np.uint16, offset=k * per)
            so.push_frame(np.ascontiguousarray(rows[:, :c]))
        final["lap"] = bool(np.array_equal(st2.tap(L.TAP_FUSED_LAP, 0)[:good, :good], so.best_lap[0][:good, :good]))
        final["idx"] = bool(np.array_equal(st2.tap(L.TAP_INDEX, 0)[:good, :good], so.best_idx[0][:good, :good]))

    out, bunches = bunches_then_stack(get_frame, N, H, W, np.uint16, on_bunch=on_bunch, on_final=on_final)
    gen.free()
    assert bunches == want and checked == list(range(16)) and final == {"lap": True, "idx": True}
    assert out.shape == (H, W, 3) and out.dtype == np.uint16 and 40 * 257 < out.mean() < 215 * 257


@pytest.mark.parametrize("arith", ["separable", "exact"])
def test_frames_beyond_67_megapixels_tiled_equals_simple(L, arith):
    """In-frame offsets are 32-bit (frames up to 357 MP are accepted, mi_stack_create); products of pixel indices must not
    be taken with the 24-bit multiplier once a LEVEL has more than 2^24 pixels -- 108 MP frames have 27 M of them at level 1.
    Three 9000 x 12000 u8 frames: the LDS-tiled kernels against the one-thread-per-output implementation."""
    H, W, N = 9000, 12000, 3
    per = H * W * 3
    buf = L.DeviceBuffer(per * N)
    L.synth_frames_device(buf.ptr, np.uint8, H, W, 0, N, N)
    outs, idx1 = {}, {}
    for impl in (L.IMPL_SIMPLE, L.IMPL_TILED):
        st = L.Stack(H, W, in_dtype=np.uint8, arith=arith, impl=impl)
        if impl == L.IMPL_SIMPLE:
            for f in range(N):
                st.push_frames_device(buf.ptr + f * per, 1, per)
        else:
            st.push_frames_device(buf.ptr, N, per)
        idx1[impl] = st.tap(L.TAP_INDEX, 1)
        outs[impl] = st.finish()
        st.close()
    buf.free()
    assert np.array_equal(idx1[L.IMPL_SIMPLE], idx1[L.IMPL_TILED])
    assert np.array_equal(outs[L.IMPL_SIMPLE], outs[L.IMPL_TILED])


def test_config5_second_stage_over_128_resident_results_50mp_u16(L, oracle):
    """BASELINE config 5's SECOND stage at its full length: 1024 frames in bunches of 10 with overlap 2 leave 128 bunch
    results, each a 5760 x 8640 uint16 image that stays on the device (37 GB), and `FocusStack` fuses those once more
    (stack.py:101-113).  Here 128 resident uint16 frames of that size stand in for the results (the generator's frames:
    the stage only sees size, type and count) and are fused in one resident push -- 7 levels, one batch; verified like
    the benchmark verifies config 2 (six windows x three levels == the oracle on the cropped frames)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    H, W, N = 5760, 8640, 128
    free, _total = L.mem_info()
    per = H * W * 3 * 2
    if free < per * N + (90 << 30):
        pytest.skip("needs ~130 GB of free device memory")
    buf = L.DeviceBuffer(per * N)
    L.synth_frames_device(buf.ptr, np.uint16, H, W, 0, N, N)
    st = L.Stack(H, W, in_dtype=np.uint16, out_dtype=np.uint16, arith="separable")
    assert st.levels == 7
    st.push_frames_device(buf.ptr, N)
    out = st.finish()
    args = types.SimpleNamespace(height=H, width=W, dtype="u16", arith="separable")
    v = bench.verify(L, st, args, N, 1)
    assert v["band_match"] > 0.9 and v["crops_equal"] and len(v["crops"]) == 6, v
    assert out.shape == (H, W, 3) and out.dtype == np.uint16 and 40 * 257 < out.mean() < 215 * 257
    st.close()
    buf.free()
