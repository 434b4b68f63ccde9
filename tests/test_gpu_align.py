"""GPU: the alignment apply step (mi_warp_affine) against oracle/align_oracle.c -- bit-exact --
plus analytic known answers."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


def rot(theta_deg, s, tx, ty, cx=0.0, cy=0.0):
    t = np.deg2rad(theta_deg)
    a, b = s * np.cos(t), s * np.sin(t)
    return [[a, b, (1 - a) * cx - b * cy + tx], [-b, a, b * cx + (1 - a) * cy + ty]]


TRANSFORMS = [
    [[1, 0, 0], [0, 1, 0]],
    [[1, 0, 7], [0, 1, -3]],
    [[1, 0, 0.5], [0, 1, 0.25]],
    rot(0.37, 1.0003, 3.37, -2.21, 150, 100),
    rot(15.0, 0.9, 30, 20, 128, 96),
    rot(-4.0, 1.2, -45.5, 61.3),
    [[1, 0, 500], [0, 1, 0]],          # entirely out of frame
]


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("ti", range(len(TRANSFORMS)))
def test_warp_matches_oracle(L, oracle, dtype, mode, ti):
    rng = np.random.default_rng(100 + ti)
    hi = 256 if dtype == np.uint8 else 65536
    img = rng.integers(0, hi, (197, 263, 3)).astype(dtype)
    M = np.array(TRANSFORMS[ti], dtype=np.float64)
    bv = (10, 200, 3000 if dtype == np.uint16 else 77, 0)
    want, wmask = oracle.warp_affine(img, M, border_mode=mode, border_value=bv, want_mask=True)
    got, gmask = L.warp_affine(img, M, border_mode=mode, border_value=bv, want_mask=True)
    assert np.array_equal(gmask, wmask)
    assert got.dtype == img.dtype and np.array_equal(got, want)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("shape", [(100, 800), (131, 1030), (70, 777)])
@pytest.mark.parametrize("tr", [(0.0, 1.0, 3.4, -2.2), (0.4, 1.002, 5.0, -3.0), (-1.1, 0.997, -9.5, 6.25), (4.0, 1.0, 0.0, 0.0)])
def test_warp_with_an_outer_ring_of_tiles(L, oracle, dtype, mode, shape, tr):
    """Images of three and more 256 x 32 (16) tiles each way: the kernel runs the outer ring of tiles first, each as four
    workgroups, and the inner tiles through LDS (8-bit: the dot-product form of cv2's fixed-point weights); widths that
    are and are not a multiple of four (dword / byte stores); a rotation whose source window no longer fits the LDS
    budget (all tiles per pixel).  Bit-equal to the oracle, mask included."""
    h, w = shape
    rng = np.random.default_rng(h * 7 + w)
    hi = 256 if dtype == np.uint8 else 65536
    img = rng.integers(0, hi, (h, w, 3)).astype(dtype)
    M = np.array(rot(tr[0], tr[1], tr[2], tr[3], (w - 1) / 2, (h - 1) / 2), dtype=np.float64)
    want, wmask = oracle.warp_affine(img, M, border_mode=mode, border_value=(0, 0, 0, 0), want_mask=True)
    got, gmask = L.warp_affine(img, M, border_mode=mode, border_value=(0, 0, 0, 0), want_mask=True)
    assert np.array_equal(gmask, wmask)
    assert np.array_equal(got, want)


def test_float32_matrix_as_the_reference_passes_it(L, oracle):
    """After sub-sampling the reference hands cv2 a float32 matrix (align.py:220-223)."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (120, 90, 3), dtype=np.uint8)
    m32 = np.array(rot(1.3, 0.998, 4.4, -3.3), dtype=np.float32)
    assert np.array_equal(L.warp_affine(img, m32), oracle.warp_affine(img, m32.astype(np.float64)))


def test_identity_and_integer_shift_are_exact(L):
    rng = np.random.default_rng(6)
    img = rng.integers(0, 65536, (64, 80, 3), dtype=np.uint16)
    out, mask = L.warp_affine(img, [[1, 0, 0], [0, 1, 0]], want_mask=True)
    assert np.array_equal(out, img) and mask.all()
    out, mask = L.warp_affine(img, [[1, 0, 5], [0, 1, 2]], border_mode=L.BORDER_REPLICATE, want_mask=True)
    assert np.array_equal(out[2:, 5:], img[:-2, :-5])
    assert np.array_equal(out[:2, 5:], np.broadcast_to(img[0, :-5], (2, 75, 3)))   # replicated edge
    assert not mask[:2].any() and not mask[:, :5].any() and mask[2:, 5:].all()
    out = L.warp_affine(img, [[1, 0, 5], [0, 1, 2]], border_mode=L.BORDER_CONSTANT, border_value=(1, 2, 3, 0))
    assert np.array_equal(out[0, 0], [1, 2, 3])


def test_half_pixel_shift_of_a_ramp(L):
    """bilinear interpolation of a linear ramp is exact up to the final rounding."""
    ramp = np.tile((np.arange(100) * 2)[None, :, None], (40, 1, 3)).astype(np.uint8)
    out = L.warp_affine(ramp, [[1, 0, 0.5], [0, 1, 0.25]], border_mode=L.BORDER_REPLICATE)
    assert np.array_equal(out[5, 1:50, 0], np.arange(1, 50) * 2 - 1)


def test_border_blur_only_touches_out_of_frame_pixels(L):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (150, 200, 3), dtype=np.uint8)
    M = [[1, 0, 12.5], [0, 1, 0]]
    plain = L.warp_affine(img, M, border_mode=L.BORDER_REPLICATE)
    blur, mask = L.warp_affine(img, M, border_mode=L.BORDER_REPLICATE_BLUR, want_mask=True)
    assert np.array_equal(blur[mask == 1], plain[mask == 1])
    assert not np.array_equal(blur[mask == 0], plain[mask == 0])
    assert (mask == 0).sum() == 150 * 12      # columns 0..11: less than half of the footprint in frame


def test_align_images_end_to_end_with_a_known_transform(L, oracle):
    """align_images with an injected estimator == oracle applied with the rescaled float32 matrix."""
    from shinestacker_amd import align_images
    rng = np.random.default_rng(8)
    ref = rng.integers(0, 256, (128, 160, 3), dtype=np.uint8)
    mov = rng.integers(0, 256, (128, 160, 3), dtype=np.uint8)
    m_sub = np.array(rot(0.8, 1.001, 1.75, -0.6))

    def est(i0, i1, fc, mc, ac):
        return 321, m_sub
    n, m, warp = align_images(ref, mov, estimator=est, alignment_config={'fast_subsampling': True})
    m_full = np.empty((2, 3), np.float32)
    m_full[:, :2] = m_sub[:, :2]
    m_full[:, 2] = m_sub[:, 2] * 2
    assert n == 321 and np.array_equal(m, m_full)
    assert np.array_equal(warp, oracle.warp_affine(mov, m_full.astype(np.float64)))


def test_align_and_stack_pipeline_equals_two_step_path(L, oracle):
    """In-memory align -> stack == align_images per frame, then the stacker on the aligned frames."""
    from shinestacker_amd import align_images
    from shinestacker_amd.pipeline import align_and_stack
    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, (160, 208, 3), dtype=np.uint8)
    frames = [np.roll(base, (k - 2, 2 * (k - 2)), axis=(0, 1)) for k in range(5)]
    transforms = {k: np.array(rot(0.1 * (k - 2), 1 + 1e-3 * (k - 2), -2.0 * (k - 2) + 0.3, -(k - 2) + 0.2))
                  for k in range(5)}
    key = {f.tobytes()[:64]: k for k, f in enumerate(frames)}

    def est(i0, i1, fc, mc, ac):
        return 500, transforms[key[np.ascontiguousarray(i0).tobytes()[:64]]]
    cfg = {'subsample': 1}
    fused, matches = align_and_stack(frames, ref_idx=2, estimator=est, alignment_config=cfg,
                                     batch_frames=2)
    aligned = []
    for k, f in enumerate(frames):
        if k == 2:
            aligned.append(f)
        else:
            _n, _m, wimg = align_images(frames[2], f, estimator=est, alignment_config=cfg)
            aligned.append(wimg)
    so = oracle.StreamingOracle(160, 208, np.uint8, arith="separable")   # the default of every high-level entry point
    for f in aligned:
        so.push_frame(f)
    assert matches == [500, 500, 0, 500, 500]
    assert np.array_equal(fused, so.finish())


def test_project_align_balance_stack_on_files(hiplib, oracle, tmp_path):
    """A whole example-project shape (stack-from-frames.fsp): CombinedActions[AlignFrames(GPU ECC), BalanceFrames]
    -> FocusStack, on files, with the codec work in the background (io_threads / decode_threads) and strictly
    sequential: identical output files."""
    from shinestacker_amd import (AlignFrames, BalanceFrames, CombinedActions, FocusStack, PyramidStack, StackJob)
    from shinestacker_amd.align import ecc_estimator
    from shinestacker_amd.imageio import read_img, write_img
    from test_gpu_ecc import make_pair, similarity
    hiplib.require_device()
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "in"))
    h, w = 256, 384
    for f in range(5):
        d = f - 2
        T = similarity(0.15 * d, 1 + 4e-4 * d, 1.3 * d, -0.9 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=31, noise=2.0)
        fr = ref if d == 0 else mov
        write_img(os.path.join(work, "in", f"f{f}.png"), np.clip(fr * (1.0 + 0.1 * d), 0, 255).astype(np.uint8))
    results = []
    for tag, io_threads, dec in (("bg", 2, 4), ("seq", 0, 1)):
        job = StackJob("job", work, input_path="in")
        job.add_action(CombinedActions(f"align-{tag}", [AlignFrames(estimator=ecc_estimator(), subsample=1),
                                                        BalanceFrames(subsample=1)],
                                       output_path=f"aligned-{tag}", io_threads=io_threads))
        job.add_action(FocusStack(f"stack-{tag}", PyramidStack(decode_threads=dec, arith="exact"), input_path=f"aligned-{tag}",
                                  output_path=f"stack-{tag}"))
        job.run()
        aligned = [read_img(os.path.join(work, f"aligned-{tag}", n))
                   for n in sorted(os.listdir(os.path.join(work, f"aligned-{tag}")))]
        out = sorted(os.listdir(os.path.join(work, f"stack-{tag}")))
        assert len(aligned) == 5 and len(out) == 1
        results.append((aligned, read_img(os.path.join(work, f"stack-{tag}", out[0]))))
    for a, b in zip(results[0][0], results[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(results[0][1], results[1][1])
    # and the stack of the aligned files is what the oracle makes of them
    so = oracle.StreamingOracle(h, w, np.uint8, keep_gauss=False)
    for a in results[0][0]:
        so.push_frame(a)
    assert np.array_equal(results[0][1], so.finish())


def test_project_align_balance_sharded_over_ranks(hiplib, oracle, tmp_path):
    """The same CombinedActions[AlignFrames, BalanceFrames] with the frames split over two ranks (shard=(rank, 2),
    SURVEY 8(e)): the per-frame tables of the sub-actions are indexed by the GLOBAL frame index, so rank 1 (frames 3, 4
    of 5) must not run off arrays sized by its own block; the files equal the unsharded run's."""
    from shinestacker_amd import AlignFrames, BalanceFrames, CombinedActions, StackJob
    from shinestacker_amd.align import ecc_estimator
    from shinestacker_amd.imageio import read_img, write_img
    from test_gpu_ecc import make_pair, similarity
    hiplib.require_device()
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "in"))
    h, w = 192, 256
    for f in range(5):
        d = f - 2
        T = similarity(0.1 * d, 1 + 3e-4 * d, 1.1 * d, -0.7 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=5, noise=2.0)
        write_img(os.path.join(work, "in", f"f{f}.png"), np.clip((ref if d == 0 else mov) * (1.0 + 0.08 * d), 0, 255).astype(np.uint8))

    def run(out, shard):
        job = StackJob("job", work, input_path="in")
        job.add_action(CombinedActions("align", [AlignFrames(estimator=ecc_estimator(), subsample=1), BalanceFrames(subsample=1)],
                                       output_path=out, shard=shard))
        job.run()
    run("whole", None)
    for rank in (0, 1):
        run("sharded", (rank, 2))
    names = sorted(os.listdir(os.path.join(work, "whole")))
    assert sorted(n for n in os.listdir(os.path.join(work, "sharded")) if not n.startswith(".")) == names and len(names) == 5
    for n in names:
        assert np.array_equal(read_img(os.path.join(work, "whole", n)), read_img(os.path.join(work, "sharded", n))), n


# ---------------------------------------------------------------- ALIGN_HOMOGRAPHY apply (align.py:231-237)
HOMOGRAPHIES = {
    "similarity": [[0.999, -0.012, 2.3], [0.012, 0.999, -1.7], [0, 0, 1]],
    "mild_perspective": [[1.002, 0.004, -3.1], [-0.003, 0.998, 4.4], [1.5e-5, -2.5e-5, 1.0]],
    "strong_perspective": [[0.9, 0.1, 12.0], [-0.08, 1.1, -9.0], [6e-4, 3e-4, 1.0]],
    "scaled": [[2.0, 0.02, -30.0], [0.01, 2.0, -20.0], [1e-5, 0, 2.0]],     # homogeneous scale 2
}


@pytest.mark.parametrize("name", sorted(HOMOGRAPHIES))
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_warp_perspective_equals_oracle(L, oracle, name, dtype, mode):
    """mi_warp_perspective (image, mask, blurred border) == oracle/align_oracle.c's restatement of cv2.warpPerspective,
    odd sizes (the 64-column blocks of the coordinate recurrence end mid-block), all three border modes."""
    rng = np.random.default_rng(11)
    hi = 256 if dtype == np.uint8 else 65536
    img = rng.integers(0, hi, (133, 203, 3)).astype(dtype)
    M = np.array(HOMOGRAPHIES[name], np.float64)
    bv = (7, 250, 99, 0)
    got, gmask = L.warp_perspective(img, M, border_mode=mode, border_value=bv, want_mask=True)
    want, wmask = oracle.warp_perspective(img, M, border_mode=mode, border_value=bv, want_mask=True)
    assert np.array_equal(gmask, wmask)
    assert np.array_equal(got, want)


def test_align_images_homography_on_gpu(L, oracle):
    from shinestacker_amd.align import align_images
    """align_images with ALIGN_HOMOGRAPHY end to end on the device apply path: an injected 3x3 estimate found at
    sub-sample 2 is rescaled (align.py:213-221) and applied by mi_warp_perspective == the oracle's warp of the
    rescaled matrix."""
    rng = np.random.default_rng(2)
    ref = rng.integers(0, 256, (120, 160, 3)).astype(np.uint8)
    Hs = np.array([[1.004, 0.006, 1.2], [-0.005, 0.997, -0.8], [2e-5, -1e-5, 1.0]])
    n, m, warp = align_images(ref, ref, estimator=lambda a, b, fc, mc, ac: (300, Hs),
                              alignment_config={'transform': 'ALIGN_HOMOGRAPHY', 'subsample': 2, 'fast_subsampling': True})
    assert n == 300 and m.shape == (3, 3)
    assert np.array_equal(warp, oracle.warp_perspective(ref, m))


def test_align_frames_default_estimator_runs_out_of_the_box(hiplib, oracle, tmp_path):
    """The reference's AlignFrames() works with no arguments (align.py:90-151 on OpenCV); here the default estimator is
    'auto': OpenCV's recipe when cv2 is importable, the GPU ECC estimator otherwise.  stack-from-frames.fsp shaped job
    (docs/job.md: step_process on, sub-sample 2 with the area mean) on shifted copies of one scene."""
    from shinestacker_amd import AlignFrames, CombinedActions, StackJob
    from shinestacker_amd.imageio import read_img, write_img
    from test_gpu_ecc import make_pair, similarity
    hiplib.require_device()
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "in"))
    h, w = 256, 384
    truth = []
    for f in range(5):
        d = f - 2
        T = similarity(0.1 * d, 1 + 3e-4 * d, 0.9 * d, -0.6 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=8, noise=1.5)
        write_img(os.path.join(work, "in", f"f{f}.png"), (ref if d == 0 else mov).astype(np.uint8))
        truth.append(T)
    af = AlignFrames()                        # no estimator argument
    job = StackJob("job", work, input_path="in")
    job.add_action(CombinedActions("align", [af], output_path="aligned", step_process=True))
    job.run()
    names = sorted(os.listdir(os.path.join(work, "aligned")))
    assert len(names) == 5
    refimg = read_img(os.path.join(work, "in", "f2.png")).astype(np.int16)
    inner = (slice(24, h - 24), slice(24, w - 24))
    for n_ in names:
        a = read_img(os.path.join(work, "aligned", n_)).astype(np.int16)
        before = np.abs(read_img(os.path.join(work, "in", n_)).astype(np.int16)[inner] - refimg[inner]).mean()
        after = np.abs(a[inner] - refimg[inner]).mean()
        assert after <= max(before, 1e-9) + 1e-9 and after < 6.0, (n_, before, after)
