"""CPU: host logic of align_images / AlignFrames (no GPU: the apply step is injected)."""
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN
from shinestacker_amd import AlignFrames, CombinedActions, StackJob, align_images
from shinestacker_amd.align import img_subsample
from shinestacker_amd.errors import AlignmentError, InvalidOptionError, ShapeError


def fake_estimator(n, m):
    calls = []

    def est(i0, i1, fc, mc, ac):
        calls.append((i0.shape, i1.shape))
        return n, (None if m is None else np.array(m, dtype=np.float64))
    est.calls = calls
    return est


def record_apply():
    seen = {}

    def apply(img, m, cfg):
        seen["m"], seen["cfg"] = np.array(m), cfg
        return img.copy()
    apply.seen = seen
    return apply


def test_subsample_rescales_translation_and_casts_to_float32():
    img = np.zeros((64, 80, 3), np.uint8)
    est = fake_estimator(500, [[0.99, 0.01, 1.5], [-0.01, 0.99, -2.25]])
    ap = record_apply()
    n, m, warp = align_images(img, img, estimator=est, apply_fn=ap,
                              alignment_config={'fast_subsampling': True})
    assert n == 500 and warp.shape == img.shape
    assert est.calls == [((32, 40, 3), (32, 40, 3))]           # default subsample = 2
    assert m.dtype == np.float32                                 # align.py:220-223
    assert np.allclose(m, [[0.99, 0.01, 3.0], [-0.01, 0.99, -4.5]])
    assert ap.seen["cfg"]["border_mode"] == "BORDER_REPLICATE_BLUR" and ap.seen["cfg"]["border_blur"] == 50


def test_retry_without_subsampling_when_few_matches():
    img = np.zeros((64, 80, 3), np.uint8)
    results = iter([(10, [[1, 0, 1], [0, 1, 1]]), (150, [[1, 0, 2], [0, 1, 2]])])
    shapes, warnings = [], []

    def est(i0, i1, fc, mc, ac):
        shapes.append(i0.shape)
        n, m = next(results)
        return n, np.array(m, float)
    n, m, _ = align_images(img, img, estimator=est, apply_fn=record_apply(),
                           alignment_config={'fast_subsampling': True},
                           callbacks={'warning': warnings.append})
    assert shapes == [(32, 40, 3), (64, 80, 3)] and n == 150
    assert m.dtype == np.float64 and m[0, 2] == 2                # no rescale at subsample 1
    assert "retrying without subsampling" in warnings[0]


def test_too_few_matches_returns_none():
    img = np.zeros((16, 16, 3), np.uint8)
    n, m, warp = align_images(img, img, estimator=fake_estimator(2, None), apply_fn=record_apply(),
                              alignment_config={'subsample': 1})
    assert (n, m, warp) == (2, None, None)


def test_option_and_shape_errors():
    img = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(InvalidOptionError):
        align_images(img, img, alignment_config={'border_mode': 'BORDER_WRAP'}, estimator=fake_estimator(5, None))
    with pytest.raises(InvalidOptionError):
        align_images(img, img, alignment_config={'transform': 'ALIGN_AFFINE'}, estimator=fake_estimator(5, None))
    with pytest.raises(ShapeError):
        align_images(img, np.zeros((16, 17, 3), np.uint8), estimator=fake_estimator(5, None))


def test_estimator_selection():
    """estimator='auto' (the default): the reference's recipe when OpenCV is importable, the GPU ECC estimator
    otherwise; 'opencv' without OpenCV says so; anything else is an InvalidOptionError."""
    import shinestacker_amd.align as al
    img = np.zeros((16, 16, 3), np.uint8)
    if not al.have_opencv():
        with pytest.raises(RuntimeError, match="OpenCV"):
            align_images(img, img, alignment_config={'subsample': 1}, estimator="opencv")
        assert al.resolve_estimator("auto").__name__ == "estimate"          # the ECC closure
    else:
        assert al.resolve_estimator(None) is al.opencv_estimator
    f = fake_estimator(5, None)
    assert al.resolve_estimator(f) is f
    with pytest.raises(InvalidOptionError):
        al.resolve_estimator("sift-gpu")


def test_feature_config_validation_as_reference():
    """align.py:71-87: the reference refuses these combinations with these messages (tests/test_0032_align_methods.py)."""
    from shinestacker_amd.align import validate_align_config
    bad = [("SIFT", "SIFT", "NORM_HAMMING", "Descriptor SIFT requires matching method KNN"),
           ("ORB", "AKAZE", "NORM_HAMMING", "Detector ORB and descriptor AKAZE require matching method KNN"),
           ("BRISK", "AKAZE", "KNN", "Detector BRISK is incompatible with descriptor AKAZE"),
           ("SURF", "AKAZE", "KNN", "Detector SURF is incompatible with descriptor AKAZE"),
           ("SIFT", "ORB", "KNN", "Detector SIFT requires descriptor SIFT"),
           ("ORB", "ORB", "KNN", "Detector ORB and descriptor ORB require matching method Hamming distance")]
    for det, des, mm, msg in bad:
        with pytest.raises(ValueError, match=msg):
            validate_align_config(det, des, mm)
    for det, des, mm in [("SIFT", "SIFT", "KNN"), ("ORB", "ORB", "NORM_HAMMING"), ("AKAZE", "AKAZE", "NORM_HAMMING"),
                         ("SURF", "ORB", "NORM_HAMMING"), ("ORB", "SIFT", "KNN")]:
        validate_align_config(det, des, mm)
    img = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(ValueError, match="requires descriptor SIFT"):      # whichever estimator runs
        align_images(img, img, feature_config={'detector': 'SIFT', 'descriptor': 'ORB'}, estimator=fake_estimator(5, None))


def test_homography_transform_path():
    """ALIGN_HOMOGRAPHY (align.py:212-221, :231-237): four matches needed, the sub-sampled estimate is conjugated with
    the corner-to-corner scalings, the apply step receives a 3x3 matrix."""
    from shinestacker_amd.align import rescale_transform
    img = np.zeros((40, 60, 3), np.uint8)
    Hs = np.array([[1.01, 0.02, 3.0], [-0.01, 0.99, -2.0], [1e-4, -2e-4, 1.0]])
    seen = {}

    def apply_fn(im, m, cfg):
        seen['m'] = np.asarray(m)
        return im
    n, m, warp = align_images(img, img, estimator=fake_estimator(200, Hs), apply_fn=apply_fn,
                              alignment_config={'transform': 'ALIGN_HOMOGRAPHY', 'subsample': 2, 'fast_subsampling': True})
    assert n == 200 and m.shape == (3, 3) and seen['m'].shape == (3, 3)
    up, down = np.diag([2.0, 2.0, 1.0]), np.diag([0.5, 0.5, 1.0])
    assert np.allclose(m, up @ Hs @ down, rtol=0, atol=1e-15)
    # a point of the sub-sampled image maps consistently: full-res map of 2p == 2 * sub-res map of p
    p = np.array([7.0, 5.0, 1.0])
    q_sub = Hs @ p
    q_full = m @ np.array([14.0, 10.0, 1.0])
    assert np.allclose(q_full[:2] / q_full[2], 2 * q_sub[:2] / q_sub[2])
    # three matches are too few for a homography (min_matches = 4), and a similarity is embedded as 3x3
    n, m, warp = align_images(img, img, estimator=fake_estimator(3, Hs), apply_fn=apply_fn,
                              alignment_config={'transform': 'ALIGN_HOMOGRAPHY', 'subsample': 1})
    assert (m, warp) == (None, None)
    n, m, _ = align_images(img, img, estimator=fake_estimator(9, [[1, 0, 2], [0, 1, 3]]), apply_fn=apply_fn,
                           alignment_config={'transform': 'ALIGN_HOMOGRAPHY', 'subsample': 1})
    assert m.shape == (3, 3) and np.array_equal(m[2], [0, 0, 1])
    assert rescale_transform(np.eye(2, 3), 'ALIGN_RIGID', 4, (8, 8), (2, 2)).dtype == np.float32


def test_area_subsample_rounds_half_up():
    a = np.array([[[1], [2], [3], [4]], [[2], [2], [3], [3]]], np.uint8)  # 2x4x1
    out = img_subsample(a, 2, fast=False)
    assert out.shape == (1, 2, 1) and out[0, 0, 0] == 2 and out[0, 1, 0] == 3   # 7/4 -> 2, 13/4 -> 3
    assert np.array_equal(img_subsample(a, 2, fast=True), a[::2, ::2])


def test_area_subsample_rules_of_cv2_resize():
    """The integer-factor INTER_AREA path as img_subsample restates it (docstring): s == 2 rounds halves up, other
    factors round the float32 product half to even, the output size is round-half-even(dim / s) and the partial blocks
    of sizes that do not divide are means over the pixels that exist."""
    a = np.zeros((4, 8, 1), np.uint8)
    a[:, :4] = [[[0], [0], [0], [0]], [[0], [0], [0], [0]], [[0], [0], [0], [0]], [[0], [0], [4], [4]]]   # sum 8 / 16 = 0.5
    a[:, 4:] = 3
    a[0, 4] = 11                                                                                           # sum 56 / 16 = 3.5
    out = img_subsample(a, 4, fast=False)
    assert out.shape == (1, 2, 1) and out[0, 0, 0] == 0 and out[0, 1, 0] == 4      # 0.5 -> 0, 3.5 -> 4 (half to even)
    b = np.arange(5 * 7, dtype=np.uint16).reshape(5, 7, 1) * 1000
    out = img_subsample(b, 2, fast=False)
    assert out.shape == (2, 4, 1)                                                  # rint(2.5) = 2 rows, rint(3.5) = 4 columns
    assert out[0, 0, 0] == (0 + 1000 + 7000 + 8000 + 2) // 4
    assert out[0, 3, 0] == int(np.rint((6000 + 13000) / 2))                        # last column: the two pixels that exist
    assert img_subsample(b, 1, fast=False) is b


def test_align_frames_subaction_protocol(tmp_path, monkeypatch):
    """AlignFrames inside CombinedActions: reference frame untouched, AlignmentError on few matches."""
    import shinestacker_amd.align as al
    monkeypatch.setattr(al, "apply_transform", lambda img, m, cfg, device=0: img[::-1].copy())
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "input"))
    for n in sorted(os.listdir(os.path.join(GOLDEN, "img_jpg_crop")))[:3]:
        shutil.copy(os.path.join(GOLDEN, "img_jpg_crop", n), os.path.join(work, "input", n))
    af = AlignFrames(estimator=fake_estimator(200, [[1, 0, 0], [0, 1, 0]]), subsample=1)
    job = StackJob("job", work, input_path="input")
    job.add_action(CombinedActions("align", [af], output_path="aligned"))
    job.run()
    from shinestacker_amd.imageio import read_img
    names = sorted(os.listdir(os.path.join(work, "input")))
    ref = read_img(os.path.join(work, "input", names[1]))
    assert np.array_equal(read_img(os.path.join(work, "aligned", names[1])), ref)       # ref_idx = 1
    mov = read_img(os.path.join(work, "input", names[0]))
    assert np.array_equal(read_img(os.path.join(work, "aligned", names[0])), mov[::-1])
    assert list(af.n_matches) == [200, 0, 200]
    bad = AlignFrames(estimator=fake_estimator(1, None), subsample=1)
    job = StackJob("job", work, input_path="input")
    job.add_action(CombinedActions("align2", [bad], output_path="aligned2"))
    with pytest.raises(AlignmentError):
        job.run()
