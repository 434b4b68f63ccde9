"""CPU: `opencv_estimator` (shinestacker_amd/align.py) -- the reference's feature estimator, algorithms/align.py:48-151 --
executed against a STAND-IN `cv2` module.  There is no OpenCV in this image, so none of OpenCV's arithmetic is tested
here; what is tested is everything the mirror itself decides: which detector / descriptor factories are called and on
what (8-bit gray, uint16 shifted down), detectAndCompute vs detect + compute, the FLANN parameters and Lowe's ratio
filter, the Hamming cross-check branch and its sort, the min-matches rule, the arguments handed to
estimateAffinePartial2D / findHomography, the shapes that come back, the error cases -- and that `estimator='auto'`
picks this path when `cv2` is importable, including cv2.resize(INTER_AREA) for the sub-sampling.

The stand-in's "images" carry their keypoints: a pixel of value 200 + k is keypoint k.  Detectors find those pixels,
descriptors encode k (a one-hot float vector for SIFT, a bit pattern for the binary descriptors), the matchers are brute
force, the model fits are plain least squares (similarity / homography with h22 = 1) -- `oracle/cv2_standin.py`, shared
with `oracle/ref_import.load_align_module`, which runs the REFERENCE's own align.py over the same stand-in
(tests/test_align_golden.py holds the comparison)."""
import sys
import types

import numpy as np
import pytest

from oracle.cv2_standin import make_cv2
from shinestacker_amd import align as A
from shinestacker_amd.defaults import constants as c
from shinestacker_amd.errors import InvalidOptionError


@pytest.fixture
def cv2log(monkeypatch):
    log = []
    monkeypatch.setitem(sys.modules, "cv2", make_cv2(log))
    return log


def scene(M, n=40, h=240, w=320, dtype=np.uint8, seed=3):
    """(moving, reference): keypoint k sits at p_k (even coordinates) in the moving image and at round(M p_k) in the
    reference image; returns the exact correspondences too."""
    rng = np.random.default_rng(seed)
    mov, ref = np.zeros((h, w, 3), dtype), np.zeros((h, w, 3), dtype)
    pts = set()
    while len(pts) < n:
        pts.add((2 * int(rng.integers(10, w // 2 - 10)), 2 * int(rng.integers(10, h // 2 - 10))))
    src, dst = [], []
    scale = 257 if dtype == np.uint16 else 1
    for k, (x, y) in enumerate(sorted(pts)):
        v = np.array(M, float) @ [x, y, 1.0]
        if len(v) == 3:
            v = v[:2] / v[2]
        u, t = 2 * int(round(v[0] / 2)), 2 * int(round(v[1] / 2))
        if not (0 <= u < w and 0 <= t < h) or ref[t, u, 0]:
            continue
        mov[y, x] = (200 + len(src)) * scale
        ref[t, u] = (200 + len(src)) * scale
        src.append((x, y))
        dst.append((u, t))
    return mov, ref, np.array(src, float), np.array(dst, float)


FC = {'detector': 'SIFT', 'descriptor': 'SIFT'}
MC = dict(A._DEFAULT_MATCHING_CONFIG)
AC = dict(A._DEFAULT_ALIGNMENT_CONFIG)
M_TRUE = [[1.0, 0.0, 6.0], [0.0, 1.0, -4.0]]


def test_default_recipe_sift_flann_ratio_affine(cv2log):
    mov, ref, src, dst = scene(M_TRUE)
    n, m = A.opencv_estimator(mov, ref, FC, MC, AC)
    assert n == len(src) and m.shape == (2, 3) and np.allclose(m, M_TRUE, atol=1e-9)
    kinds = [e[0] for e in cv2log]
    # one factory call, detectAndCompute on both gray images, no separate descriptor object
    assert cv2log.count(("create", "SIFT")) == 1 and kinds.count("detectAndCompute") == 2 and "compute" not in kinds[:2]
    assert kinds.count("cvtColor") == 2
    assert ("flann", {'algorithm': 2, 'trees': 5}, {'checks': 50}) in cv2log
    fit = [e for e in cv2log if e[0] == "estimateAffinePartial2D"][0]
    assert fit[1] == (len(src), 1, 2) and fit[2] == np.float32                 # np.float32(...).reshape(-1, 1, 2)
    assert fit[3:] == (8, 3.0, 99.9 / 100.0, 100)                             # RANSAC, threshold, confidence / 100, refine


def test_ratio_filter_drops_ambiguous_matches(cv2log):
    """Two reference keypoints with the SAME descriptor as a moving one: nearest and second nearest are equally far, so
    Lowe's test `m.distance < 0.75 n.distance` must drop that match (align.py:59-60)."""
    mov, ref, src, _ = scene(M_TRUE, n=20)
    ys, xs = np.nonzero(ref[..., 0] == 200 + 5)
    ref[ys[0] + 30 if ys[0] + 30 < ref.shape[0] else ys[0] - 30, xs[0]] = 200 + 5   # a twin of keypoint 5
    n, _ = A.opencv_estimator(mov, ref, FC, MC, AC)
    assert n == len(src) - 1
    n_loose, _ = A.opencv_estimator(mov, ref, FC, {**MC, 'threshold': 1.01}, AC)
    assert n_loose == len(src) - 1      # 0 < 1.01 * 0 is still false: an exact twin never passes a strict '<'


@pytest.mark.parametrize("det,des,creates,calls", [
    ("ORB", "ORB", ["ORB", "ORB"], ["detect", "compute"]),
    ("SURF", "BRISK", ["FAST", "BRISK"], ["detect", "compute"]),
    ("AKAZE", "AKAZE", ["AKAZE"], ["detectAndCompute"]),
    ("BRISK", "BRISK", ["BRISK"], ["detectAndCompute"]),
    ("ORB", "BRISK", ["ORB", "BRISK"], ["detect", "compute"]),
])
def test_binary_descriptors_hamming_cross_check(cv2log, det, des, creates, calls):
    mov, ref, src, _ = scene(M_TRUE, n=30)
    n, m = A.opencv_estimator(mov, ref, {'detector': det, 'descriptor': des}, {**MC, 'match_method': 'NORM_HAMMING'}, AC)
    assert n == len(src) and np.allclose(m, M_TRUE, atol=1e-9)
    assert [e[1] for e in cv2log if e[0] == "create"] == creates
    assert all(any(e[0] == k for e in cv2log) for k in calls)
    assert ("bf", 6, True) in cv2log and not any(e[0] == "flann" for e in cv2log)


def test_hamming_matches_are_sorted_by_distance(cv2log, monkeypatch):
    import cv2
    seen = {}
    fit0 = cv2.estimateAffinePartial2D

    def spy(src, dst, **kw):
        seen["src"] = src.copy()
        return fit0(src, dst, **kw)
    monkeypatch.setattr(cv2, "estimateAffinePartial2D", spy)
    mov, ref, src, _ = scene(M_TRUE, n=12)
    A.opencv_estimator(mov, ref, {'detector': 'ORB', 'descriptor': 'ORB'}, {**MC, 'match_method': 'NORM_HAMMING'}, AC)
    # all distances are 0 here and sorted() is stable: the stand-in returns the matches reversed, the estimator must keep
    # exactly that (sorted by distance, ties in matcher order) -- i.e. it sorts, it does not re-derive the order
    got = seen["src"].reshape(-1, 2)
    assert got.shape == (len(src), 2) and {tuple(p) for p in got} == {tuple(p) for p in src}


def test_homography_arguments_and_shape(cv2log):
    H = np.array([[1.0, 0.02, 4.0], [-0.01, 1.0, 2.0], [1e-5, 0.0, 1.0]])
    mov, ref, src, dst = scene(H, n=40)
    n, m = A.opencv_estimator(mov, ref, FC, MC, {**AC, 'transform': c.ALIGN_HOMOGRAPHY, 'align_method': 'LMEDS',
                                                 'max_iters': 1234})
    assert n == len(src) and m.shape == (3, 3)
    fit = [e for e in cv2log if e[0] == "findHomography"][0]
    assert fit[3:] == (4, 3.0, 1234) and fit[1] == (len(src), 1, 2)
    proj = (m @ np.c_[src, np.ones(len(src))].T).T
    assert np.abs(proj[:, :2] / proj[:, 2:] - dst).max() < 4.0           # an unnormalised DLT over positions rounded to the even grid


def test_too_few_matches_returns_no_transform(cv2log):
    mov, ref, src, _ = scene(M_TRUE, n=2)
    assert A.opencv_estimator(mov, ref, FC, MC, AC) == (len(src), None)
    mov, ref, src, _ = scene(M_TRUE, n=3)
    n, m = A.opencv_estimator(mov, ref, FC, MC, {**AC, 'transform': c.ALIGN_HOMOGRAPHY})
    assert (n, m) == (3, None)                                            # a homography needs four (align.py:165)
    n, m = A.opencv_estimator(mov, ref, FC, MC, AC)
    assert n == 3 and m is not None                                       # a similarity three


def test_uint16_frames_are_shifted_to_8_bit_first(cv2log):
    mov, ref, src, _ = scene(M_TRUE, n=25, dtype=np.uint16)
    n, m = A.opencv_estimator(mov, ref, FC, MC, AC)                       # utils.py:37-43: (img >> 8).astype(uint8)
    assert n == len(src) and np.allclose(m, M_TRUE, atol=1e-9)


def test_option_errors(cv2log):
    mov, ref, _, _ = scene(M_TRUE, n=10)
    with pytest.raises(InvalidOptionError):
        A.opencv_estimator(mov, ref, FC, {**MC, 'match_method': 'BRUTE'}, AC)
    with pytest.raises(InvalidOptionError):
        A.opencv_estimator(mov, ref, FC, MC, {**AC, 'align_method': 'MAGSAC'})
    with pytest.raises(ValueError, match="SIFT requires matching method KNN"):
        A.opencv_estimator(mov, ref, FC, {**MC, 'match_method': 'NORM_HAMMING'}, AC)


def test_auto_estimator_is_the_opencv_recipe_when_cv2_is_importable(cv2log):
    """`AlignFrames()` / `align_images()` defaults on a box WITH OpenCV: estimator 'auto' -> opencv_estimator, sub-sample 2
    through cv2.resize(INTER_AREA) (fast_subsampling False is the reference's default), translation scaled back and the
    matrix stored as float32 (align.py:212-227)."""
    assert A.have_opencv() and A.resolve_estimator("auto") is A.opencv_estimator
    mov, ref, src, _ = scene(M_TRUE, n=40)
    seen = {}

    def apply(img, m, cfg):
        seen["m"] = np.array(m)
        return img
    n, m, warp = A.align_images(ref, mov, alignment_config={'min_good_matches': 10}, apply_fn=apply)
    assert n == len(src) and warp is mov
    assert ("resize", 0.5, 0.5, 3) in cv2log
    assert m.dtype == np.float32 and np.allclose(m, M_TRUE, atol=1e-6) and np.array_equal(seen["m"], m)


def test_without_cv2_the_recipe_says_what_to_do(monkeypatch):
    monkeypatch.setitem(sys.modules, "cv2", None)     # import cv2 -> ImportError
    assert not A.have_opencv()
    with pytest.raises(RuntimeError, match="needs OpenCV"):
        A.opencv_estimator(np.zeros((32, 32, 3), np.uint8), np.zeros((32, 32, 3), np.uint8), FC, MC, AC)
