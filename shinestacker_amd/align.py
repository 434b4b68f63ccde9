"""Frame alignment: `align_images` and the `AlignFrames` sub-action.

Mirrors the reference's `algorithms/align.py:154-353` at the boundary (same signature, same
configuration dictionaries and defaults, same return value `(n_good_matches, M, img_warp)`,
same errors), with the work split the way SURVEY.md 8(a) A1-A7 prescribes:

* **estimate** (A1-A4: sub-sample, feature detection, matching, RANSAC) is irregular CPU work
  that the reference delegates entirely to OpenCV; it sits behind a small `estimator` callable.
  The default (`estimator="auto"`) is the reference's own recipe on `cv2` when OpenCV is importable
  (`opencv_estimator`: every detector / descriptor / matcher / transform option of align.py:48-151),
  and the GPU ECC estimator (`ecc_estimator`, mi_ecc_similarity) otherwise -- so `AlignFrames()` works
  out of the box on a box without OpenCV, as the reference's does with it.
* **apply** (A5-A6: `cv2.warpAffine` / `cv2.warpPerspective`, the warped all-ones mask, the
  blurred-border composite, align.py:231-251) runs on the MI355X through `mi_warp_affine` /
  `mi_warp_perspective` (C ABI), bit-identical to `oracle/align_oracle.c`.
"""
import logging

import numpy as np

from . import _lib
from .actions import SubAction
from .defaults import constants
from .errors import AlignmentError, DeviceError, InvalidOptionError
from .imageio import get_img_metadata, validate_image

_DEFAULT_FEATURE_CONFIG = {'detector': 'SIFT', 'descriptor': 'SIFT'}
_DEFAULT_MATCHING_CONFIG = {'match_method': 'KNN', 'flann_idx_kdtree': 2, 'flann_trees': 5,
                            'flann_checks': 50, 'threshold': 0.75}
_DEFAULT_ALIGNMENT_CONFIG = {
    'transform': constants.DEFAULT_TRANSFORM, 'align_method': 'RANSAC', 'rans_threshold': 3.0,
    'refine_iters': 100, 'align_confidence': 99.9, 'max_iters': 2000,
    'border_mode': constants.DEFAULT_BORDER_MODE, 'border_value': list(constants.DEFAULT_BORDER_VALUE),
    'border_blur': constants.DEFAULT_BORDER_BLUR, 'subsample': constants.DEFAULT_ALIGN_SUBSAMPLE,
    'fast_subsampling': False, 'min_good_matches': 100}

_BORDER_CODE = {constants.BORDER_CONSTANT: _lib.BORDER_CONSTANT,
                constants.BORDER_REPLICATE: _lib.BORDER_REPLICATE,
                constants.BORDER_REPLICATE_BLUR: _lib.BORDER_REPLICATE_BLUR}


def img_subsample(img, subsample, fast=True):
    """utils.py:79-86.  `fast`: strided view.  Otherwise cv2.resize(img, (0, 0), fx=1/s, fy=1/s, INTER_AREA): OpenCV itself
    when it is importable, else its integer-factor area path restated [from memory, parity unpinned]: output size
    round-half-even(dim / s); whole s x s blocks -> (sum + 2) >> 2 for s == 2, else the float32 product sum * (1 / s^2)
    rounded half to even; the partial blocks of a last row / column (sizes not divisible by s) -> float32 sum / count
    over the pixels that exist, rounded half to even."""
    if fast:
        return img[::subsample, ::subsample]
    s = int(subsample)
    if s == 1:
        return img
    if have_opencv():
        import cv2
        return cv2.resize(img, (0, 0), fx=1 / s, fy=1 / s, interpolation=cv2.INTER_AREA)
    h, w = img.shape[:2]
    dh, dw = int(np.rint(h * (1.0 / s))), int(np.rint(w * (1.0 / s)))
    ph, pw = dh * s, dw * s
    pad = np.zeros((ph, pw) + img.shape[2:], np.uint32)
    cnt = np.zeros((ph, pw), np.uint32)
    hh, ww = min(h, ph), min(w, pw)
    pad[:hh, :ww] = img[:hh, :ww]
    cnt[:hh, :ww] = 1
    tail = img.shape[2:]
    sums = pad.reshape(dh, s, dw, s, -1).sum(axis=(1, 3))
    counts = cnt.reshape(dh, s, dw, s).sum(axis=(1, 3))[..., None]
    full = counts == s * s
    if s == 2:
        whole = (sums + 2) >> 2
    else:
        whole = np.rint(sums.astype(np.float32) * np.float32(1.0 / (s * s))).astype(np.int64)
    part = np.rint(sums.astype(np.float32) / np.maximum(counts, 1).astype(np.float32)).astype(np.int64)
    out = np.where(full, whole, np.where(counts > 0, part, 0))
    hi = np.iinfo(img.dtype).max if np.issubdtype(img.dtype, np.integer) else None
    if hi is not None:
        out = np.clip(out, 0, hi)
    return out.astype(img.dtype).reshape((dh, dw) + tail)


def validate_align_config(detector, descriptor, match_method):
    """align.py:71-87: the detector x descriptor x matcher combinations the reference refuses, with its messages."""
    c = constants
    if descriptor == c.DESCRIPTOR_SIFT and match_method == c.MATCHING_NORM_HAMMING:
        raise ValueError("Descriptor SIFT requires matching method KNN")
    if detector == c.DETECTOR_ORB and descriptor == c.DESCRIPTOR_AKAZE and match_method == c.MATCHING_NORM_HAMMING:
        raise ValueError("Detector ORB and descriptor AKAZE require matching method KNN")
    if detector == c.DETECTOR_BRISK and descriptor == c.DESCRIPTOR_AKAZE:
        raise ValueError("Detector BRISK is incompatible with descriptor AKAZE")
    if detector == c.DETECTOR_SURF and descriptor == c.DESCRIPTOR_AKAZE:
        raise ValueError("Detector SURF is incompatible with descriptor AKAZE")
    if detector == c.DETECTOR_SIFT and descriptor != c.DESCRIPTOR_SIFT:
        raise ValueError("Detector SIFT requires descriptor SIFT")
    if detector in c.NOKNN_METHODS['detectors'] and descriptor in c.NOKNN_METHODS['descriptors'] and \
            match_method != c.MATCHING_NORM_HAMMING:
        raise ValueError(f"Detector {detector} and descriptor {descriptor} require matching method Hamming distance")


def _cv2():
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError(
            "opencv_estimator needs OpenCV (cv2), which is not installed; use estimator='auto' / ecc_estimator(), or "
            "pass estimator=callable(img_0_sub, img_1_sub, feature_cfg, matching_cfg, alignment_cfg) "
            "-> (n_good_matches, M)") from e
    return cv2


def get_good_matches(des_0, des_1, matching_config=None):
    """align.py:48-68: FLANN k-nearest-neighbour matches through Lowe's ratio test, or Hamming brute force with cross-check,
    sorted by distance."""
    cv2 = _cv2()
    matching_config = {**_DEFAULT_MATCHING_CONFIG, **(matching_config or {})}
    match_method = matching_config['match_method']
    c = constants
    if match_method == c.MATCHING_KNN:
        flann = cv2.FlannBasedMatcher({'algorithm': matching_config['flann_idx_kdtree'], 'trees': matching_config['flann_trees']},
                                      {'checks': matching_config['flann_checks']})
        return [m for m, n in flann.knnMatch(des_0, des_1, k=2) if m.distance < matching_config['threshold'] * n.distance]
    if match_method == c.MATCHING_NORM_HAMMING:
        return sorted(cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(des_0, des_1), key=lambda x: x.distance)
    raise InvalidOptionError('match_method', match_method, f". Valid options are: {c.MATCHING_KNN}, {c.MATCHING_NORM_HAMMING}")


def detect_and_compute(img_0, img_1, feature_config=None, matching_config=None):
    """align.py:90-122: key points and descriptors of both images (8-bit gray, utils.py:37-43) and their good matches:
    returns (kp_0, kp_1, good_matches)."""
    feature_config = {**_DEFAULT_FEATURE_CONFIG, **(feature_config or {})}
    matching_config = {**_DEFAULT_MATCHING_CONFIG, **(matching_config or {})}
    detector_name, descriptor_name = feature_config['detector'], feature_config['descriptor']
    validate_align_config(detector_name, descriptor_name, matching_config['match_method'])
    cv2 = _cv2()
    c = constants

    def gray8(im):  # (utils.py:37-43 img_bw_8bit)
        im = (im >> 8).astype('uint8') if im.dtype == np.uint16 else im
        return cv2.cvtColor(im, cv2.COLOR_BGR2GRAY) if im.ndim == 3 else im
    det_map = {c.DETECTOR_SIFT: cv2.SIFT_create, c.DETECTOR_ORB: cv2.ORB_create,
               c.DETECTOR_SURF: cv2.FastFeatureDetector_create, c.DETECTOR_AKAZE: cv2.AKAZE_create,
               c.DETECTOR_BRISK: cv2.BRISK_create}
    des_map = {c.DESCRIPTOR_SIFT: cv2.SIFT_create, c.DESCRIPTOR_ORB: cv2.ORB_create,
               c.DESCRIPTOR_AKAZE: cv2.AKAZE_create, c.DESCRIPTOR_BRISK: cv2.BRISK_create}
    g0, g1 = gray8(img_0), gray8(img_1)
    det = det_map[detector_name]()
    if detector_name == descriptor_name and detector_name in (c.DETECTOR_SIFT, c.DETECTOR_AKAZE, c.DETECTOR_BRISK):
        kp0, d0 = det.detectAndCompute(g0, None)
        kp1, d1 = det.detectAndCompute(g1, None)
    else:
        des = des_map[descriptor_name]()
        kp0, d0 = des.compute(g0, det.detect(g0, None))
        kp1, d1 = des.compute(g1, det.detect(g1, None))
    return kp0, kp1, get_good_matches(d0, d1, matching_config)


def find_transform(src_pts, dst_pts, transform=constants.DEFAULT_TRANSFORM, method='RANSAC', rans_threshold=3.0,
                   max_iters=2000, align_confidence=99.9, refine_iters=100):
    """align.py:125-151: cv2.findHomography / cv2.estimateAffinePartial2D with the reference's arguments; returns what
    they return, (matrix, inlier mask)."""
    cv2 = _cv2()
    c = constants
    cv2_method = {'RANSAC': cv2.RANSAC, 'LMEDS': cv2.LMEDS}.get(method)
    if cv2_method is None:
        raise InvalidOptionError('align_method', method, f". Valid options are: {c.ALIGN_RANSAC}, {c.ALIGN_LMEDS}")
    if transform == c.ALIGN_HOMOGRAPHY:
        return cv2.findHomography(src_pts, dst_pts, method=cv2_method, ransacReprojThreshold=rans_threshold, maxIters=max_iters)
    if transform == c.ALIGN_RIGID:
        return cv2.estimateAffinePartial2D(src_pts, dst_pts, method=cv2_method, ransacReprojThreshold=rans_threshold,
                                           confidence=align_confidence / 100.0, refineIters=refine_iters)
    raise InvalidOptionError("transform", transform)


def opencv_estimator(img_0_sub, img_1_sub, feature_config, matching_config, alignment_config):
    """The reference's estimator (align.py:48-151, :186-199) on OpenCV -- `detect_and_compute`, `get_good_matches`,
    `find_transform` above, in the reference's order: returns (n_good_matches, M or None); M maps img_0 (moving) onto
    img_1 (reference): 2x3 for ALIGN_RIGID, 3x3 for ALIGN_HOMOGRAPHY."""
    kp0, kp1, good = detect_and_compute(img_0_sub, img_1_sub, feature_config, matching_config)
    transform = alignment_config['transform']
    if len(good) < (4 if transform == constants.ALIGN_HOMOGRAPHY else 3):
        return len(good), None
    src = np.float32([kp0[m.queryIdx].pt for m in good]).reshape(-1, 1, 2)
    dst = np.float32([kp1[m.trainIdx].pt for m in good]).reshape(-1, 1, 2)
    m, _ = find_transform(src, dst, transform, alignment_config['align_method'],
                          *(alignment_config[k] for k in ['rans_threshold', 'max_iters', 'align_confidence', 'refine_iters']))
    return len(good), m


def have_opencv():
    try:
        import cv2  # noqa: F401
        return True
    except Exception:  # noqa: BLE001
        return False


_auto_fallback_logged = False


def auto_estimator(device=0):
    """`estimator="auto"`: the reference's own recipe when OpenCV is importable, the GPU ECC estimator otherwise -- a different
    algorithm (the reference matches SIFT features and fits with RANSAC, align.py:90-151), so the swap is logged, once."""
    global _auto_fallback_logged
    if have_opencv():
        return opencv_estimator
    if not _auto_fallback_logged:
        _auto_fallback_logged = True
        logging.getLogger("shinestacker_amd").warning(
            "estimator='auto': OpenCV (cv2) is not importable, the reference's detector / matcher / RANSAC recipe "
            "(align.py:90-151) cannot run; frames are registered with the GPU ECC estimator instead (estimator='ecc')")
    return ecc_estimator(device=device)


def resolve_estimator(estimator, device=0):
    if estimator is None or estimator == "auto":
        return auto_estimator(device)
    if estimator == "opencv":
        return opencv_estimator
    if estimator == "ecc":
        return ecc_estimator(device=device)
    if callable(estimator):
        return estimator
    raise InvalidOptionError("estimator", estimator, ". Valid options are: 'auto', 'opencv', 'ecc' or a callable")


def ecc_estimator(min_correlation=0.5, max_iters=60, device=0, phase_init=False):
    """Estimator for `align_images` that runs on the GPU (mi_ecc_similarity): ECC maximisation of
    a 4-DoF similarity, coarse to fine.  It needs no feature matches; to satisfy the protocol it
    reports 1000 "good matches" when the final correlation coefficient reaches `min_correlation`
    and 0 otherwise (which makes align_images / AlignFrames raise AlignmentError as usual).
    `phase_init`: start every estimate from the translation found by phase correlation (mi_aligner_set_phase_init):
    shifts far beyond the ECC pyramid's capture range (tens of per cent of the frame) are then recovered too."""
    def estimate(img_0_sub, img_1_sub, _feature_config, _matching_config, alignment_config):
        homography = (alignment_config or {}).get('transform') == constants.ALIGN_HOMOGRAPHY
        if phase_init or homography:
            ref, mov = np.ascontiguousarray(img_1_sub), np.ascontiguousarray(img_0_sub)
            # the same contract as _lib.ecc_similarity: the estimator reads H * W * 3 elements of the handle's dtype from
            # each half of the buffer
            if ref.shape != mov.shape or ref.dtype != mov.dtype or ref.ndim != 3 or ref.shape[2] != 3 \
                    or ref.dtype not in _lib.DTYPE_CODE:
                raise ValueError("ecc_estimator expects two H x W x 3 images of the same shape and dtype (uint8 / uint16)")
            al = buf = None
            try:
                al = _lib.Aligner(ref.shape[0], ref.shape[1], ref.dtype, subsample=1, device=device, phase_init=phase_init)
                buf = _lib.DeviceBuffer(2 * ref.nbytes, device)
                buf.upload(ref)
                buf.upload(mov, ref.nbytes)
                al.set_reference(buf.ptr)
                try:
                    if homography:   # the similarity refined to 8 degrees of freedom (cv2.findHomography's role, align.py:138-140)
                        ms, ccs, _ = al.estimate_homography_batch([buf.ptr + ref.nbytes], max_iters=max_iters)
                        m, cc = ms[0], float(ccs[0])
                    else:
                        m, cc, _iters = al.estimate(buf.ptr + ref.nbytes, max_iters=max_iters)
                except DeviceError as e:
                    # the estimator's own documented failure (cc == -2: no overlap / constant image / degenerate transform, in
                    # either motion model) means "no matches"; anything else -- a HIP fault, a bad handle -- is not an alignment result
                    if not str(e).startswith("ECC:"):
                        raise
                    return 0, None
            finally:
                if al is not None:
                    al.close()
                if buf is not None:
                    buf.free()
        else:
            m, cc, _iters = _lib.ecc_similarity(img_1_sub, img_0_sub, max_iters=max_iters, device=device)
        if not cc >= min_correlation:
            return 0, None
        return 1000, m
    return estimate


def apply_transform(img, m, alignment_config, device=0):
    """align.py:231-251 on the GPU: warp (affine for a 2x3, perspective for a 3x3 matrix) + mask + blurred-border
    composite."""
    mode = _BORDER_CODE[alignment_config['border_mode']]
    fn = _lib.warp_perspective if np.asarray(m).shape == (3, 3) else _lib.warp_affine
    return fn(img, m, border_mode=mode, border_value=alignment_config['border_value'],
              blur_ksize=21, blur_sigma=alignment_config['border_blur'], device=device)


def rescale_transform(m, transform, subsample, shape, shape_sub):
    """align.py:212-227: a transform found on images sub-sampled by `subsample`, for the full-size images.
    ALIGN_RIGID: translation times the factor, stored as float32 like the reference does.  ALIGN_HOMOGRAPHY: conjugation
    with the corner-to-corner scalings cv2.getPerspectiveTransform returns for the two image rectangles, i.e.
    diag(w / w_sub, h / h_sub, 1) and its inverse."""
    m = np.asarray(m)
    if transform == constants.ALIGN_HOMOGRAPHY:
        (h, w), (hs, ws) = shape[:2], shape_sub[:2]
        up = np.diag([w / ws, h / hs, 1.0])
        down = np.diag([ws / w, hs / h, 1.0])
        return up @ m @ down
    if transform == constants.ALIGN_RIGID:
        full = np.empty((2, 3), dtype=np.float32)
        full[:2, :2] = m[:2, :2]
        full[:, 2] = m[:, 2] * subsample
        return full
    raise InvalidOptionError("transform", transform)


def align_images(img_1, img_0, feature_config=None, matching_config=None, alignment_config=None,
                 plot_path=None, callbacks=None, estimator=None, apply_fn=None):
    """Align `img_0` (moving) onto `img_1` (reference).  Returns (n_good_matches, M, img_warp);
    `M` and `img_warp` are None when fewer than 3 good matches were found (align.py:154-252)."""
    feature_config = {**_DEFAULT_FEATURE_CONFIG, **(feature_config or {})}
    matching_config = {**_DEFAULT_MATCHING_CONFIG, **(matching_config or {})}
    alignment_config = {**_DEFAULT_ALIGNMENT_CONFIG, **(alignment_config or {})}
    if alignment_config['border_mode'] not in _BORDER_CODE:
        raise InvalidOptionError("border_mode", alignment_config['border_mode'])
    transform = alignment_config['transform']
    if transform not in (constants.ALIGN_RIGID, constants.ALIGN_HOMOGRAPHY):
        raise InvalidOptionError("transform", transform)
    min_matches = 4 if transform == constants.ALIGN_HOMOGRAPHY else 3
    # the reference refuses these combinations inside detect_and_compute (align.py:71-87, :97): same errors here,
    # whichever estimator runs
    validate_align_config(feature_config['detector'], feature_config['descriptor'], matching_config['match_method'])
    validate_image(img_0, *get_img_metadata(img_1))
    if callbacks and 'message' in callbacks:
        callbacks['message']()
    estimator = resolve_estimator(estimator)
    subsample = alignment_config['subsample']
    fast = alignment_config['fast_subsampling']
    while True:
        if subsample > 1:
            img_0_sub, img_1_sub = img_subsample(img_0, subsample, fast), img_subsample(img_1, subsample, fast)
        else:
            img_0_sub, img_1_sub = img_0, img_1
        n_good_matches, m = estimator(img_0_sub, img_1_sub, feature_config, matching_config,
                                      alignment_config)
        if n_good_matches > alignment_config['min_good_matches'] or subsample == 1:
            break
        subsample = 1
        if callbacks and 'warning' in callbacks:
            callbacks['warning'](f"only {n_good_matches} < {alignment_config['min_good_matches']} "
                                 "matches found, retrying without subsampling")
    if callbacks and 'matches_message' in callbacks:
        callbacks['matches_message'](n_good_matches)
    img_warp = None
    if n_good_matches >= min_matches and m is not None:
        m = np.asarray(m)
        if transform == constants.ALIGN_HOMOGRAPHY and m.shape == (2, 3):
            m = np.vstack([m, [0.0, 0.0, 1.0]])
        if subsample > 1:
            m = rescale_transform(m, transform, subsample, img_0.shape, img_0_sub.shape)
        if callbacks and 'align_message' in callbacks:
            callbacks['align_message']()
        blur = alignment_config['border_mode'] == constants.BORDER_REPLICATE_BLUR
        if blur and callbacks and 'blur_message' in callbacks:
            callbacks['blur_message']()
        img_warp = (apply_fn or apply_transform)(img_0, m, alignment_config)
    else:
        m = None
    return n_good_matches, m, img_warp


class AlignFrames(SubAction):
    """Sub-action of CombinedActions (align.py:255-353)."""

    def __init__(self, enabled=True, feature_config=None, matching_config=None, alignment_config=None,
                 estimator=None, **kwargs):
        super().__init__(enabled)
        self.process = None
        self.n_matches = None
        self.estimator = estimator
        self.feature_config = {**_DEFAULT_FEATURE_CONFIG, **(feature_config or {})}
        self.matching_config = {**_DEFAULT_MATCHING_CONFIG, **(matching_config or {})}
        self.alignment_config = {**_DEFAULT_ALIGNMENT_CONFIG, **(alignment_config or {})}
        self.min_matches = 4 if self.alignment_config['transform'] == constants.ALIGN_HOMOGRAPHY else 3
        self.plot_summary = kwargs.get('plot_summary', False)
        self.plot_matches = kwargs.get('plot_matches', False)
        for cfg in (self.feature_config, self.matching_config, self.alignment_config):
            for k in cfg:
                if k in kwargs:
                    cfg[k] = kwargs[k]

    def begin(self, process):
        self.process = process
        # indexed by the GLOBAL frame index: a sharded process counts only its own block (process.counts)
        self.n_matches = np.zeros(len(process.filenames))

    def run_frame(self, idx, ref_idx, img_0):
        if idx == self.process.ref_idx:
            return img_0
        img_ref = self.process.img_ref(ref_idx)
        return self.align_images(idx, img_ref, img_0)

    def sub_msg(self, msg):
        self.process.sub_message_r(msg)

    def align_images(self, idx, img_1, img_0):
        callbacks = {
            'message': lambda: self.sub_msg(': find matches'),
            'matches_message': lambda n: self.sub_msg(f": good matches: {n}"),
            'align_message': lambda: self.sub_msg(': align images'),
            'blur_message': lambda: self.sub_msg(': blur borders'),
            'warning': lambda msg: self.sub_msg(f': {msg}'),
        }
        n_good_matches, _m, img = align_images(
            img_1, img_0, feature_config=self.feature_config, matching_config=self.matching_config,
            alignment_config=self.alignment_config, callbacks=callbacks, estimator=self.estimator)
        self.n_matches[idx] = n_good_matches
        if n_good_matches < self.min_matches:
            self.process.sub_message(f": image not aligned, too few matches found: {n_good_matches}",
                                     level=logging.CRITICAL)
            raise AlignmentError(idx, f"too few matches found: {n_good_matches} < {self.min_matches}")
        return img
