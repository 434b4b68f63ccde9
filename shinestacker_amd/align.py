"""Frame alignment: `align_images` and the `AlignFrames` sub-action.

Mirrors the reference's `algorithms/align.py:154-353` at the boundary (same signature, same
configuration dictionaries and defaults, same return value `(n_good_matches, M, img_warp)`,
same errors), with the work split the way SURVEY.md 8(a) A1-A7 prescribes:

* **estimate** (A1-A4: sub-sample, feature detection, matching, RANSAC) is irregular CPU work
  that the reference delegates entirely to OpenCV; it stays on the host behind a small
  `estimator` callable.  The default estimator is the reference's recipe on `cv2` when OpenCV is
  importable and raises a clear error otherwise -- no estimator is bundled, none is faked.
* **apply** (A5-A6: `cv2.warpAffine`, the warped all-ones mask, the blurred-border composite,
  align.py:238-251) runs on the MI355X through `mi_warp_affine` (C ABI), bit-identical to
  `oracle/align_oracle.c`.

Only the default `ALIGN_RIGID` transform has a GPU apply path; `ALIGN_HOMOGRAPHY` raises
`InvalidOptionError`.
"""
import logging

import numpy as np

from . import _lib
from .actions import SubAction
from .defaults import constants
from .errors import AlignmentError, InvalidOptionError
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
    """utils.py:79-86.  `fast`: strided view; otherwise the integer-factor area average that
    cv2.resize(INTER_AREA) computes (mean of the s x s block, rounded half up) [from memory]."""
    if fast:
        return img[::subsample, ::subsample]
    h, w = img.shape[0] // subsample * subsample, img.shape[1] // subsample * subsample
    blk = img[:h, :w].reshape(h // subsample, subsample, w // subsample, subsample, -1).astype(np.uint32)
    s = blk.sum(axis=(1, 3))
    area = subsample * subsample
    return ((s + area // 2) // area).astype(img.dtype).reshape(h // subsample, w // subsample, *img.shape[2:])


def opencv_estimator(img_0_sub, img_1_sub, feature_config, matching_config, alignment_config):
    """The reference's estimator (align.py:90-151, :186-199) on OpenCV: returns
    (n_good_matches, M or None).  M maps img_0 (moving) onto img_1 (reference)."""
    try:
        import cv2
    except ImportError as e:  # pragma: no cover - OpenCV is not part of this image
        raise RuntimeError(
            "align_images: the transform estimator needs OpenCV (cv2), which is not installed; "
            "pass estimator=callable(img_0_sub, img_1_sub, feature_cfg, matching_cfg, alignment_cfg) "
            "-> (n_good_matches, M)") from e
    def gray8(im):  # pragma: no cover
        im = (im >> 8).astype('uint8') if im.dtype == np.uint16 else im
        return cv2.cvtColor(im, cv2.COLOR_BGR2GRAY) if im.ndim == 3 else im
    det = cv2.SIFT_create()  # pragma: no cover
    kp0, d0 = det.detectAndCompute(gray8(img_0_sub), None)  # pragma: no cover
    kp1, d1 = det.detectAndCompute(gray8(img_1_sub), None)  # pragma: no cover
    flann = cv2.FlannBasedMatcher({'algorithm': matching_config['flann_idx_kdtree'],  # pragma: no cover
                                   'trees': matching_config['flann_trees']},
                                  {'checks': matching_config['flann_checks']})
    good = [m for m, n in flann.knnMatch(d0, d1, k=2)  # pragma: no cover
            if m.distance < matching_config['threshold'] * n.distance]
    min_matches = 3  # pragma: no cover
    if len(good) < min_matches:  # pragma: no cover
        return len(good), None
    src = np.float32([kp0[m.queryIdx].pt for m in good]).reshape(-1, 1, 2)  # pragma: no cover
    dst = np.float32([kp1[m.trainIdx].pt for m in good]).reshape(-1, 1, 2)  # pragma: no cover
    m, _ = cv2.estimateAffinePartial2D(  # pragma: no cover
        src, dst, method=cv2.RANSAC, ransacReprojThreshold=alignment_config['rans_threshold'],
        confidence=alignment_config['align_confidence'] / 100.0,
        refineIters=alignment_config['refine_iters'])
    return len(good), m  # pragma: no cover


def ecc_estimator(min_correlation=0.5, max_iters=60, device=0):
    """Estimator for `align_images` that runs on the GPU (mi_ecc_similarity): ECC maximisation of
    a 4-DoF similarity, coarse to fine.  It needs no feature matches; to satisfy the protocol it
    reports 1000 "good matches" when the final correlation coefficient reaches `min_correlation`
    and 0 otherwise (which makes align_images / AlignFrames raise AlignmentError as usual)."""
    def estimate(img_0_sub, img_1_sub, _feature_config, _matching_config, _alignment_config):
        m, cc, _iters = _lib.ecc_similarity(img_1_sub, img_0_sub, max_iters=max_iters, device=device)
        return (1000, m) if cc >= min_correlation else (0, None)
    return estimate


def apply_transform(img, m, alignment_config, device=0):
    """align.py:238-251 on the GPU: warp + mask + blurred-border composite."""
    mode = _BORDER_CODE[alignment_config['border_mode']]
    return _lib.warp_affine(img, m, border_mode=mode, border_value=alignment_config['border_value'],
                            blur_ksize=21, blur_sigma=alignment_config['border_blur'], device=device)


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
    if transform == constants.ALIGN_HOMOGRAPHY:
        raise InvalidOptionError("transform", transform,
                                 "the MI355X apply path implements ALIGN_RIGID only")
    if transform != constants.ALIGN_RIGID:
        raise InvalidOptionError("transform", transform)
    min_matches = 3
    validate_image(img_0, *get_img_metadata(img_1))
    if callbacks and 'message' in callbacks:
        callbacks['message']()
    estimator = estimator or opencv_estimator
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
        if subsample > 1:
            # translation found on the sub-sampled pair, applied at full resolution; the
            # reference stores the rescaled matrix as float32 (align.py:217-223)
            full = np.empty((2, 3), dtype=np.float32)
            full[:2, :2] = m[:2, :2]
            full[:, 2] = m[:, 2] * subsample
            m = full
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
