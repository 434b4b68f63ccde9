"""`align_images` against recordings of the REFERENCE's own `align_images` (algorithms/align.py:154-252).

tests/golden/align.npz was written by oracle/gen_golden.py::align_case, which imports the reference's align.py in the
build container (oracle/ref_import.load_align_module) and runs it on twelve scenes: rigid / homography x sub-sample
1 / 2 / 4 (fast and INTER_AREA) x REPLICATE_BLUR / REPLICATE / CONSTANT x 8 / 16 bit, the retry without sub-sampling, too
few matches for either transform, the ORB + Hamming + LMEDS recipe.  The estimator calls ran on the stand-in of
oracle/cv2_standin.py, the numeric cv2 calls on oracle.py's raw primitives; everything else -- the loop, the rules,
the rescale, the casts, the mask warp and composite -- was the reference's code.  Frozen per scene:
(n_good_matches, M, img_warp), the callback sequence, the stand-in calls and the arguments of the model fit.

Here `shinestacker_amd.align_images` runs over the SAME stand-in (estimator='opencv') and must reproduce all of it:
on the CPU with the oracle's apply restatement (host logic: sub-sampling, retry, min-matches, rescale, dtype of M,
callbacks), on the GPU with the device apply (mi_warp_affine / mi_warp_perspective)."""
import json
import os
import sys

import numpy as np
import pytest

from oracle import cv2_standin as cs
from oracle import oracle as orc
from shinestacker_amd import align as A

GOLD = os.path.join(os.path.dirname(__file__), "golden", "align.npz")
Z = np.load(GOLD)
META = json.loads(str(Z["meta"]))
IDS = [e["name"] for e in META]
MODES = {"BORDER_CONSTANT": orc.BORDER_CONSTANT, "BORDER_REPLICATE": orc.BORDER_REPLICATE,
         "BORDER_REPLICATE_BLUR": orc.BORDER_REPLICATE_BLUR}


def oracle_apply(img, m, cfg):
    fn = orc.warp_perspective if np.asarray(m).shape == (3, 3) else orc.warp_affine
    return fn(img, m, MODES[cfg["border_mode"]], cfg["border_value"], 21, cfg["border_blur"])


def standin(monkeypatch):
    log = []
    monkeypatch.setitem(sys.modules, "cv2", cs.make_cv2(log, resize=lambda img, s: orc.resize_area_int(img, s),
                                                        gray=orc.bgr2gray_int))
    return log


def run_ours(e, log, apply_fn=None):
    name = e["name"]
    trace = []
    callbacks = {k: (lambda *a, _k=k: trace.append([_k] + [str(x) for x in a]))
                 for k in ("message", "matches_message", "align_message", "ecc_message", "blur_message", "warning",
                           "save_plot")}
    n, m, warp = A.align_images(Z[f"{name}_ref"], Z[f"{name}_mov"].copy(), feature_config=e["feature_config"],
                                matching_config=e["matching_config"], alignment_config=e["alignment_config"],
                                callbacks=callbacks, estimator="opencv", apply_fn=apply_fn)
    calls = [c[0] for c in log]
    assert n == e["n_good_matches"]
    assert trace == e["callbacks"]                      # same messages, same order, same texts (the retry warning)
    assert [c for c in calls if c in ("detectAndCompute", "detect", "flann", "bf", "estimateAffinePartial2D",
                                      "findHomography")] == e["standin_calls"]
    assert [[str(x) for x in c[1:]] for c in log if c[0] in ("estimateAffinePartial2D", "findHomography")] == e["fit_args"]
    if not e["aligned"]:
        assert m is None and warp is None
        return
    want_m = Z[f"{name}_m"]
    assert m.dtype == want_m.dtype and str(m.dtype) == e["m_dtype"] and np.array_equal(m, want_m)
    want = Z[f"{name}_warp"]
    assert warp.dtype == want.dtype and warp.shape == want.shape
    assert np.array_equal(warp, want), f"{(warp != want).sum()} values differ from the reference's align_images"


@pytest.mark.parametrize("e", META, ids=IDS)
def test_oracle_apply_equals_the_reference_recording(e):
    """oracle.warp_affine / warp_perspective (what every test in test_gpu_align.py compares the kernels with) fed the
    recorded M reproduce the reference's img_warp: its mask rule and composite are pinned."""
    if not e["aligned"]:
        pytest.skip("no transform in this scene")
    cfg = {**A._DEFAULT_ALIGNMENT_CONFIG, **(e["alignment_config"] or {})}
    name = e["name"]
    assert np.array_equal(oracle_apply(Z[f"{name}_mov"], Z[f"{name}_m"], cfg), Z[f"{name}_warp"])
    assert e["out_of_frame_pixels"] > 500               # the border rules are exercised, not a no-op warp


@pytest.mark.parametrize("e", META, ids=IDS)
def test_host_logic_equals_the_reference_recording(e, monkeypatch):
    run_ours(e, standin(monkeypatch), apply_fn=oracle_apply)


def test_area_subsampling_without_opencv_equals_the_shim_the_reference_ran_on(monkeypatch):
    """The recordings' INTER_AREA frames came from oracle.resize_area_int; the product's own restatement (used when cv2 is
    absent) must give the same frames, ragged sizes included."""
    monkeypatch.setitem(sys.modules, "cv2", None)
    for name, s in (("rigid_sub2_area_blur_u8", 2), ("rigid_sub2_area_blur_u16", 2), ("rigid_sub4_area_ragged_blur_u8", 4),
                    ("rigid_sub2_fast_replicate_u8", 2), ("rigid_sub4_area_ragged_blur_u8", 3)):
        img = Z[f"{name}_mov"]
        assert np.array_equal(A.img_subsample(img, s, False), orc.resize_area_int(img, s))


@pytest.mark.gpu
@pytest.mark.parametrize("e", META, ids=IDS)
def test_device_align_images_equals_the_reference_recording(e, monkeypatch):
    """The product path end to end: estimator recipe on the stand-in, apply on the MI355X."""
    from shinestacker_amd import _lib
    _lib.require_device()
    run_ours(e, standin(monkeypatch))
