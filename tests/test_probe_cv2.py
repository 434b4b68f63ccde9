"""CPU: oracle/probe_cv2.py -- the script that pins the restated cv2 primitives on a machine WITH OpenCV -- runs end to
end.  There is no OpenCV here, so `cv2` is a stand-in built from the oracle's own functions: every comparison then holds
trivially (or is skipped where the stand-in has no such function), which checks the script's plumbing -- argument shapes,
file names, report keys -- not OpenCV.  Nothing is written into tests/golden/."""
import json
import sys

import numpy as np


def _fake_cv2(orc):
    from oracle import ref_import
    cv2 = ref_import.make_cv2_shim(use_fma=True)
    cv2.__version__ = "stand-in"
    cv2.getBuildInformation = lambda: "stand-in built from oracle/\n"
    blur0 = cv2.GaussianBlur

    def warpAffine(img, M, dsize, borderMode=None, borderValue=0):
        mode = 1 if borderMode == cv2.BORDER_REPLICATE else 0
        return orc.warp_affine(img, np.asarray(M, np.float64), border_mode=mode, border_value=(borderValue,) * 4)

    def warpPerspective(img, M, dsize, borderMode=None, borderValue=0):
        return orc.warp_perspective(img, M, border_mode=1)

    def GaussianBlur(img, ksize, sigma=0, sigmaX=None):
        if img.ndim == 3 and img.dtype in (np.uint8, np.uint16):   # the composite's blur: the restated fixed-point path
            return orc.gaussian_blur_fixed(img, ksize[0], sigma if sigmaX is None else sigmaX)
        return blur0(img, ksize, sigma if sigmaX is None else sigmaX)

    cv2.warpAffine, cv2.warpPerspective, cv2.GaussianBlur = warpAffine, warpPerspective, GaussianBlur
    return cv2


def test_probe_script_runs_with_a_stand_in_cv2(oracle, tmp_path, monkeypatch, capsys):
    from oracle import probe_cv2
    monkeypatch.setitem(sys.modules, "cv2", _fake_cv2(oracle))
    import os
    import shutil
    shutil.copy(os.path.join(probe_cv2.GOLDEN, "g1_u8.npz"), tmp_path / "g1_u8.npz")
    monkeypatch.setattr(probe_cv2, "GOLDEN", str(tmp_path))
    assert probe_cv2.main() == 0
    report = json.loads((tmp_path / "cv2_probe_report.json").read_text())
    assert report["filter2D"]["matches_use_fma"] == [1] or 1 in report["filter2D"]["matches_use_fma"]
    assert report["cvtColor_u8"]["matches"] and report["warpAffine"]["matches"] and report["warpPerspective"]["matches"]
    assert report["warp+blur_composite"]["matches"] and report["GaussianBlur_21_fixed"] == {"u8": True, "u16": True}
    assert all(report["resize_INTER_AREA"].values()) and len(report["resize_INTER_AREA"]) == 12
    assert all(report["cvtColor_HSV_HLS_u8"].values()) and len(report["cvtColor_HSV_HLS_u8"]) == 4
    dm = report["depth_map_primitives"]
    assert dm["GaussianBlur_5"] and dm["Laplacian_64F_5"] and dm["pyrDown"] and dm["pyrUp"] and dm["bilateralFilter_15"]["equal"]
    assert capsys.readouterr().out.strip().endswith("}")
