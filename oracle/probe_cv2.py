"""oracle/probe_cv2.py -- pin the cv2 primitives the day OpenCV is importable.   python -m oracle.probe_cv2

TEST INFRASTRUCTURE.  The oracle restates filter2D, cvtColor(BGR2GRAY) on float32 and on integers, cvtColor BGR <-> HSV / HLS
(8-bit), warpAffine, warpPerspective, GaussianBlur, resize(INTER_AREA) and DepthMapStack's Laplacian / pyrDown / pyrUp /
bilateralFilter from OpenCV's published algorithms [from memory] because OpenCV is neither
vendored in the reference (pyproject.toml:26, unpinned) nor installed in the build image: "parity unpinned".
When `import cv2` works, this script runs the REAL primitives on the golden inputs of tests/golden/, reports for
each one whether the restatement matches bit for bit (filter2D: which of use_fma in {1, 0}, or neither), and writes
the real outputs to tests/golden/cv2_<primitive>.npz so that the fact survives in the repository.  Without cv2 it
says so and exits 0 (nothing to pin)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def probe(write=True):
    """-> {"available": False, "error": ...} without OpenCV; else the comparison report ({"available": True, "cv2_version",
    one entry per primitive}).  `write`: also save the real outputs under tests/golden/ (the script form; bench.py's
    cpu_baseline leg passes False: a GPU box's copy of the repository is scratch)."""
    try:
        import cv2
    except Exception as e:  # noqa: BLE001
        return {"available": False, "error": f"{type(e).__name__}: {e}"}
    _savez = np.savez_compressed

    def savez(*a, **k):
        if write:
            _savez(*a, **k)
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    orc.build()
    report = {"available": True, "cv2_version": cv2.__version__, "build": cv2.getBuildInformation().split("\n")[0:1]}
    with np.load(os.path.join(GOLDEN, "g1_u8.npz")) as z:
        frames = z["frames"]
    img = frames[0].astype(np.float32)
    k2d = orc.gen_kernel_2d(0.4)

    # ---- filter2D (pyramid.py:24-25): float32 image, float64 kernel, REFLECT101
    real = np.stack([cv2.filter2D(np.ascontiguousarray(img[:, :, c]), -1, k2d, borderType=cv2.BORDER_REFLECT101)
                     for c in range(3)], axis=-1)
    match = [fma for fma in (1, 0)
             if all(np.array_equal(orc.filter2D(np.ascontiguousarray(img[:, :, c]), k2d, use_fma=bool(fma)), real[:, :, c])
                    for c in range(3))]
    report["filter2D"] = {"matches_use_fma": match, "max_abs_diff_fma": float(max(
        np.abs(orc.filter2D(np.ascontiguousarray(img[:, :, c]), k2d, True) - real[:, :, c]).max() for c in range(3)))}
    savez(os.path.join(GOLDEN, "cv2_filter2D.npz"), src=img, kernel=k2d, dst=real)

    # ---- cvtColor BGR2GRAY, float32 (pyramid.py:49) and uint8 (utils.py:37-43)
    lap = (img - real)
    real_g = cv2.cvtColor(lap, cv2.COLOR_BGR2GRAY)
    report["cvtColor_f32"] = {"matches_use_fma": [f for f in (1, 0) if np.array_equal(orc.bgr2gray_f32(lap, bool(f)), real_g)]}
    real_g8 = cv2.cvtColor(frames[0], cv2.COLOR_BGR2GRAY)
    report["cvtColor_u8"] = {"matches": bool(np.array_equal(orc.bgr2gray_int(frames[0]), real_g8))}
    savez(os.path.join(GOLDEN, "cv2_cvtColor.npz"), src_f32=lap, dst_f32=real_g, src_u8=frames[0], dst_u8=real_g8)

    # ---- warpAffine + mask + GaussianBlur composite (align.py:238-251)
    h, w = frames[0].shape[:2]
    M = np.array([[0.9998, -0.0123, 3.37], [0.0123, 0.9998, -2.21]], np.float32)
    real_w = cv2.warpAffine(frames[0], M, (w, h), borderMode=cv2.BORDER_REPLICATE)
    mask = cv2.warpAffine(np.ones_like(frames[0]), M, (w, h), borderMode=cv2.BORDER_CONSTANT, borderValue=0)
    blurred = cv2.GaussianBlur(real_w, (21, 21), sigmaX=50)
    comp = real_w.copy()
    comp[cv2.cvtColor(mask, cv2.COLOR_BGR2GRAY) == 0] = blurred[cv2.cvtColor(mask, cv2.COLOR_BGR2GRAY) == 0]
    # the blur on its own, 8 and 16 bit (OpenCV's fixed-point path as align_oracle.c restates it)
    src16 = frames[0].astype(np.uint16) * 257
    blurred16 = cv2.GaussianBlur(src16, (21, 21), sigmaX=50)
    report["GaussianBlur_21_fixed"] = {"u8": bool(np.array_equal(orc.gaussian_blur_fixed(real_w, 21, 50.0), blurred)),
                                       "u16": bool(np.array_equal(orc.gaussian_blur_fixed(src16, 21, 50.0), blurred16))}
    savez(os.path.join(GOLDEN, "cv2_gaussian_blur.npz"), src_u8=real_w, dst_u8=blurred, src_u16=src16, dst_u16=blurred16)
    report["warpAffine"] = {"matches": bool(np.array_equal(orc.warp_affine(frames[0], M, border_mode=1), real_w))}
    report["warp+blur_composite"] = {"matches": bool(np.array_equal(orc.warp_affine(frames[0], M), comp))}
    savez(os.path.join(GOLDEN, "cv2_warp.npz"), src=frames[0], M=M, warp=real_w, mask=mask, composite=comp)

    # ---- resize INTER_AREA (utils.py:79-86): sizes that divide and sizes that do not (output size, partial blocks), 8 / 16 bit
    res = {}
    odd = np.ascontiguousarray(frames[0][: h - 3, : w - 5])
    for name, src in (("even", frames[0]), ("odd", odd), ("odd16", odd.astype(np.uint16) * 257)):
        for s in (2, 3, 4, 8):
            real_r = cv2.resize(src, (0, 0), fx=1 / s, fy=1 / s, interpolation=cv2.INTER_AREA)
            mine = orc.resize_area_int(src, s)
            res[f"{name}_{s}"] = bool(mine.shape == real_r.shape and np.array_equal(mine, real_r))
            savez(os.path.join(GOLDEN, f"cv2_resize_area_{name}_{s}.npz"), src=src, dst=real_r)
    report["resize_INTER_AREA"] = res

    # ---- warpPerspective + mask (align.py:240-241, ALIGN_HOMOGRAPHY)
    Hm = np.array([[0.9997, -0.0121, 2.9], [0.0119, 1.0004, -1.7], [1.5e-6, -2.0e-6, 1.0]], np.float64)
    real_p = cv2.warpPerspective(frames[0], Hm, (w, h), borderMode=cv2.BORDER_REPLICATE)
    report["warpPerspective"] = {"matches": bool(np.array_equal(orc.warp_perspective(frames[0], Hm, border_mode=1), real_p))}
    savez(os.path.join(GOLDEN, "cv2_warp_perspective.npz"), src=frames[0], M=Hm, warp=real_p)

    # ---- 8-bit BGR <-> HSV / HLS (balance.py:340-363)
    cvt = {}
    r = np.arange(0, 256, 3, dtype=np.uint8)
    cube = np.stack(np.meshgrid(r, r, r, indexing="ij"), axis=-1).reshape(len(r), -1, 3)
    for name, code, cvcode in (("BGR2HSV", orc.CVT_BGR2HSV, cv2.COLOR_BGR2HSV), ("BGR2HLS", orc.CVT_BGR2HLS, cv2.COLOR_BGR2HLS)):
        real_c = cv2.cvtColor(cube, cvcode)
        cvt[name] = bool(np.array_equal(orc.cvt_color_u8(cube, code), real_c))
        back_code, back_cv = ((orc.CVT_HSV2BGR, cv2.COLOR_HSV2BGR) if name == "BGR2HSV" else (orc.CVT_HLS2BGR, cv2.COLOR_HLS2BGR))
        real_b = cv2.cvtColor(real_c, back_cv)
        cvt[name[4:] + "2BGR"] = bool(np.array_equal(orc.cvt_color_u8(real_c, back_code), real_b))
        savez(os.path.join(GOLDEN, f"cv2_cvt_{name}.npz"), src=cube, dst=real_c, back=real_b)
    report["cvtColor_HSV_HLS_u8"] = cvt

    # ---- DepthMapStack's primitives (depth_map.py:28-62, :94-112) on the gray plane of the first frame
    from oracle import depth_map_oracle as dmo
    gray = real_g8.astype(np.float32)
    dm = {}
    dm["GaussianBlur_5"] = bool(np.array_equal(dmo.gaussian_blur(gray, 5), cv2.GaussianBlur(gray, (5, 5), 0)))
    dm["Laplacian_64F_5"] = bool(np.array_equal(dmo.filter2d_f64(dmo.gaussian_blur(gray, 5), dmo.laplacian_kernel2d(5)),
                                                cv2.Laplacian(cv2.GaussianBlur(gray, (5, 5), 0), cv2.CV_64F, ksize=5)))
    dm["pyrDown"] = bool(np.array_equal(dmo.pyr_down(gray), cv2.pyrDown(gray)))
    small = cv2.pyrDown(gray)
    dm["pyrUp"] = bool(np.array_equal(dmo.pyr_up(small, (gray.shape[1], gray.shape[0])), cv2.pyrUp(small, dstsize=(gray.shape[1], gray.shape[0]))))
    en = (gray / 255.0).astype(np.float32)
    real_bi = cv2.bilateralFilter(en, 15, 25, 25)
    mine_bi = dmo.bilateral_f32(en, 15, 25.0, 25.0)
    dm["bilateralFilter_15"] = {"equal": bool(np.array_equal(mine_bi, real_bi)), "max_abs_diff": float(np.abs(mine_bi - real_bi).max())}
    report["depth_map_primitives"] = dm

    return report


def main():
    report = probe(write=True)
    if not report["available"]:
        print(f"probe_cv2: OpenCV is not importable here ({report['error']}); the cv2 primitives stay "
              "'parity unpinned' (oracle/pyramid_oracle.c header).  Nothing written.")
        return 0
    print(json.dumps(report, indent=1))
    with open(os.path.join(GOLDEN, "cv2_probe_report.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
