"""oracle/ref_import.py -- load the reference's OWN hot-path modules in this container.

Only usable where /root/reference exists (the build container); never on the GPU
box, never from the product.  Used by gen_golden.py to pin oracle.py's
reference-shaped restatement against the reference's NumPy control flow.

`import shinestacker` fails here (generated _version.py missing; cv2, tifffile,
psdtags not installed), so the parent packages are pre-seeded as empty modules
whose __path__ points at the real directories -- their __init__ never runs --
and a `cv2` shim supplies exactly the symbols pyramid.py / utils.py touch at
import or run time.  The shim's three numeric primitives are oracle.py's
(liboracle.so): that is what "parity unpinned for the cv2 primitives, pinned
for the control flow" means.
"""
import importlib
import os
import sys
import types

import numpy as np

from . import oracle as orc

REF_SRC = "/root/reference/src"


def available():
    return os.path.isdir(os.path.join(REF_SRC, "shinestacker", "algorithms"))


def make_cv2_shim(use_fma=True):
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_REFLECT101 = 4
    cv2.BORDER_REFLECT_101 = 4
    cv2.BORDER_CONSTANT = 0
    cv2.BORDER_REPLICATE = 1
    cv2.COLOR_BGR2GRAY = 6
    cv2.COLOR_BGR2RGB = 4
    cv2.COLOR_BGR2HSV, cv2.COLOR_HSV2BGR, cv2.COLOR_BGR2HLS, cv2.COLOR_HLS2BGR = 40, 54, 52, 60
    cv2.INTER_AREA = 3
    cv2.IMREAD_UNCHANGED = -1
    cv2.IMWRITE_JPEG_QUALITY = 1
    cv2.IMWRITE_TIFF_COMPRESSION = 259
    cv2.RANSAC = 8
    cv2.LMEDS = 4
    cv2.NORM_HAMMING = 6

    def filter2D(img, ddepth, kernel, borderType=None, **_kw):
        assert ddepth == -1 and borderType == cv2.BORDER_REFLECT101
        if img.ndim == 3:   # cv2 filters the channels of an interleaved image independently
            return np.stack([orc.filter2D(np.ascontiguousarray(img[..., c]), kernel, use_fma) for c in range(img.shape[2])], axis=-1)
        return orc.filter2D(np.ascontiguousarray(img), kernel, use_fma)

    def cvtColor(img, code):
        hue = {cv2.COLOR_BGR2HSV: orc.CVT_BGR2HSV, cv2.COLOR_HSV2BGR: orc.CVT_HSV2BGR,
               cv2.COLOR_BGR2HLS: orc.CVT_BGR2HLS, cv2.COLOR_HLS2BGR: orc.CVT_HLS2BGR}
        if code in hue:
            return orc.cvt_color_u8(img, hue[code])      # raises for 16-bit input, as cv2 does
        assert code == cv2.COLOR_BGR2GRAY
        if img.dtype in (np.uint8, np.uint16):
            return orc.bgr2gray_int(img)
        return orc.bgr2gray_f32(np.ascontiguousarray(img), use_fma)

    def LUT(img, lut):
        return np.asarray(lut)[img]

    def split(img):
        return [np.ascontiguousarray(img[..., c]) for c in range(img.shape[2])]

    def merge(chans):
        return np.stack(chans, axis=-1)

    def resize(img, dsize, fx=None, fy=None, interpolation=None):
        s = int(round(1.0 / fx))
        assert dsize == (0, 0) and fx == fy and abs(1.0 / fx - s) < 1e-9 and interpolation == cv2.INTER_AREA
        a = img if img.ndim == 3 else img[..., None]
        out = orc.resize_area_int(a, s)
        return out if img.ndim == 3 else out[..., 0]

    def copyMakeBorder(img, t, b, l, r, borderType):
        assert borderType == cv2.BORDER_REFLECT101 and t == b == l == r
        return orc.pad_reflect101(img, t)

    def _unavailable(*_a, **_k):
        raise RuntimeError("cv2 shim: function not available in the oracle container")

    # DepthMapStack's primitives (oracle/depth_map_oracle.py; parity unpinned, see there)
    from . import depth_map_oracle as dmo
    cv2.CV_64F = 6

    def Sobel(img, ddepth, dx, dy, ksize=3):
        assert ddepth == cv2.CV_64F and img.dtype in (np.float32, np.float64) and (dx, dy) in ((1, 0), (0, 1))
        kx, ky = dmo.sobel_kernels(dx, dy, ksize)
        return dmo.filter2d_f64(img, np.outer(ky, kx))

    def GaussianBlur(img, ksize, sigma=None, sigmaX=None):
        sigma = sigmaX if sigma is None else sigma
        assert ksize[0] == ksize[1]
        if img.dtype in (np.uint8, np.uint16):      # align.py:249 (the blurred border): cv2's fixed-point path
            return orc.gaussian_blur_fixed(img, ksize[0], sigma)
        assert sigma == 0 and img.dtype in (np.float32, np.float64)
        return dmo.gaussian_blur(img, ksize[0])

    # the alignment apply step (align.py:231-247): the RAW remap primitives of oracle/align_oracle.c -- no mask, no
    # composite; those are the reference's own lines and this shim must not pre-empt them
    def _border(mode, value):
        assert mode in (cv2.BORDER_CONSTANT, cv2.BORDER_REPLICATE)
        return (orc.BORDER_CONSTANT if mode == cv2.BORDER_CONSTANT else orc.BORDER_REPLICATE), \
            (0, 0, 0, 0) if value is None or np.isscalar(value) and value == 0 else value

    def warpAffine(img, m, dsize, borderMode=None, borderValue=None):
        assert dsize == (img.shape[1], img.shape[0]) and np.asarray(m).shape == (2, 3)
        mode, bv = _border(borderMode, borderValue)
        return orc.warp_affine_raw(img, np.asarray(m, np.float64), mode, bv)

    def warpPerspective(img, m, dsize, borderMode=None, borderValue=None):
        assert dsize == (img.shape[1], img.shape[0]) and np.asarray(m).shape == (3, 3)
        mode, bv = _border(borderMode, borderValue)
        return orc.warp_perspective_raw(img, np.asarray(m, np.float64), mode, bv)

    def getPerspectiveTransform(src, dst):
        from . import cv2_standin
        return cv2_standin.get_perspective_transform(src, dst)

    def Laplacian(img, ddepth, ksize=1):
        assert ddepth == cv2.CV_64F and img.dtype in (np.float32, np.float64)
        return dmo.filter2d_f64(img, dmo.laplacian_kernel2d(ksize))

    def bilateralFilter(img, d, sigma_color, sigma_space):
        assert img.dtype == np.float32 and img.ndim == 2
        return dmo.bilateral_f32(img, d, sigma_color, sigma_space)

    def pyrDown(img):
        assert img.dtype in (np.float32, np.float64)
        return dmo.pyr_down(img)

    def pyrUp(img, dstsize=None):
        assert img.dtype in (np.float32, np.float64)
        return dmo.pyr_up(img, dstsize)

    cv2.Sobel, cv2.Laplacian, cv2.bilateralFilter = Sobel, Laplacian, bilateralFilter
    cv2.pyrDown, cv2.pyrUp = pyrDown, pyrUp
    cv2.filter2D = filter2D
    cv2.cvtColor = cvtColor
    cv2.copyMakeBorder = copyMakeBorder
    cv2.LUT, cv2.split, cv2.merge, cv2.resize = LUT, split, merge, resize
    cv2.GaussianBlur = GaussianBlur
    cv2.warpAffine, cv2.warpPerspective, cv2.getPerspectiveTransform = warpAffine, warpPerspective, getPerspectiveTransform
    for name in ("imread", "imwrite",
                 "SIFT_create", "ORB_create", "AKAZE_create", "BRISK_create",
                 "FastFeatureDetector_create", "FlannBasedMatcher", "BFMatcher",
                 "findHomography", "estimateAffinePartial2D",
                 "drawMatches", "fastNlMeansDenoisingColored"):
        setattr(cv2, name, _unavailable)
    return cv2


class _NumpyWithExactLog:
    """Proxy for the `np` name inside the reference module: identical to numpy
    except float32 log is the correctly rounded one the oracle uses (NumPy's
    SIMD float32 log is CPU-dispatch dependent, so it cannot be a parity target);
    float64 log likewise goes through the x87 long-double logl and is rounded once."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def log(x):
        x = np.asarray(x)
        if x.dtype == np.float32:
            return np.log(x.astype(np.float64)).astype(np.float32)
        if x.dtype == np.float64:
            return np.log(x.astype(np.longdouble)).astype(np.float64)
        return np.log(x)


def load_pyramid_module(use_fma=True, exact_log=False):
    """Returns the reference's shinestacker.algorithms.pyramid module object."""
    if not available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True
    for name in [m for m in sys.modules if m == "cv2" or m.startswith("shinestacker")]:
        del sys.modules[name]
    for pkg in ("shinestacker", "shinestacker.algorithms", "shinestacker.core",
                "shinestacker.config"):
        mod = types.ModuleType(pkg)
        mod.__path__ = [os.path.join(REF_SRC, *pkg.split("."))]
        sys.modules[pkg] = mod
    sys.modules["cv2"] = make_cv2_shim(use_fma)
    mod = importlib.import_module("shinestacker.algorithms.pyramid")
    if exact_log:
        mod.np = _NumpyWithExactLog()
    return mod


class FakeProcess:
    """Stand-in for the owning action, as tests/test_0061_depth_map.py:43-48 does."""
    id = 0
    name = "oracle"

    def callback(self, *_a):
        return True

    def sub_message_r(self, *_a, **_k):
        pass


def reference_stack(frames, use_fma=True, exact_log=False, **algo_kwargs):
    """Run the reference's PyramidStack on in-memory frames (list of HxWx3 arrays).
    Mirrors focus_stack (pyramid.py:150-179) minus file I/O.  Returns (out, detail)."""
    mod = load_pyramid_module(use_fma, exact_log)
    algo = mod.PyramidStack(**algo_kwargs)
    algo.process = FakeProcess()
    first = frames[0]
    algo.dtype = first.dtype
    algo.num_pixel_values = 256 if first.dtype == np.uint8 else 65536
    algo.max_pixel_value = 255 if first.dtype == np.uint8 else 65535
    levels = int(np.log2(min(first.shape[:2]) / algo.min_size))
    pyrs = [algo.process_single_image(f, levels) for f in frames]
    fused = algo.fuse_pyramids(pyrs)
    collapsed = algo.collapse(fused)
    out = collapsed.astype(algo.dtype)
    return out, {"pyramids": pyrs, "fused": fused, "collapsed": collapsed, "algo": algo}


def load_balance_module():
    """The reference's shinestacker.algorithms.balance (its correction maps and Correction classes).
    matplotlib is stubbed when absent; no plot is ever requested."""
    load_pyramid_module()
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:  # noqa: BLE001
        mpl = types.ModuleType("matplotlib")
        plt = types.ModuleType("matplotlib.pyplot")
        mpl.pyplot = plt
        sys.modules["matplotlib"], sys.modules["matplotlib.pyplot"] = mpl, plt
    for stub in ("shinestacker.algorithms.exif", "shinestacker.algorithms.denoise"):
        if stub not in sys.modules:
            m = types.ModuleType(stub)
            m.copy_exif_from_file_to_file = lambda *a, **k: None
            m.denoise = lambda img, *a, **k: img
            sys.modules[stub] = m
    cfg = importlib.import_module("shinestacker.config.config").config
    try:
        cfg.init(DISABLE_TQDM=True)
    except Exception:  # noqa: BLE001
        pass
    return importlib.import_module("shinestacker.algorithms.balance")


def load_align_module(log=None):
    """The reference's shinestacker.algorithms.align (align.py:1-353), importable here: the numeric cv2 calls of its
    APPLY step (warpAffine / warpPerspective / GaussianBlur / cvtColor / resize / getPerspectiveTransform) are the raw
    primitives of oracle.py / align_oracle.c, the calls of its ESTIMATE step (feature detectors, matchers,
    estimateAffinePartial2D / findHomography) are oracle/cv2_standin.py.  Everything between those calls -- sub-sampling
    and the retry without it, the min-matches rules, the rescale of the transform, the float32 cast, the mask warp,
    `mask == 0` composite, the argument order -- is the reference's own code.  `log` collects the stand-in's calls."""
    load_balance_module()            # pyramid + stubs for exif / denoise / matplotlib + config
    from . import cv2_standin
    cv2 = sys.modules["cv2"]
    cv2_standin.install_features(cv2, [] if log is None else log)
    plt = sys.modules["matplotlib.pyplot"]
    for name in ("figure", "imshow", "savefig", "plot", "close"):
        if not hasattr(plt, name):
            setattr(plt, name, lambda *a, **k: None)
    return importlib.import_module("shinestacker.algorithms.align")


class _NumpyWithExactExp:
    """`np` inside the reference's depth_map module: float32 exp is the correctly rounded one
    (depth_map_oracle.exp_f32), for the same reason as _NumpyWithExactLog."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def exp(x):
        x = np.asarray(x)
        if x.dtype == np.float32:
            return np.exp(x.astype(np.float64)).astype(np.float32)
        if x.dtype == np.float64:
            return np.exp(x.astype(np.longdouble)).astype(np.float64)
        return np.exp(x)


def reference_depth_map(frames, **algo_kwargs):
    """Run the reference's DepthMapStack.focus_stack (depth_map.py:64-123) on in-memory frames: the
    file reads are replaced by look-ups ("file names" are the frame indices), everything else is the
    reference's code over the cv2 shim.  Returns (fused frame, callback trace)."""
    load_pyramid_module()
    mod = importlib.import_module("shinestacker.algorithms.depth_map")
    base = importlib.import_module("shinestacker.algorithms.base_stack_algo")
    mod.np = _NumpyWithExactExp()
    reader = lambda path: frames[int(path)].copy()  # noqa: E731
    mod.read_img = reader
    base.read_img = reader
    algo = mod.DepthMapStack(**algo_kwargs)
    trace = []

    class Proc(FakeProcess):
        def callback(self, key, *a):
            trace.append((key,) + tuple(a[2:]))
            return True
    algo.process = Proc()
    out = algo.focus_stack([str(i) for i in range(len(frames))])
    return out, trace
