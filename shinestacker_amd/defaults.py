"""Defaults of the hot path, as the reference's config/constants.py holds them
(:11-14 pixel ranges, :100-110 alignment apply, :112-127 balance, :139-141 bunches, :144-157 depth map, :154-162 float types
and pyramid parameters).  Read-only."""
from types import SimpleNamespace

constants = SimpleNamespace(
    EXTENSIONS=frozenset(["jpeg", "jpg", "png", "tif", "tiff"]),
    NUM_UINT8=256, NUM_UINT16=65536, MAX_UINT8=255, MAX_UINT16=65535,
    FLOAT_32="float-32", FLOAT_64="float-64",
    DEFAULT_PY_FLOAT="float-32", DEFAULT_PY_MIN_SIZE=32, DEFAULT_PY_KERNEL_SIZE=5,
    DEFAULT_PY_GEN_KERNEL=0.4,
    # evaluation order of the pyramid stencils -- an option the reference does not have:
    # "separable" (5 + 5 tap form, within the stated float-32 tolerance of the reference's float-64 mode; the default of
    # PyramidStack()) or "exact" (the reference's own row-major 25-tap order, bit-identical to its restatement: the audit
    # mode).  ONE default for every high-level entry point (PyramidStack, align_and_stack, align_and_stack_device,
    # bunches_then_stack -- they all go through resolve_arith below); the C struct's zero value / `_lib.Stack` (the thin
    # binding) stay "exact", see include/mi355stack.h.  No environment override: the arithmetic a job ran with is the
    # stacker's `arith` attribute and is logged with the job (round 5).
    DEFAULT_PY_ARITH="separable",
    DEFAULT_FRAMES=10, DEFAULT_OVERLAP=2, DEFAULT_STACK_PREFIX="stack_",
    DEFAULT_PLOT_STACK=True, DEFAULT_PLOTS_PATH="plots", DEFAULT_FILE_REVERSE_ORDER=False,
    ALIGN_RIGID="ALIGN_RIGID", ALIGN_HOMOGRAPHY="ALIGN_HOMOGRAPHY",
    BORDER_CONSTANT="BORDER_CONSTANT", BORDER_REPLICATE="BORDER_REPLICATE",
    BORDER_REPLICATE_BLUR="BORDER_REPLICATE_BLUR",
    DEFAULT_TRANSFORM="ALIGN_RIGID", DEFAULT_BORDER_MODE="BORDER_REPLICATE_BLUR",
    DEFAULT_BORDER_VALUE=(0, 0, 0, 0), DEFAULT_BORDER_BLUR=50, DEFAULT_ALIGN_SUBSAMPLE=2,
    # feature / matching options of the reference's estimator (constants.py:69-94)
    DETECTOR_SIFT="SIFT", DETECTOR_ORB="ORB", DETECTOR_SURF="SURF", DETECTOR_AKAZE="AKAZE", DETECTOR_BRISK="BRISK",
    DESCRIPTOR_SIFT="SIFT", DESCRIPTOR_ORB="ORB", DESCRIPTOR_AKAZE="AKAZE", DESCRIPTOR_BRISK="BRISK",
    MATCHING_KNN="KNN", MATCHING_NORM_HAMMING="NORM_HAMMING", ALIGN_RANSAC="RANSAC", ALIGN_LMEDS="LMEDS",
    NOKNN_METHODS={'detectors': ["ORB", "SURF", "AKAZE", "BRISK"], 'descriptors': ["ORB", "AKAZE", "BRISK"]},
    # balance (constants.py:112-127)
    BALANCE_LINEAR="LINEAR", BALANCE_GAMMA="GAMMA", BALANCE_MATCH_HIST="MATCH_HIST",
    BALANCE_LUMI="LUMI", BALANCE_RGB="RGB", BALANCE_HSV="HSV", BALANCE_HLS="HLS",
    DEFAULT_BALANCE_SUBSAMPLE=8, DEFAULT_BALANCE_FAST_SUBSAMPLING=False,
    DEFAULT_CORR_MAP="LINEAR", DEFAULT_CHANNEL="LUMI",
    # depth map stacker (constants.py:144-157)
    DEFAULT_DM_FLOAT="float-32", DM_ENERGY_LAPLACIAN="laplacian", DM_ENERGY_SOBEL="sobel",
    DM_MAP_AVERAGE="average", DM_MAP_MAX="max",
    VALID_DM_MAP=("average", "max"), VALID_DM_ENERGY=("laplacian", "sobel"),
    DEFAULT_DM_MAP="average", DEFAULT_DM_ENERGY="laplacian", DEFAULT_DM_KERNEL_SIZE=5,
    DEFAULT_DM_BLUR_SIZE=5, DEFAULT_DM_SMOOTH_SIZE=15, DEFAULT_DM_TEMPERATURE=0.1, DEFAULT_DM_LEVELS=3,
)


def resolve_arith(arith=None, float_type=None):
    """The evaluation order a stack runs with: the caller's choice, else constants.DEFAULT_PY_ARITH ("exact" for float-64
    stacks, which the separable kernels do not serve)."""
    f64 = (float_type in (constants.FLOAT_64, "float64") or getattr(float_type, "__name__", None) == "float64"
           or (isinstance(float_type, int) and not isinstance(float_type, bool) and float_type == 3))   # 3 = _lib.MI_F64, Stack's code
    if arith is None:
        return "exact" if f64 else constants.DEFAULT_PY_ARITH
    return arith
