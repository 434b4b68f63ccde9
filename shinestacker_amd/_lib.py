"""ctypes binding of libmi355stack.so (include/mi355stack.h).

There is no CPU fallback: if the shared object is missing or no HIP device is
visible, every compute entry point raises DeviceError.
"""
import ctypes as C
import os

import numpy as np

from .errors import DeviceError, InvalidOptionError

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355STACK_LIB: another build of the same library (tools/study_build.sh puts its -DMI_STUDY variant there)
LIB_PATH = os.environ.get("MI355STACK_LIB") or os.path.join(_HERE, "csrc", "libmi355stack.so")

# enums of mi355stack.h
MI_OK, MI_ERR_INVALID, MI_ERR_NO_DEVICE, MI_ERR_HIP, MI_ERR_STATE, MI_ERR_NOMEM, \
    MI_ERR_UNSUPPORTED, MI_ERR_ALIGNMENT = range(8)
MI_U8, MI_U16, MI_F32, MI_F64 = range(4)
(TAP_GAUSS, TAP_FUSED_LAP, TAP_ENERGY, TAP_INDEX, TAP_FUSED_BASE, TAP_BASE_IDX_E,
 TAP_BASE_IDX_D, TAP_COLLAPSED, TAP_BASE_ENT, TAP_BASE_DEV) = range(10)
IMPL_AUTO, IMPL_SIMPLE, IMPL_TILED = range(3)
PROF_LEVEL, PROF_BASE, PROF_COLLAPSE, PROF_LEVEL0 = range(4)
ARITH_EXACT, ARITH_SEPARABLE = 0, 1
ARITH_CODE = {"exact": ARITH_EXACT, "separable": ARITH_SEPARABLE, ARITH_EXACT: ARITH_EXACT,
              ARITH_SEPARABLE: ARITH_SEPARABLE}

DTYPE_CODE = {np.dtype(np.uint8): MI_U8, np.dtype(np.uint16): MI_U16,
              np.dtype(np.float32): MI_F32}
CODE_DTYPE = {v: k for k, v in DTYPE_CODE.items()}


class StackParams(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("in_dtype", C.c_int32),
                ("out_dtype", C.c_int32), ("min_size", C.c_int32), ("kernel_size", C.c_int32),
                ("gen_kernel", C.c_double), ("float_type", C.c_int32), ("use_fma", C.c_int32),
                ("device", C.c_int32), ("impl", C.c_int32), ("batch_frames", C.c_int32),
                ("arith", C.c_int32), ("pair_levels", C.c_int32), ("reserved", C.c_int32 * 3)]


class AlignStackOpts(C.Structure):
    """mi_align_stack_opts_t"""
    _fields_ = [("transform", C.c_int), ("border_mode", C.c_int), ("border_value", C.c_double * 4),
                ("blur_ksize", C.c_int), ("blur_sigma", C.c_double), ("min_correlation", C.c_double),
                ("max_iters", C.c_int), ("eps", C.c_double), ("ecc_batch", C.c_int), ("batch_frames", C.c_int)]


class BalanceLinearOpts(C.Structure):
    """mi_balance_linear_opts_t"""
    _fields_ = [("mode", C.c_int), ("subsample", C.c_int), ("fast", C.c_int), ("mask_size", C.c_double),
                ("lo", C.c_int), ("hi", C.c_int), ("first_channel", C.c_int), ("cvt_to", C.c_int), ("cvt_from", C.c_int),
                ("ref_means", C.c_double * 3), ("dev_hist_scratch", C.c_void_p), ("dev_lut", C.c_void_p),
                ("dev_corr_out", C.c_void_p), ("ncorr", C.c_int)]


class DepthMapParams(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("dtype", C.c_int32), ("device", C.c_int32),
                ("map_type", C.c_int32), ("energy", C.c_int32), ("kernel_size", C.c_int32),
                ("blur_size", C.c_int32), ("smooth_size", C.c_int32), ("levels", C.c_int32),
                ("temperature", C.c_float), ("float_type", C.c_int32)]


DM_MAP_AVERAGE, DM_MAP_MAX = 0, 1
DM_ENERGY_LAPLACIAN, DM_ENERGY_SOBEL = 0, 1

# name -> (restype, argtypes); also the list the symbol-export test walks
SIGNATURES = {
    "mi_abi_version": (C.c_int, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mi_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "mi_stack_default_params": (None, [C.POINTER(StackParams)]),
    "mi_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mi_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "mi_memcpy_h2d": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_memcpy_d2h": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_memcpy_d2d": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_memcpy_d2d_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_device_synchronize": (C.c_int, [C.c_int]),
    "mi_pyr_step": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "mi_device_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "mi_stack_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(StackParams)]),
    "mi_stack_destroy": (None, [C.c_void_p]),
    "mi_stack_reset": (C.c_int, [C.c_void_p]),
    "mi_stack_levels": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mi_stack_level_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]),
    "mi_stack_frames_pushed": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mi_stack_set_first_index": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_stack_set_index_stride": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_stack_export_indices": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_stack_push_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_stack_push_frame_pinned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_stack_wait_uploads": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "mi_host_free": (C.c_int, [C.c_void_p]),
    "mi_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "mi_host_unregister": (C.c_int, [C.c_void_p]),
    "mi_stack_push_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "mi_stack_sync": (C.c_int, [C.c_void_p]),
    "mi_stack_sync_level": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_stack_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_stack_finish_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_stack_get_level": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "mi_stack_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_size_t)]),
    "mi_stack_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mi_stack_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_stack_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "mi_combine_select": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_combine_winner": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mi_combine_winner_idx": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mi_combine_plan_bytes": (C.c_size_t, [C.c_size_t, C.c_int]),
    "mi_combine_plan": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                  C.POINTER(C.c_int64)]),
    "mi_combine_pack": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_void_p]),
    "mi_combine_unpack": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_void_p]),
    "mi_warp_affine": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                 C.c_double]),
    "mi_warp_affine_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                        C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double]),
    "mi_warp_perspective": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double]),
    "mi_warp_perspective_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                             C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double]),
    "mi_ecc_similarity": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int)]),
    "mi_aligner_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int]),
    "mi_aligner_set_area_subsampling": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_aligner_set_phase_init": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_aligner_estimate_homography_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                                       C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                       C.POINTER(C.c_int)]),
    "mi_phase_correlate_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.POINTER(C.c_double)]),
    "mi_aligner_destroy": (C.c_int, [C.c_void_p]),
    "mi_aligner_set_reference": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_aligner_estimate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mi_histogram": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_double, C.c_void_p]),
    "mi_histogram_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "mi_histogram_device_batch": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "mi_apply_lut": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                               C.c_int]),
    "mi_apply_lut_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                      C.c_void_p, C.c_int]),
    "mi_balance_linear_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_double), C.c_void_p]),
    "mi_cvt_color": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi_cvt_color_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "mi_aligner_estimate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                            C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                            C.POINTER(C.c_int)]),
    "mi_aligner_estimate_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int,
                                            C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mi_aligner_refine_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_double),
                                          C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_int)]),
    "mi_align_stack_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int,
                                        C.POINTER(AlignStackOpts), C.POINTER(BalanceLinearOpts), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mi_dmap_default_params": (None, [C.POINTER(DepthMapParams)]),
    "mi_dmap_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(DepthMapParams)]),
    "mi_dmap_destroy": (None, [C.c_void_p]),
    "mi_dmap_reset": (C.c_int, [C.c_void_p]),
    "mi_dmap_set_temperature": (C.c_int, [C.c_void_p, C.c_double]),
    "mi_dmap_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mi_dmap_frames_pushed": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mi_dmap_push_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_dmap_push_frame_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_dmap_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi_dmap_finish_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_synth_frames_device": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_uint32]),
}

_lib = None


def load():
    """dlopen the library and declare every prototype. Raises DeviceError if absent."""
    global _lib
    if _lib is None:
        # In a torch.distributed job torch's bundled HIP runtime must be loaded before this
        # library pulls in /opt/rocm's: two runtimes in one process leave torch without a GPU.
        import sys
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "torch" not in sys.modules:
            import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise DeviceError(
                f"{LIB_PATH} not found: build it with `python -m shinestacker_amd.build` "
                "(there is no CPU fallback)")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise DeviceError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.mi_abi_version() != 1:
            raise DeviceError("libmi355stack.so ABI version mismatch")
        _lib = lib
    return _lib


def last_error():
    return load().mi_last_error().decode("utf-8", "replace")


def check(rc):
    """Map an MI_* status to the exception types of the boundary."""
    if rc == MI_OK:
        return
    msg = last_error()
    if rc == MI_ERR_INVALID:
        raise ValueError(msg)
    if rc == MI_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == MI_ERR_UNSUPPORTED:
        raise InvalidOptionError("hip", "unsupported", msg)
    raise DeviceError(msg)


def device_count():
    n = C.c_int(0)
    rc = load().mi_device_count(C.byref(n))
    return n.value if rc == MI_OK else 0


def require_device():
    if device_count() < 1:
        raise DeviceError("no HIP device visible: the MI355X path cannot run "
                          "(there is no CPU fallback)")


# ---- pinned host memory (zero-copy uploads: Stack.push_frame(frame, zero_copy=True) sends an array that lies in it
# straight over PCIe)
_PINNED = {}    # base address -> bytes, of every live pinned range this module knows about


def host_alloc(shape, dtype):
    """An ndarray in pinned host memory (mi_host_alloc): decode / generate into it, then `Stack.push_frame(arr,
    zero_copy=True)` uploads it without the bounce copy.  Freed when the array (and every view of it) is garbage collected."""
    import weakref
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    p = C.c_void_p()
    check(load().mi_host_alloc(C.byref(p), nbytes))
    addr = p.value
    raw = (C.c_char * nbytes).from_address(addr)
    arr = np.frombuffer(raw, dtype=dt).reshape(shape)
    _PINNED[addr] = nbytes

    def release(a=addr):
        _PINNED.pop(a, None)
        load().mi_host_free(a)
    weakref.finalize(raw, release)
    return arr


def host_register(arr):
    """Pin the memory of an existing C-contiguous array in place (mi_host_register); undo with host_unregister.  The
    registration also ends when the array that owns the memory is garbage collected, so a later allocation at the same
    address is never mistaken for pinned memory."""
    import weakref
    a = np.asarray(arr)
    if not a.flags.c_contiguous:
        raise ValueError("only C-contiguous arrays can be pinned")
    addr = a.ctypes.data
    check(load().mi_host_register(addr, a.nbytes))
    _PINNED[addr] = a.nbytes
    owner = a
    while isinstance(owner.base, np.ndarray):   # the array whose death frees the memory
        owner = owner.base

    def release(ad=addr):
        if _PINNED.pop(ad, None) is not None:
            load().mi_host_unregister(ad)
    try:
        weakref.finalize(owner, release)
    except TypeError:       # an owner that cannot be weakly referenced: the caller unregisters
        pass


def host_unregister(arr):
    a = np.asarray(arr)
    if _PINNED.pop(a.ctypes.data, None) is not None:
        check(load().mi_host_unregister(a.ctypes.data))


def is_pinned(arr):
    """does the array lie inside a pinned range made by host_alloc / host_register?"""
    a0 = arr.ctypes.data
    a1 = a0 + arr.nbytes
    return any(b <= a0 and a1 <= b + n for b, n in _PINNED.items())


PYR_CONVOLVE, PYR_REDUCE, PYR_EXPAND, PYR_FUSE_LAPLACIAN, PYR_COLLAPSE_STEP, PYR_CLIP_ABS = range(6)


def pyr_step(op, img, img2=None, gen_kernel=0.4, use_fma=True, maxv=255.0, device=0):
    """mi_pyr_step on float32 host arrays: (h, w), (h, w, 1 | 3), or (n, h, w, 3) for PYR_FUSE_LAPLACIAN"""
    require_device()
    a = np.ascontiguousarray(img, np.float32)
    n = 1
    if op == PYR_FUSE_LAPLACIAN:
        n, h, w, c = a.shape
    else:
        h, w = a.shape[:2]
        c = 1 if a.ndim == 2 else a.shape[2]
    tail = () if a.ndim == 2 else (c,)
    shape = {PYR_REDUCE: ((h + 1) // 2, (w + 1) // 2), PYR_EXPAND: (2 * h, 2 * w)}.get(op, (h, w)) + tail
    out = np.empty(shape, np.float32)
    b, h2, w2 = None, 0, 0
    if img2 is not None:
        b = np.ascontiguousarray(img2, np.float32)
        h2, w2 = b.shape[:2]
    check(load().mi_pyr_step(device, int(op), int(bool(use_fma)), float(gen_kernel), a.ctypes.data,
                             b.ctypes.data if b is not None else None, int(n), int(h), int(w), int(c), int(h2), int(w2),
                             float(maxv), out.ctypes.data))
    return out


def mem_info(device=0):
    """(free, total) device memory in bytes"""
    f, t = C.c_size_t(), C.c_size_t()
    check(load().mi_device_mem_info(device, C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def device_name(device=0):
    buf = C.create_string_buffer(256)
    check(load().mi_device_name(device, buf, 256))
    return buf.value.decode()


class DeviceBuffer:
    """Owning wrapper of a mi_device_malloc allocation."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(load().mi_device_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        check(load().mi_memcpy_h2d(self.device, self.ptr + offset, a.ctypes.data, a.nbytes))

    def download(self, shape, dtype, offset=0):
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        check(load().mi_memcpy_d2h(self.device, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            load().mi_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class Stack:
    """Thin object wrapper over an mi_stack_t handle."""

    def __init__(self, height, width, in_dtype=np.uint8, out_dtype=None, min_size=32,
                 kernel_size=5, gen_kernel=0.4, use_fma=True, device=0, impl=IMPL_AUTO,
                 batch_frames=0, float_type=MI_F32, arith=ARITH_EXACT, pair_levels=None):
        lib = load()
        require_device()
        p = StackParams()
        lib.mi_stack_default_params(C.byref(p))
        in_dtype = np.dtype(in_dtype)
        out_dtype = np.dtype(out_dtype if out_dtype is not None else
                             (np.uint8 if in_dtype == np.float32 else in_dtype))
        p.height, p.width = int(height), int(width)
        p.in_dtype, p.out_dtype = DTYPE_CODE[in_dtype], DTYPE_CODE[out_dtype]
        p.min_size, p.kernel_size, p.gen_kernel = int(min_size), int(kernel_size), float(gen_kernel)
        p.float_type, p.use_fma, p.device = float_type, int(bool(use_fma)), int(device)
        p.impl, p.batch_frames = int(impl), int(batch_frames)
        if arith not in ARITH_CODE:
            raise InvalidOptionError("arith", arith, "valid values are 'exact' and 'separable'")
        p.arith = ARITH_CODE[arith]
        if pair_levels is None:   # (test runs: SHINESTACKER_AMD_PAIR_LEVELS=1 sends every separable stack down the pair kernels)
            pair_levels = int(os.environ.get("SHINESTACKER_AMD_PAIR_LEVELS", "0"))
        if pair_levels not in (0, 1, 2, 3):
            raise InvalidOptionError("pair_levels", pair_levels, "0 = automatic, 1 = pairs from level 0 on, 2 = none, 3 = pairs from level 1 on")
        p.pair_levels = int(pair_levels)
        self.arith = p.arith
        self.index_stride = 1
        self.params = p
        self.in_dtype, self.out_dtype = in_dtype, out_dtype
        self.float_type = int(float_type)
        self.height, self.width, self.device = p.height, p.width, p.device
        h = C.c_void_p()
        check(lib.mi_stack_create(C.byref(h), C.byref(p)))
        self._h = h
        self._inflight = []     # pinned frames whose zero-copy upload may still be running (push_frame(zero_copy=True))
        n = C.c_int()
        check(lib.mi_stack_levels(self._h, C.byref(n)))
        self.levels = n.value
        self.shapes = [self.level_shape(l) for l in range(self.levels + 1)]

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None):
            load().mi_stack_destroy(self._h)   # waits for the handle's streams: no upload reads a pinned frame any more
            self._h = None
        self._inflight = []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        check(load().mi_stack_reset(self._h))   # synchronises the handle: every upload has completed
        self._inflight = []

    # -- geometry
    def level_shape(self, level):
        h, w = C.c_int(), C.c_int()
        check(load().mi_stack_level_shape(self._h, level, C.byref(h), C.byref(w)))
        return h.value, w.value

    @property
    def frames_pushed(self):
        n = C.c_int()
        check(load().mi_stack_frames_pushed(self._h, C.byref(n)))
        return n.value

    def set_first_index(self, idx, stride=1):
        """global index of this handle's k-th frame = idx + k * stride (stride > 1: interleaved frame shards)"""
        check(load().mi_stack_set_first_index(self._h, int(idx)))
        check(load().mi_stack_set_index_stride(self._h, int(stride)))
        self.index_stride = int(stride)

    def export_indices(self, level=-1):
        """winner indices of `level` (-1: all levels and both base twins) into the global numbering (no-op at stride 1)"""
        check(load().mi_stack_export_indices(self._h, int(level)))

    # -- data path
    def push_frame(self, frame, zero_copy=False):
        """Push one host frame.  Default: the frame is copied (bounce buffers) before the call returns, whatever memory it
        lies in -- the caller may overwrite it at once.  `zero_copy=True` asks for the asynchronous upload straight out of
        the caller's PINNED array (`host_alloc` / `host_register`): the array is kept alive here and must not be modified
        until `wait_uploads()` (or any sync / reset / finish of the handle) has returned; an array that turns out not to
        be pinned takes the copying path."""
        a = np.asarray(frame)
        if a.shape != (self.height, self.width, 3):
            raise ValueError(f"frame shape {a.shape} != {(self.height, self.width, 3)}")
        if a.dtype != self.in_dtype:
            raise ValueError(f"frame dtype {a.dtype} != {self.in_dtype}")
        if not a.flags.c_contiguous:
            if a.strides[1:] == (3 * a.itemsize, a.itemsize) and a.strides[0] > 0:
                check(load().mi_stack_push_frame(self._h, a.ctypes.data, a.strides[0]))
                return
            a = np.ascontiguousarray(a)
        if zero_copy and is_pinned(a):
            rc = load().mi_stack_push_frame_pinned(self._h, a.ctypes.data, 0)
            if rc == MI_OK:
                self._inflight = getattr(self, "_inflight", [])
                self._inflight.append(a)
                if len(self._inflight) > 64:
                    self.wait_uploads(32)
                return
            if rc != MI_ERR_INVALID:     # MI_ERR_INVALID: the library does not see pinned memory there (stale record)
                check(rc)
        check(load().mi_stack_push_frame(self._h, a.ctypes.data, 0))

    def wait_uploads(self, max_outstanding=0):
        """block until at most `max_outstanding` of the pinned frames pushed so far are still being uploaded: a producer that
        cycles k pinned buffers calls wait_uploads(k - 1) before it overwrites the oldest"""
        check(load().mi_stack_wait_uploads(self._h, int(max_outstanding)))
        fl = getattr(self, "_inflight", None)
        if fl:
            del fl[:max(0, len(fl) - int(max_outstanding))]

    def push_frames_device(self, dev_ptr, n, frame_stride_bytes=0):
        check(load().mi_stack_push_frames_device(self._h, dev_ptr, int(n),
                                                 int(frame_stride_bytes)))

    def sync(self):
        check(load().mi_stack_sync(self._h))
        self._inflight = []

    def sync_level(self, level):
        """wait until the state of `level` covers every pushed frame (level 0: before the coarser levels finish)"""
        check(load().mi_stack_sync_level(self._h, int(level)))

    def finish(self):
        out = np.empty((self.height, self.width, 3), self.out_dtype)
        check(load().mi_stack_finish(self._h, out.ctypes.data, 0))
        self._inflight = []      # the result is on the host: every upload it depends on has completed
        return out

    def finish_device(self, dev_ptr=None):
        # the collapse is enqueued behind every push of the handle; the uploads themselves have been waited for by then
        # (a frame's kernels run behind its copy), but the result is asynchronous: the pinned frames stay referenced until
        # the next sync / reset / wait_uploads
        check(load().mi_stack_finish_device(self._h, dev_ptr))

    # -- taps
    def tap(self, what, level=0):
        L = self.levels
        ft = np.float64 if self.float_type == MI_F64 else np.float32   # pyramid.py:126 float_type
        if what in (TAP_GAUSS, TAP_FUSED_LAP):
            shape, dt = self.shapes[level] + (3,), ft
        elif what == TAP_ENERGY:
            shape, dt = self.shapes[level], np.float32
        elif what == TAP_INDEX:
            shape, dt = self.shapes[level], np.int32
        elif what == TAP_FUSED_BASE:
            shape, dt, level = self.shapes[L] + (3,), ft, L
        elif what in (TAP_BASE_IDX_E, TAP_BASE_IDX_D):
            shape, dt, level = self.shapes[L], np.int32, L
        elif what in (TAP_BASE_ENT, TAP_BASE_DEV):
            shape, dt, level = self.shapes[L], ft, L
        elif what == TAP_COLLAPSED:
            shape, dt, level = (self.height, self.width, 3), ft, 0
        else:
            raise ValueError(f"unknown tap {what}")
        out = np.empty(shape, dt)
        check(load().mi_stack_get_level(self._h, level, what, out.ctypes.data, out.nbytes))
        return out

    def state_ptrs(self, level):
        e, l, i, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        check(load().mi_stack_state(self._h, level, C.byref(e), C.byref(l), C.byref(i),
                                    C.byref(n)))
        return e.value, l.value, i.value, n.value

    @property
    def stream(self):
        s = C.c_void_p()
        check(load().mi_stack_stream(self._h, C.byref(s)))
        return s.value

    # -- timing
    def profile(self, enable=True):
        check(load().mi_stack_profile(self._h, int(enable)))

    def profile_get(self, kind):
        ms, n, b = C.c_double(), C.c_int64(), C.c_double()
        check(load().mi_stack_profile_get(self._h, kind, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value


class DepthMap:
    """Thin object wrapper over an mi_dmap_t handle (DepthMapStack's arithmetic, include/mi355stack.h)."""

    def __init__(self, height, width, dtype=np.uint8, map_type=DM_MAP_AVERAGE, energy=DM_ENERGY_LAPLACIAN,
                 kernel_size=5, blur_size=5, smooth_size=15, temperature=0.1, levels=3, device=0,
                 float_type=MI_F32):
        lib = load()
        require_device()
        p = DepthMapParams()
        lib.mi_dmap_default_params(C.byref(p))
        self.dtype = np.dtype(dtype)
        p.height, p.width, p.dtype, p.device = int(height), int(width), DTYPE_CODE[self.dtype], int(device)
        p.map_type, p.energy = int(map_type), int(energy)
        p.kernel_size, p.blur_size, p.smooth_size = int(kernel_size), int(blur_size), int(smooth_size)
        p.levels, p.temperature, p.float_type = int(levels), float(temperature), int(float_type)
        self.height, self.width, self.device = p.height, p.width, p.device
        h = C.c_void_p()
        check(lib.mi_dmap_create(C.byref(h), C.byref(p)))
        self._h = h
        if int(map_type) == DM_MAP_MAX:
            check(lib.mi_dmap_set_temperature(h, float(temperature)))   # the exact double (float-64 stacks divide by it)

    def close(self):
        if getattr(self, "_h", None):
            load().mi_dmap_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        check(load().mi_dmap_reset(self._h))

    def planes(self, stage, planes, out_dtype):
        """mi_dmap_planes: one step of the stacker on n host planes (n x H x W) -> n planes of `out_dtype`"""
        a = np.ascontiguousarray(planes)
        if a.ndim != 3 or a.shape[1:] != (self.height, self.width):
            raise ValueError(f"planes of shape {a.shape}, expected (n, {self.height}, {self.width})")
        out = np.empty(a.shape, out_dtype)
        check(load().mi_dmap_planes(self._h, int(stage), a.ctypes.data, a.shape[0], out.ctypes.data))
        return out

    @property
    def frames_pushed(self):
        n = C.c_int()
        check(load().mi_dmap_frames_pushed(self._h, C.byref(n)))
        return n.value

    def push_frame(self, frame):
        a = np.asarray(frame)
        if a.shape != (self.height, self.width, 3):
            raise ValueError(f"frame shape {a.shape} != {(self.height, self.width, 3)}")
        if a.dtype != self.dtype:
            raise ValueError(f"frame dtype {a.dtype} != {self.dtype}")
        a = np.ascontiguousarray(a)
        check(load().mi_dmap_push_frame(self._h, a.ctypes.data, 0))

    def push_frame_device(self, dev_ptr):
        check(load().mi_dmap_push_frame_device(self._h, dev_ptr))

    def finish(self):
        out = np.empty((self.height, self.width, 3), self.dtype)
        check(load().mi_dmap_finish(self._h, out.ctypes.data, 0))
        return out

    def finish_device(self, dev_ptr=None):
        check(load().mi_dmap_finish_device(self._h, dev_ptr))


def synth_frames_device(dev_ptr, dtype, height, width, first_frame, n_frames, stack_size,
                        seed=20250824, device=0, frame_step=1):
    """frames first_frame, first_frame + frame_step, ... of the `stack_size`-frame generator stack, packed at dev_ptr"""
    code = DTYPE_CODE[np.dtype(dtype)]
    if frame_step == 1:
        check(load().mi_synth_frames_device(device, dev_ptr, code, height, width, first_frame, n_frames, stack_size, seed))
        return
    per = height * width * 3 * np.dtype(dtype).itemsize
    for k in range(n_frames):
        check(load().mi_synth_frames_device(device, dev_ptr + k * per, code, height, width, first_frame + k * frame_step, 1,
                                            stack_size, seed))


BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REPLICATE_BLUR = 0, 1, 2


def warp_affine(img, M, border_mode=BORDER_REPLICATE_BLUR, border_value=(0, 0, 0, 0), blur_ksize=21,
                blur_sigma=50.0, want_mask=False, device=0):
    """cv2.warpAffine + warped mask + blurred-border composite on the GPU (mi_warp_affine)."""
    require_device()
    a = np.ascontiguousarray(img)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("warp_affine expects an H x W x 3 uint8/uint16 image")
    h, w = a.shape[:2]
    m = (C.c_double * 6)(*np.asarray(M, dtype=np.float64).reshape(6))
    bv = (C.c_double * 4)(*(list(border_value) + [0, 0, 0, 0])[:4])
    out = np.empty_like(a)
    mask = np.empty((h, w), np.uint8) if want_mask else None
    check(load().mi_warp_affine(device, a.ctypes.data, out.ctypes.data,
                                mask.ctypes.data if want_mask else None, h, w, DTYPE_CODE[a.dtype], m,
                                int(border_mode), bv, int(blur_ksize), float(blur_sigma)))
    return (out, mask) if want_mask else out


def warp_perspective(img, M, border_mode=BORDER_REPLICATE_BLUR, border_value=(0, 0, 0, 0), blur_ksize=21,
                     blur_sigma=50.0, want_mask=False, device=0):
    """cv2.warpPerspective + warped mask + blurred-border composite on the GPU (mi_warp_perspective); M: 3x3."""
    require_device()
    a = np.ascontiguousarray(img)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("warp_perspective expects an H x W x 3 uint8/uint16 image")
    h, w = a.shape[:2]
    m = (C.c_double * 9)(*np.asarray(M, dtype=np.float64).reshape(9))
    bv = (C.c_double * 4)(*(list(border_value) + [0, 0, 0, 0])[:4])
    out = np.empty_like(a)
    mask = np.empty((h, w), np.uint8) if want_mask else None
    check(load().mi_warp_perspective(device, a.ctypes.data, out.ctypes.data,
                                     mask.ctypes.data if want_mask else None, h, w, DTYPE_CODE[a.dtype], m,
                                     int(border_mode), bv, int(blur_ksize), float(blur_sigma)))
    return (out, mask) if want_mask else out


def ecc_similarity(ref, mov, max_levels=0, max_iters=60, eps=1e-9, device=0):
    """GPU ECC estimate of the similarity that maps `mov` onto `ref` (mi_ecc_similarity).
    Returns (M 2x3 float64, correlation coefficient, iterations)."""
    require_device()
    a, b = np.ascontiguousarray(ref), np.ascontiguousarray(mov)
    if a.shape != b.shape or a.dtype != b.dtype or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("ecc_similarity expects two H x W x 3 images of the same shape and dtype")
    m = (C.c_double * 6)()
    cc, it = C.c_double(), C.c_int()
    check(load().mi_ecc_similarity(device, a.ctypes.data, b.ctypes.data, a.shape[0], a.shape[1],
                                   DTYPE_CODE[a.dtype], int(max_levels), int(max_iters), float(eps), m,
                                   C.byref(cc), C.byref(it)))
    return np.array(list(m), dtype=np.float64).reshape(2, 3), cc.value, it.value


def phase_correlate(ref, mov, device=0):
    """(dx, dy, response) with mov(x + dx, y + dy) ~ ref(x, y) for two float32 planes (mi_phase_correlate_device)"""
    require_device()
    a, b = np.ascontiguousarray(ref, np.float32), np.ascontiguousarray(mov, np.float32)
    if a.ndim != 2 or a.shape != b.shape:
        raise ValueError("phase_correlate expects two H x W planes of the same shape")
    buf = DeviceBuffer(2 * a.nbytes, device)
    try:
        buf.upload(a)
        buf.upload(b, a.nbytes)
        out = (C.c_double * 3)()
        check(load().mi_phase_correlate_device(device, None, buf.ptr, buf.ptr + a.nbytes, a.shape[0], a.shape[1], out))
    finally:
        buf.free()
    return out[0], out[1], out[2]


class Aligner:
    """Device-resident ECC estimator (mi_aligner_t): frames stay in HBM, the pyramids are allocated
    once, `subsample` is the reference's fast sub-sampling factor (align.py default 2)."""

    def __init__(self, height, width, dtype=np.uint8, subsample=1, max_levels=0, device=0, fast=True, phase_init=False):
        require_device()
        self._h = C.c_void_p()
        self.device = device
        check(load().mi_aligner_create(C.byref(self._h), device, height, width,
                                       DTYPE_CODE[np.dtype(dtype)], int(subsample), int(max_levels)))
        if not fast and subsample > 1:   # the reference's default: cv2.resize(INTER_AREA) (utils.py:83)
            check(load().mi_aligner_set_area_subsampling(self._h, 1))
        if phase_init:                   # translation by phase correlation as the starting point (mi_aligner_set_phase_init)
            check(load().mi_aligner_set_phase_init(self._h, 1))

    def set_reference(self, dev_ptr, stream=None):
        check(load().mi_aligner_set_reference(self._h, stream, dev_ptr))

    def estimate(self, dev_ptr, max_iters=60, eps=1e-9, stream=None):
        """-> (M 2x3 float64 in full-resolution pixels, correlation coefficient, iterations)"""
        m = (C.c_double * 6)()
        cc, it = C.c_double(), C.c_int()
        check(load().mi_aligner_estimate(self._h, stream, dev_ptr, int(max_iters), float(eps), m,
                                         C.byref(cc), C.byref(it)))
        return np.array(list(m), dtype=np.float64).reshape(2, 3), cc.value, it.value

    MAX_BATCH = 128

    def estimate_batch(self, dev_ptrs, max_iters=60, eps=1e-9, stream=None):
        """Up to MAX_BATCH (128) moving frames in one batched Gauss-Newton (mi_aligner_estimate_batch).
        -> (M n x 2 x 3 float64, cc n float64 [-2 where the method failed], iterations n int32)"""
        n = len(dev_ptrs)
        ptrs = (C.c_void_p * n)(*dev_ptrs)
        m = (C.c_double * (6 * n))()
        cc = (C.c_double * n)()
        it = (C.c_int * n)()
        check(load().mi_aligner_estimate_batch(self._h, stream, ptrs, n, int(max_iters), float(eps), m, cc, it))
        return (np.array(list(m), dtype=np.float64).reshape(n, 2, 3), np.array(list(cc), dtype=np.float64),
                np.array(list(it), dtype=np.int32))

    def estimate_pairs(self, dev_ptrs, ref_of, max_iters=60, eps=1e-9, stream=None):
        """Every frame against another frame of the batch: ref_of[k] = index (into dev_ptrs) of frame k's reference
        (mi_aligner_estimate_pairs).  -> (M n x 2 x 3, frame k -> frame ref_of[k]; cc n; iterations n)"""
        n = len(dev_ptrs)
        ptrs = (C.c_void_p * n)(*dev_ptrs)
        refs = (C.c_int * n)(*[int(r) for r in ref_of])
        m = (C.c_double * (6 * n))()
        cc = (C.c_double * n)()
        it = (C.c_int * n)()
        check(load().mi_aligner_estimate_pairs(self._h, stream, ptrs, n, refs, int(max_iters), float(eps), m, cc, it))
        return (np.array(list(m), dtype=np.float64).reshape(n, 2, 3), np.array(list(cc), dtype=np.float64),
                np.array(list(it), dtype=np.int32))

    def refine_batch(self, dev_ptrs, m_init, levels=2, max_iters=20, eps=1e-9, stream=None):
        """The iteration started from the transforms `m_init` (n x 2 x 3, moving -> reference, full-resolution pixels) on the
        `levels` finest pyramid levels (mi_aligner_refine_batch).  -> as estimate_batch"""
        n = len(dev_ptrs)
        ptrs = (C.c_void_p * n)(*dev_ptrs)
        m0 = (C.c_double * (6 * n))(*np.asarray(m_init, dtype=np.float64).reshape(-1))
        m = (C.c_double * (6 * n))()
        cc = (C.c_double * n)()
        it = (C.c_int * n)()
        check(load().mi_aligner_refine_batch(self._h, stream, ptrs, n, m0, int(levels), int(max_iters), float(eps), m, cc, it))
        return (np.array(list(m), dtype=np.float64).reshape(n, 2, 3), np.array(list(cc), dtype=np.float64),
                np.array(list(it), dtype=np.int32))

    def estimate_homography_batch(self, dev_ptrs, max_iters=60, eps=1e-9, stream=None):
        """ALIGN_HOMOGRAPHY: the same estimate refined to 8 degrees of freedom (mi_aligner_estimate_homography_batch).
        -> (M n x 3 x 3 float64 with M[2, 2] = 1, cc n float64 [-2 where the method failed], iterations n int32)"""
        n = len(dev_ptrs)
        ptrs = (C.c_void_p * n)(*dev_ptrs)
        m = (C.c_double * (9 * n))()
        cc = (C.c_double * n)()
        it = (C.c_int * n)()
        check(load().mi_aligner_estimate_homography_batch(self._h, stream, ptrs, n, int(max_iters), float(eps), m, cc, it))
        return (np.array(list(m), dtype=np.float64).reshape(n, 3, 3), np.array(list(cc), dtype=np.float64),
                np.array(list(it), dtype=np.int32))

    def close(self):
        if self._h:
            load().mi_aligner_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


HIST_BGR, HIST_LUMI = 0, 1


def histogram(img, mode=HIST_BGR, subsample=1, fast=True, mask_size=0.0, device=0):
    """Histogram(s) of an H x W x 3 uint8/uint16 BGR image as balance.py:158-180 takes them
    (mi_histogram): int64 array [3][nbins] (B, G, R) or [1][nbins] (luminance)."""
    require_device()
    a = np.ascontiguousarray(img)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("histogram expects an H x W x 3 uint8/uint16 image")
    nbins = 256 if a.dtype == np.uint8 else 65536
    out = np.zeros((3 if mode == HIST_BGR else 1, nbins), np.int64)
    check(load().mi_histogram(device, a.ctypes.data, a.shape[0], a.shape[1], DTYPE_CODE[a.dtype], int(mode),
                              int(subsample), int(bool(fast)), float(mask_size), out.ctypes.data))
    return out


CVT_BGR2HSV, CVT_HSV2BGR, CVT_BGR2HLS, CVT_HLS2BGR = range(4)


def cvt_color(img, code, device=0):
    """cv2.cvtColor(BGR <-> HSV / HLS) of an H x W x 3 uint8 image on the GPU (mi_cvt_color)."""
    require_device()
    a = np.ascontiguousarray(img)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("cvt_color expects an H x W x 3 image")
    if a.dtype not in DTYPE_CODE:
        raise ValueError(f"unsupported dtype {a.dtype}")
    out = np.empty_like(a)
    check(load().mi_cvt_color(device, a.ctypes.data, out.ctypes.data, a.shape[0], a.shape[1], DTYPE_CODE[a.dtype], int(code)))
    return out


def apply_lut(img, luts, device=0):
    """dst[..., c] = luts[0 if one table else c][img[..., c]] on the GPU (mi_apply_lut)."""
    require_device()
    a = np.ascontiguousarray(img)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("apply_lut expects an H x W x 3 uint8/uint16 image")
    nbins = 256 if a.dtype == np.uint8 else 65536
    t = np.ascontiguousarray(np.asarray(luts, dtype=a.dtype).reshape(-1, nbins))
    if t.shape[0] not in (1, 3):
        raise ValueError("one look-up table, or one per channel")
    out = np.empty_like(a)
    check(load().mi_apply_lut(device, a.ctypes.data, out.ctypes.data, a.shape[0], a.shape[1],
                              DTYPE_CODE[a.dtype], t.ctypes.data, t.shape[0]))
    return out
