"""PyramidStack on MI355X: the stacker plug-in behind FocusStack / FocusStackBunch.

Drop-in for the reference's `PyramidStack` (algorithms/pyramid.py:114-179) and its
`BaseStackAlgo` protocol (algorithms/base_stack_algo.py:9-42):

* constructor  PyramidStack(min_size=32, kernel_size=5, gen_kernel=0.4, float_type='float-32')
  (+ keyword-only extensions; `arith="separable"|"exact"`: the evaluation order of the stencils, see __init__)
* set by the owning action: ``.process`` (stack.py:23), ``.do_step_callback`` (stack.py:103 / :76)
* name() -> 'pyramid', steps_per_frame() -> 2, print_message(), focus_stack(filenames) -> H x W x 3
  array of the input dtype, BGR, C-contiguous, host memory
* callbacks: 'after_step' (indices 0..2N-1) and 'check_running' after every file of both
  passes (pyramid.py:166-169, :174-177); RunStopException when 'check_running' returns False
* errors: ImageLoadError / ShapeError / BitDepthError / InvalidOptionError / RuntimeError as the
  reference raises them (base_stack_algo.py:19-22, :33-42; utils.py:12-13, :56-63)

All arithmetic runs in libmi355stack.so (HIP, gfx950).  There is no CPU path: without the
library or without a GPU, focus_stack raises DeviceError.

Differences from the reference that do not change results: every frame is decoded once (the
reference decodes twice, pyramid.py:158 and :172) and is pushed to the device during the
validation pass, and no per-frame pyramid is kept -- selection is a running first-max on the
device.
"""
import logging

import numpy as np

from . import _lib
from .defaults import constants, resolve_arith
from .errors import ImageLoadError, InvalidOptionError, RunStopException
from .imageio import get_img_metadata, read_img, validate_image


def _cyan(msg):
    return f"\033[36m{msg}\033[0m"


class BaseStackAlgo:
    """Plug-in base shared by stackers (reference base_stack_algo.py:9-42)."""

    def __init__(self, name, steps_per_frame, float_type=constants.DEFAULT_PY_FLOAT):
        self._name = name
        self._steps_per_frame = steps_per_frame
        self.process = None
        self.decode_threads = 8   # files are decoded this many at a time, ahead of the GPU (_decode_ahead)
        if float_type == constants.FLOAT_32:
            self.float_type = np.float32
        elif float_type == constants.FLOAT_64:
            self.float_type = np.float64
        else:
            raise InvalidOptionError("float_type", float_type,
                                     details=" valid values are FLOAT_32 and FLOAT_64")

    def name(self):
        return self._name

    def steps_per_frame(self):
        return self._steps_per_frame

    def print_message(self, msg):
        self.process.sub_message_r(_cyan(msg))

    def read_image_and_update_metadata(self, img_path, metadata, img=None):
        """base_stack_algo.py:33-42; `img`: the already decoded file (decode-ahead), else it is read here."""
        if img is None:
            img = read_img(img_path)
        if img is None:
            raise ImageLoadError(img_path)
        updated = metadata is None
        if updated:
            metadata = get_img_metadata(img)
        else:
            validate_image(img, *metadata)
        return img, metadata, updated

    def _decode_ahead(self, filenames):
        """(path, future-or-None) in file order.  With decode_threads > 1 the files are decoded on a
        thread pool a few frames ahead of the consumer (image codecs release the GIL): a 24 MP JPEG
        takes the host far longer to decode than the GPU needs to fuse it.  Errors surface in order,
        at the file that caused them, as in the sequential loop (pyramid.py:155-169)."""
        n = int(self.decode_threads or 0)
        if n <= 1 or len(filenames) <= 1:
            for p in filenames:
                yield p, None
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=n) as pool:
            window = deque()
            it = iter(filenames)
            for p in it:
                window.append((p, pool.submit(read_img, p)))
                if len(window) >= 2 * n:
                    break
            while window:
                p, fut = window.popleft()
                nxt = next(it, None)
                if nxt is not None:
                    window.append((nxt, pool.submit(read_img, nxt)))
                try:
                    yield p, fut
                except GeneratorExit:
                    for _, f in window:
                        f.cancel()
                    raise


class PyramidStack(BaseStackAlgo):
    def __init__(self, min_size=constants.DEFAULT_PY_MIN_SIZE,
                 kernel_size=constants.DEFAULT_PY_KERNEL_SIZE,
                 gen_kernel=constants.DEFAULT_PY_GEN_KERNEL,
                 float_type=constants.DEFAULT_PY_FLOAT, *, device=0, use_fma=True,
                 impl=_lib.IMPL_AUTO, batch_frames=0, decode_threads=8, arith=None):
        super().__init__("pyramid", 2, float_type)
        # arith (keyword-only extension; the reference has one evaluation order):
        #   "separable" = MI_ARITH_SEPARABLE, the 5 + 5 tap form of the same stencils: within the stated float-32 tolerance
        #                 of the reference's float-64 mode, ~1.3x the throughput -- the default since round 4
        #                 (constants.DEFAULT_PY_ARITH; float-64 stacks always run "exact")
        #   "exact"     = the reference's own row-major 25-tap order: bit-identical to its restatement, the audit mode
        # None -> the default (defaults.resolve_arith: the same rule in every high-level entry point).  The arithmetic a
        # stack ran with is `self.arith`, part of `name()`'s sibling `describe()` and of the job log.
        arith = resolve_arith(arith, float_type)
        if arith not in _lib.ARITH_CODE:
            raise InvalidOptionError("arith", arith, details=" valid values are 'exact' and 'separable'")
        if arith == "separable" and float_type == constants.FLOAT_64:
            raise InvalidOptionError("arith", arith, details=" the separable arithmetic is float-32 only")
        self.arith = arith
        self.min_size = min_size
        self.kernel_size = kernel_size
        self.pad_amount = (kernel_size - 1) // 2
        self.gen_kernel_a = gen_kernel
        self.do_step_callback = False
        self.device = device
        self.use_fma = use_fma
        self.impl = impl
        self.batch_frames = batch_frames
        self.decode_threads = decode_threads   # files are decoded this many at a time, ahead of the GPU
        self.dtype = None
        self.num_pixel_values = None
        self.max_pixel_value = None
        self._stack = None

    # ------------------------------------------------------------------ device handle
    def _handle(self, shape, dtype):
        """One handle per (shape, dtype); FocusStackBunch reuses the stacker for every bunch."""
        key = (tuple(shape[:2]), np.dtype(dtype))
        if self._stack is not None and self._stack_key == key:
            self._stack.reset()
            return self._stack
        if self._stack is not None:
            self._stack.close()
        self._stack = _lib.Stack(shape[0], shape[1], in_dtype=dtype, out_dtype=dtype,
                                 min_size=self.min_size, kernel_size=self.kernel_size,
                                 gen_kernel=self.gen_kernel_a, use_fma=self.use_fma,
                                 device=self.device, impl=self.impl,
                                 batch_frames=self.batch_frames, arith=self.arith,
                                 float_type=_lib.MI_F64 if self.float_type is np.float64 else _lib.MI_F32)
        self._stack_key = key
        # which arithmetic this stack runs with goes to the log (not to print_message: the callback trace is the reference's)
        logging.getLogger("shinestacker_amd").info("PyramidStack: %s", self.describe())
        return self._stack

    def describe(self):
        """the options that decide the bits of the result, for logs and output metadata"""
        return (f"arith={self.arith} float_type={'float-64' if self.float_type is np.float64 else 'float-32'} "
                f"min_size={self.min_size} kernel_size={self.kernel_size} gen_kernel={self.gen_kernel_a} use_fma={self.use_fma}")

    def close(self):
        if self._stack is not None:
            self._stack.close()
            self._stack = None

    # ------------------------------------------------------------------ the steps, one at a time (pyramid.py:24-148)
    # The reference's public methods of the same names, on NumPy arrays, run on the GPU.  They always use the reference's
    # own evaluation order (the 25-tap chain) whatever `arith` the stacker fuses with, and float-32 arithmetic.
    def _f32_steps(self):
        if self.float_type is not np.float32:
            raise InvalidOptionError("float_type", "float-64", details=" the step methods run in float-32; focus_stack supports both")

    def _pyr(self, op, img, img2=None, maxv=255.0):
        self._f32_steps()
        return _lib.pyr_step(op, img, img2, gen_kernel=self.gen_kernel_a, use_fma=self.use_fma, maxv=maxv, device=self.device)

    def convolve(self, image):
        """pyramid.py:24-25: cv2.filter2D(image, -1, gen_kernel, BORDER_REFLECT101), 2-D or H x W x C."""
        return self._pyr(_lib.PYR_CONVOLVE, image)

    def reduce_layer(self, layer):
        """pyramid.py:27-32: convolve(layer)[::2, ::2] per channel."""
        return self._pyr(_lib.PYR_REDUCE, layer)

    def expand_layer(self, layer):
        """pyramid.py:34-46: 4 * convolve(zero-stuffed 2h x 2w image) per channel."""
        return self._pyr(_lib.PYR_EXPAND, layer)

    def fuse_laplacian(self, laplacians):
        """pyramid.py:48-55: energy = convolve(gray(lap)^2), first arg-max over the stack, sum of np.where(best == i, lap, 0)."""
        return self._pyr(_lib.PYR_FUSE_LAPLACIAN, np.stack([np.asarray(lap, np.float32) for lap in laplacians]))

    def collapse(self, pyramid):
        """pyramid.py:57-64: from the base up, expand_layer(img)[:h, :w] + layer, then clip(abs(.), 0, max_pixel_value)."""
        img = np.asarray(pyramid[-1], np.float32)
        for layer in pyramid[-2::-1]:
            img = self._pyr(_lib.PYR_COLLAPSE_STEP, layer, img)
        return self._pyr(_lib.PYR_CLIP_ABS, img, maxv=float(self.max_pixel_value))

    def process_single_image(self, img, levels):
        """pyramid.py:125-139: the Laplacian pyramid of one frame, [lap_0, ..., lap_{levels-1}, base] -- produced by the fused
        level kernels themselves: in a stack of ONE frame that frame wins every pixel, so the stack's fused Laplacians are
        the frame's own and its coarsest Gaussian is the base."""
        self._f32_steps()
        img = np.ascontiguousarray(img)
        h, w = img.shape[:2]
        if levels < 1:
            return [img.astype(np.float32)]
        min_size = max(1, int(min(h, w) // (1 << levels)))
        while int(np.log2(min(h, w) / min_size)) > levels:      # int(log2(min / min_size)) must be exactly `levels`
            min_size += 1
        st = _lib.Stack(h, w, in_dtype=img.dtype if img.dtype in (np.uint8, np.uint16) else np.float32,
                        out_dtype=self.dtype if self.dtype is not None else np.uint8, min_size=min_size,
                        kernel_size=self.kernel_size, gen_kernel=self.gen_kernel_a, use_fma=self.use_fma, device=self.device,
                        arith="exact")
        try:
            # (the reference stops early when a level would get a side below 4 pixels, :129-130: so does the library)
            st.push_frame(img if img.dtype in (np.uint8, np.uint16) else img.astype(np.float32))
            out = [st.tap(_lib.TAP_FUSED_LAP, lv) for lv in range(st.levels)]
            out.append(st.tap(_lib.TAP_GAUSS, st.levels))
            return out
        finally:
            st.close()

    def get_fused_base(self, images):
        """pyramid.py:95-111: entropy / deviation rule over the stack of base images (n, h, w, 3) -- a stack without
        Laplacian levels run through the library, whose fused base is tapped before the final clip and cast."""
        self._f32_steps()
        images = np.ascontiguousarray(images, np.float32)
        n, h, w = images.shape[:3]
        out_dtype = self.dtype if self.dtype is not None else np.uint8
        st = _lib.Stack(h, w, in_dtype=np.float32, out_dtype=out_dtype, min_size=max(h, w) + 1, kernel_size=self.kernel_size,
                        gen_kernel=self.gen_kernel_a, use_fma=self.use_fma, device=self.device, arith="exact")
        try:
            assert st.levels == 0
            for i in range(n):
                st.push_frame(images[i])
            st.finish()
            return st.tap(_lib.TAP_FUSED_BASE)
        finally:
            st.close()

    def fuse_pyramids(self, all_laplacians):
        """pyramid.py:141-148: the base by get_fused_base, every other level by fuse_laplacian, coarsest first; returned in
        the order [lap_0, ..., base] the reference returns (fused[::-1])."""
        fused = [self.get_fused_base(np.stack([lap[-1] for lap in all_laplacians], axis=0))]
        for layer in range(len(all_laplacians[0]) - 2, -1, -1):
            if self.process is not None:
                self.print_message(f': fusing pyramids, layer: {layer + 1}')
            fused.append(self.fuse_laplacian(np.stack([lap[layer] for lap in all_laplacians], axis=0)))
        if self.process is not None:
            self.print_message(': pyramids fusion completed')
        return fused[::-1]

    # ------------------------------------------------------------------ callbacks
    def _step(self, i):
        if self.do_step_callback:
            self.process.callback('after_step', self.process.id, self.process.name, i)
        if self.process.callback('check_running', self.process.id, self.process.name) is False:
            raise RunStopException(self.process.name)

    def _set_dtype(self, dtype):
        self.dtype = dtype
        is8 = dtype == np.uint8
        self.num_pixel_values = constants.NUM_UINT8 if is8 else constants.NUM_UINT16
        self.max_pixel_value = constants.MAX_UINT8 if is8 else constants.MAX_UINT16

    # ------------------------------------------------------------------ the hot path
    def focus_stack(self, filenames):
        """pyramid.py:150-179.  `filenames`: sorted list of image paths."""
        _lib.require_device()
        n = len(filenames)
        metadata = None
        stack = None
        # pass 1: decode + validate every file (as pyramid.py:155-169) and hand each
        # frame to the device right away, so decoding overlaps the GPU work.  A
        # validation error still aborts the whole stack before any result exists.
        for i, (img_path, decoded) in enumerate(self._decode_ahead(filenames)):
            self.print_message(f": validating file {img_path.split('/')[-1]}")
            img, metadata, updated = self.read_image_and_update_metadata(
                img_path, metadata, decoded.result() if decoded is not None else None)
            if updated:
                self._set_dtype(metadata[1])
                stack = self._handle(metadata[0], metadata[1])
            stack.push_frame(img)
            self._step(i)
        # pass 2: the reference decodes and builds pyramids here (pyramid.py:170-177);
        # that work is already enqueued, only the progress protocol remains.
        for i, img_path in enumerate(filenames):
            self.print_message(f": processing file {img_path.split('/')[-1]}")
            self._step(i + n)
        self.print_message(': pyramids fusion completed')
        return stack.finish()

    def focus_stack_arrays(self, frames):
        """In-memory variant (no file I/O): `frames` is a sequence of H x W x 3 uint8/uint16
        BGR arrays.  Same callbacks as focus_stack, one step per frame per pass."""
        _lib.require_device()
        n = len(frames)
        if n == 0:
            raise ValueError("no frames")
        meta = get_img_metadata(frames[0])
        self._set_dtype(meta[1])
        for i, fr in enumerate(frames):
            validate_image(fr, *meta)
            if self.process is not None:
                self._step(i)
        stack = self._handle(meta[0], meta[1])
        for i, fr in enumerate(frames):
            stack.push_frame(fr)
            if self.process is not None:
                self._step(i + n)
        return stack.finish()
