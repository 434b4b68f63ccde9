"""DepthMapStack on MI355X: the second stacker plug-in behind FocusStack / FocusStackBunch
(SURVEY.md 8(f) rank 4).

Drop-in for the reference's `DepthMapStack` (algorithms/depth_map.py:10-123):

* constructor  DepthMapStack(map_type='average', energy='laplacian', kernel_size=5, blur_size=5,
  smooth_size=15, temperature=0.1, levels=3, float_type='float-32')  (depth_map.py:11-26)
* name() -> 'depth map', steps_per_frame() -> 2, focus_stack(filenames) -> H x W x 3 array of the input
  dtype (:64-123)
* callbacks: 'after_step' (indices 0..2N-1, always -- this stacker has no do_step_callback switch, :74, :113)
  and 'check_running' after every file of both loops; RunStopException when it returns False
* errors: InvalidOptionError for an unknown `energy` after the first loop (:83-87) and for an unknown
  `map_type` before the second (:62-63), ImageLoadError / ShapeError / BitDepthError from the reads

All arithmetic runs in libmi355stack.so (mi_dmap_*, csrc/kernels_depthmap.hpp); there is no CPU path.
Every frame is decoded once and stays on the device for the second loop (the reference reads each file
twice, :69 and :96).

Both float types (float-64: gray / energy planes and blend pyramids in float64; the bilateral filter still runs
on float32 copies, as in the reference).  The AVERAGE map's value where all energies of a pixel are 0 is undefined
in the reference (np.divide(..., where=...) without out=, :57); it is 0 here.
"""
import numpy as np

from . import _lib
from .defaults import constants
from .errors import InvalidOptionError, RunStopException
from .pyramid import BaseStackAlgo

_MAP_CODE = {constants.DM_MAP_AVERAGE: _lib.DM_MAP_AVERAGE, constants.DM_MAP_MAX: _lib.DM_MAP_MAX}
_ENERGY_CODE = {constants.DM_ENERGY_LAPLACIAN: _lib.DM_ENERGY_LAPLACIAN,
                constants.DM_ENERGY_SOBEL: _lib.DM_ENERGY_SOBEL}


class DepthMapStack(BaseStackAlgo):
    def __init__(self, map_type=constants.DEFAULT_DM_MAP, energy=constants.DEFAULT_DM_ENERGY,
                 kernel_size=constants.DEFAULT_DM_KERNEL_SIZE, blur_size=constants.DEFAULT_DM_BLUR_SIZE,
                 smooth_size=constants.DEFAULT_DM_SMOOTH_SIZE, temperature=constants.DEFAULT_DM_TEMPERATURE,
                 levels=constants.DEFAULT_DM_LEVELS, float_type=constants.DEFAULT_DM_FLOAT, *, device=0,
                 decode_threads=8):
        super().__init__("depth map", 2, float_type)
        self.map_type = map_type
        self.energy = energy
        self.kernel_size = kernel_size
        self.blur_size = blur_size
        self.smooth_size = smooth_size
        self.temperature = temperature
        self.levels = levels
        self.device = device
        self.decode_threads = decode_threads
        self._dmap = None
        self._key = None

    # ------------------------------------------------------------------ device handle
    def _handle(self, shape, dtype):
        key = (tuple(shape[:2]), np.dtype(dtype), self.map_type, self.energy, self.kernel_size, self.blur_size,
               self.smooth_size, self.temperature, self.levels, self.float_type)
        if self._dmap is not None and self._key == key:
            self._dmap.reset()
            return self._dmap
        self.close()
        # an unknown option must surface where the reference raises it, not here: placeholders
        self._dmap = _lib.DepthMap(shape[0], shape[1], dtype=dtype,
                                   map_type=_MAP_CODE.get(self.map_type, _lib.DM_MAP_AVERAGE),
                                   energy=_ENERGY_CODE.get(self.energy, _lib.DM_ENERGY_LAPLACIAN),
                                   kernel_size=self.kernel_size, blur_size=self.blur_size,
                                   smooth_size=self.smooth_size, temperature=self.temperature,
                                   levels=self.levels, device=self.device,
                                   float_type=_lib.MI_F64 if self.float_type is np.float64 else _lib.MI_F32)
        self._key = key
        return self._dmap

    def close(self):
        if self._dmap is not None:
            self._dmap.close()
            self._dmap = None

    def _step(self, i):
        self.process.callback('after_step', self.process.id, self.process.name, i)
        if self.process.callback('check_running', self.process.id, self.process.name) is False:
            raise RunStopException(self.name())

    def _check_energy(self):
        if self.energy not in _ENERGY_CODE:   # depth_map.py:83-87
            raise InvalidOptionError('energy', self.energy, details=" valid values are "
                                     f"{constants.DM_ENERGY_SOBEL} and {constants.DM_ENERGY_LAPLACIAN}.")

    def _check_map(self):
        if self.map_type not in _MAP_CODE:    # depth_map.py:62-63
            raise InvalidOptionError("map_type", self.map_type, details=" valid values are "
                                     f"{constants.DM_MAP_AVERAGE} and {constants.DM_MAP_MAX}.")

    # ------------------------------------------------------------------ the steps, one at a time (depth_map.py:28-62)
    # The reference's public methods of the same names; its tests call them (tests/test_0061_depth_map.py:30-41).  Same
    # kernels as focus_stack, run on host arrays: `gray_images` / `energies` are (n, H, W) arrays as the reference passes them.
    def _step_handle(self, planes):
        planes = np.asarray(planes)
        if planes.ndim != 3:
            raise ValueError("expected an array of shape (n, H, W)")
        return self._handle(planes.shape[1:], np.uint8), planes

    def get_sobel_map(self, gray_images):
        """depth_map.py:28-34: |Sobel_x| + |Sobel_y| (ksize 3) of every gray plane, in `float_type`."""
        saved, self.energy = self.energy, constants.DM_ENERGY_SOBEL
        try:
            d, g = self._step_handle(gray_images)
            return d.planes(0, g.astype(self.float_type), self.float_type)
        finally:
            self.energy = saved

    def get_laplacian_map(self, gray_images):
        """depth_map.py:36-41: |Laplacian(GaussianBlur(gray, blur_size), kernel_size)|, in `float_type`."""
        saved, self.energy = self.energy, constants.DM_ENERGY_LAPLACIAN
        try:
            d, g = self._step_handle(gray_images)
            return d.planes(1, g.astype(self.float_type), self.float_type)
        finally:
            self.energy = saved

    def smooth_energy(self, energy_map):
        """depth_map.py:43-52: cv2.bilateralFilter(energy, smooth_size, 25, 25) plane by plane; float32 out (the reference
        fills a float32 array), the input unchanged when smooth_size <= 0."""
        if self.smooth_size <= 0:
            return energy_map
        d, e = self._step_handle(energy_map)
        return d.planes(2, e.astype(self.float_type), np.float32)

    def get_focus_map(self, energies):
        """depth_map.py:54-62: AVERAGE e / sum(e) (0 where the sum is 0: the reference leaves those undefined), MAX the
        softmax of (e - max) / temperature; float32 unless a float-64 stacker runs without smoothing."""
        self._check_map()
        d, e = self._step_handle(energies)
        wt = np.float32 if (self.smooth_size > 0 or self.float_type is np.float32) else np.float64
        return d.planes(3, e.astype(wt), wt)

    # ------------------------------------------------------------------ the stacker
    def focus_stack(self, filenames):
        """depth_map.py:64-123.  `filenames`: sorted list of image paths."""
        _lib.require_device()
        n = len(filenames)
        metadata = None
        dmap = None
        for i, (img_path, decoded) in enumerate(self._decode_ahead(filenames)):
            self.print_message(f": reading file (1/2) {img_path.split('/')[-1]}")
            img, metadata, updated = self.read_image_and_update_metadata(
                img_path, metadata, decoded.result() if decoded is not None else None)
            if updated:
                dmap = self._handle(metadata[0], metadata[1])
            dmap.push_frame(img)
            self._step(i)
        self._check_energy()
        self._check_map()
        # the second loop of the reference (:94-115) re-reads and blends; that work is one device call
        # here, only the progress protocol remains
        for i, img_path in enumerate(filenames):
            self.print_message(f": reading file (2/2) {img_path.split('/')[-1]}")
            self._step(i + n)
        self.print_message(': blend levels')
        return dmap.finish()

    def focus_stack_arrays(self, frames):
        """In-memory variant (no file I/O): `frames` is a sequence of H x W x 3 uint8 / uint16 BGR arrays."""
        from .imageio import get_img_metadata, validate_image
        _lib.require_device()
        if len(frames) == 0:
            raise ValueError("no frames")
        meta = get_img_metadata(frames[0])
        dmap = self._handle(meta[0], meta[1])
        n = len(frames)
        for i, fr in enumerate(frames):
            validate_image(fr, *meta)
            dmap.push_frame(fr)
            if self.process is not None:
                self._step(i)
        self._check_energy()
        self._check_map()
        if self.process is not None:
            for i in range(n):
                self._step(i + n)
        return dmap.finish()
