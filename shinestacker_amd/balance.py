"""BalanceFrames on the MI355X path (reference algorithms/balance.py).

Per frame the reference takes a histogram of the (sub-sampled, optionally circular-masked) image
-- of the luminance (LUMI) or of each colour channel (RGB) -- derives a correction that brings it
to the reference frame's (LINEAR: ratio of histogram means; GAMMA: bisect on the gamma that matches
the means; MATCH_HIST: interpolate the cumulative histograms), turns it into a look-up table and
applies the table to the full-resolution frame.

Here the two data-parallel steps run on the GPU (`mi_histogram`, `mi_apply_lut`:
csrc/kernels_balance.hpp); the table itself (256 or 65536 entries) is computed on the host with the
same NumPy/SciPy calls the reference makes (`np.average(weights=)`, `scipy.optimize.bisect`,
`scipy.interpolate.interp1d`), so tables are identical to the reference's given the same histogram
(tests/golden/balance.npz, recorded from the reference's own classes).

Same class and constructor names as the reference for the pieces a project touches
(`BalanceFrames(enabled=True, mask_size=, intensity_interval=, subsample=, fast_subsampling=,
corr_map=, channel=, plot_summary=, plot_histograms=)`), same sub-action protocol
(begin(process) / run_frame(idx, ref_idx, image) / end()).  Not implemented: the HSV / HLS channel
modes (OpenCV's 8-bit hue arithmetic is not restated) and the matplotlib plots -- both raise
InvalidOptionError / are skipped, stated in DESIGN.md.
"""
import numpy as np

from . import _lib
from .actions import SubAction
from .defaults import constants
from .errors import InvalidOptionError
from .imageio import read_img


def _pixel_range(dtype):
    return constants.NUM_UINT8 if np.dtype(dtype) == np.uint8 else constants.NUM_UINT16


# ---------------------------------------------------------------- correction maps (balance.py:14-120)
class _MeanMap:
    """LINEAR / GAMMA: match the histogram mean inside the intensity interval (balance.py:87-120)."""

    def __init__(self, dtype, ref_hist, intensity_interval=None):
        iv = {'min': 0, 'max': -1, **(intensity_interval or {})}
        self.dtype = np.dtype(dtype)
        self.n = _pixel_range(dtype)
        self.vmax = self.n - 1
        self.ramp = np.array(list(range(self.n)))
        self.lo = iv['min']
        self.hi = iv['max'] + 1 if iv['max'] >= 0 else self.n
        self.reference = [self.mean_of(self.ramp, h) for h in ref_hist]

    def mean_of(self, table, hist):
        return np.average(table[self.lo:self.hi], weights=hist.flatten()[self.lo:self.hi])

    def table(self, correction, _reference=None):
        raise NotImplementedError

    def correction_size(self, correction):
        return correction


class LinearMap(_MeanMap):
    def correction(self, hist):
        return [r / self.mean_of(self.ramp, h) for h, r in zip(hist, self.reference)]

    def table(self, correction, _reference=None):
        ar = np.arange(0, self.n)
        return np.clip(ar * correction, 0, self.vmax).astype(self.dtype)


class GammaMap(_MeanMap):
    def correction(self, hist):
        from scipy.optimize import bisect
        return [bisect(lambda g, h=h, r=r: self.mean_of(self.table(g), h) - r, 0.1, 5)
                for h, r in zip(hist, self.reference)]

    def table(self, correction, _reference=None):
        ar = np.arange(0, self.n)
        return (((ar / self.vmax) ** (1.0 / correction)) * self.vmax).astype(self.dtype)


class MatchHist:
    """MATCH_HIST: map each level to the reference level of the same cumulative count
    (balance.py:53-84)."""

    def __init__(self, dtype, ref_hist, intensity_interval=None):
        self.dtype = np.dtype(dtype)
        self.n = _pixel_range(dtype)
        self.vmax = self.n - 1
        self.ramp = np.array(list(range(self.n)))
        self.levels = [*range(self.n)]
        self.reference = self.cumulative(ref_hist)
        self.reference_mean = [r.mean() for r in self.reference]

    def cumulative(self, hist):
        return [np.cumsum(h) / h.sum() * self.vmax for h in hist]

    def correction(self, hist):
        return self.cumulative(hist)

    def correction_size(self, correction):
        return [c.mean() / m for c, m in zip(correction, self.reference_mean)]

    def table(self, correction, reference):
        from scipy.interpolate import interp1d
        inv = interp1d(reference, self.levels)
        t = np.array(inv(np.clip(correction, reference.min(), reference.max())), dtype=np.float64)
        first, last = t[0], t[-1]
        inner = t[(t != first) & (t != last)]
        if inner.size > 0:
            # the flat runs at both ends (levels the reference never reaches) are spread linearly
            # from 0 up to the first inner value and from the last inner value up to the maximum
            lo, hi = inner.min(), inner.max()
            head = self.ramp[t == first]
            tail = self.ramp[t == last]
            head_max = head.max()
            t[t == first] = (head / head_max * lo) if head_max > 0 else 0
            t[t == last] = tail + (tail - self.vmax) * (self.vmax - hi) / float(tail.size) \
                if tail.size > 0 else self.vmax
        return t.astype(self.dtype)


_MAPS = {constants.BALANCE_LINEAR: LinearMap, constants.BALANCE_GAMMA: GammaMap,
         constants.BALANCE_MATCH_HIST: MatchHist}


# ---------------------------------------------------------------- per-channel-set corrections
class Correction:
    """balance.py:123-201 with the histogram and the table apply on the GPU."""

    hist_mode = None   # _lib.HIST_LUMI or _lib.HIST_BGR
    channels = 0

    def __init__(self, mask_size=0, intensity_interval=None, subsample=-1,
                 fast_subsampling=constants.DEFAULT_BALANCE_FAST_SUBSAMPLING,
                 corr_map=constants.DEFAULT_CORR_MAP, plot_histograms=False, plot_summary=False,
                 device=0):
        self.mask_size = mask_size
        self.intensity_interval = intensity_interval
        self.subsample = constants.DEFAULT_BALANCE_SUBSAMPLE if subsample == -1 else subsample
        self.fast_subsampling = fast_subsampling
        self.corr_map = corr_map
        self.plot_histograms = plot_histograms   # plots are not produced on this path
        self.plot_summary = plot_summary
        self.device = device
        self.dtype = None
        self.corrections = None
        self.process = None

    def get_hist(self, image, _idx=None):
        h = _lib.histogram(image, self.hist_mode, self.subsample, self.fast_subsampling,
                           self.mask_size, self.device)
        return [h[c] for c in range(h.shape[0])]

    def begin(self, ref_image, size, ref_idx):
        self.dtype = ref_image.dtype
        if self.corr_map not in _MAPS:
            raise InvalidOptionError("corr_map", self.corr_map)
        self.corr_map = _MAPS[self.corr_map](self.dtype, self.get_hist(ref_image, ref_idx),
                                             self.intensity_interval)
        self.corrections = np.ones((size, self.channels))

    def tables(self, correction):
        m = self.corr_map
        return [m.table(correction[c], m.reference[c]) for c in range(self.channels)]

    def apply_correction(self, idx, image):
        correction = self.corr_map.correction(self.get_hist(image, idx))
        out = _lib.apply_lut(image, self.tables(correction), self.device)
        self.corrections[idx] = self.corr_map.correction_size(correction)
        return out

    def end(self, _ref_idx):
        pass

    # -- frames resident in HBM (pipeline.align_and_stack_device): same steps on device pointers
    def begin_device(self, dev_ref, height, width, dtype, size):
        """`begin` for a reference frame that lives on the device."""
        self.dtype = np.dtype(dtype)
        self._shape = (height, width)
        nbins = _pixel_range(dtype)
        self._dev_corr = None
        self._scratch = _lib.DeviceBuffer(3 * nbins * 4, self.device)
        self._dev_lut = _lib.DeviceBuffer(3 * nbins * self.dtype.itemsize, self.device)
        if self.corr_map not in _MAPS:
            raise InvalidOptionError("corr_map", self.corr_map)
        self.corr_map = _MAPS[self.corr_map](self.dtype, self.hist_device(dev_ref), self.intensity_interval)
        self.corrections = np.ones((size, self.channels))

    def hist_device(self, dev_img, stream=None):
        import ctypes as C
        nbins = _pixel_range(self.dtype)
        out = np.zeros((self.channels, nbins), np.int64)
        _lib.check(_lib.load().mi_histogram_device(
            self.device, stream, dev_img, self._scratch.ptr, self._shape[0], self._shape[1],
            _lib.DTYPE_CODE[self.dtype], self.hist_mode, int(self.subsample), int(bool(self.fast_subsampling)),
            C.c_double(float(self.mask_size)), out.ctypes.data))
        return [out[c] for c in range(self.channels)]

    def _linear_device(self, idx, dev_img, stream, first_channel=0):
        """LINEAR map: histogram -> table -> apply entirely on the device (mi_balance_linear_device), nothing waits for the
        host; the correction factors land in a device array and are fetched by fetch_corrections()."""
        import ctypes as C
        m = self.corr_map
        ncorr = len(m.reference)
        if getattr(self, "_dev_corr", None) is None:
            self._dev_corr = _lib.DeviceBuffer(8 * self.corrections.size, self.device)
            self._corr_pending = []
        ref = (C.c_double * ncorr)(*[float(r) for r in m.reference])
        _lib.check(_lib.load().mi_balance_linear_device(
            self.device, stream, dev_img, self._scratch.ptr, self._dev_lut.ptr, self._shape[0], self._shape[1],
            _lib.DTYPE_CODE[self.dtype], self.hist_mode, int(self.subsample), int(bool(self.fast_subsampling)),
            C.c_double(float(self.mask_size)), int(m.lo), int(min(m.hi, m.n)), int(first_channel), ref,
            self._dev_corr.ptr + 8 * idx * self.corrections.shape[1]))
        self._corr_pending.append(idx)

    def supports_native_linear(self):
        """True when the map is LINEAR (before or after begin): the only one mi_align_stack_device balances itself"""
        return self.corr_map == constants.BALANCE_LINEAR or isinstance(self.corr_map, LinearMap)

    def native_linear_opts(self, first_channel=0, cvt_to=-1, cvt_from=-1):
        """mi_balance_linear_opts_t for mi_align_stack_device (the LINEAR map only; None otherwise): the aligned frames are
        balanced inside the library's frame loop, the correction factors land in the device array `fetch_corrections` reads."""
        import ctypes as C
        if not isinstance(self.corr_map, LinearMap):
            return None
        m = self.corr_map
        ncorr = len(m.reference)
        if getattr(self, "_dev_corr", None) is None:
            self._dev_corr = _lib.DeviceBuffer(8 * self.corrections.size, self.device)
            self._corr_pending = []
        o = _lib.BalanceLinearOpts(mode=self.hist_mode, subsample=int(self.subsample), fast=int(bool(self.fast_subsampling)),
                                   mask_size=float(self.mask_size), lo=int(m.lo), hi=int(min(m.hi, m.n)),
                                   first_channel=int(first_channel), cvt_to=int(cvt_to), cvt_from=int(cvt_from),
                                   ref_means=(C.c_double * 3)(*([float(r) for r in m.reference] + [0.0] * (3 - ncorr))),
                                   dev_hist_scratch=self._scratch.ptr, dev_lut=self._dev_lut.ptr,
                                   dev_corr_out=self._dev_corr.ptr, ncorr=int(self.corrections.shape[1]))
        return o

    def fetch_corrections(self):
        """Correction factors of the frames balanced by the device-only LINEAR path (synchronises the device)."""
        if getattr(self, "_corr_pending", None):
            _lib.check(_lib.load().mi_device_synchronize(self.device))
            host = self._dev_corr.download(self.corrections.shape, np.float64)
            for i in self._corr_pending:
                self.corrections[i] = host[i]
            self._corr_pending = []
        return self.corrections

    def hist_device_batch(self, dev_imgs, stream=None):
        """histograms of several device frames behind ONE synchronisation (mi_histogram_device_batch)"""
        import ctypes as C
        nbins, n = _pixel_range(self.dtype), len(dev_imgs)
        if getattr(self, "_scratch_n", 1) < n:
            _lib.check(_lib.load().mi_device_synchronize(self.device))   # nothing enqueued still uses the old buffer
            self._scratch.free()
            self._scratch = _lib.DeviceBuffer(3 * nbins * 4 * n, self.device)
            self._scratch_n = n
        out = np.zeros((n, self.channels, nbins), np.int64)
        ptrs = (C.c_void_p * n)(*dev_imgs)
        _lib.check(_lib.load().mi_histogram_device_batch(
            self.device, stream, ptrs, n, self._scratch.ptr, self._shape[0], self._shape[1], _lib.DTYPE_CODE[self.dtype],
            self.hist_mode, int(self.subsample), int(bool(self.fast_subsampling)), C.c_double(float(self.mask_size)),
            out.ctypes.data))
        return [[out[k, c] for c in range(self.channels)] for k in range(n)]

    def apply_correction_device_batch(self, indices, dev_imgs, stream=None):
        """Balance several device frames in place with one host round trip for all of them (LINEAR: none at all): the
        histograms of the whole batch come back together, the tables are built on the host as the reference builds them
        (balance.py:53-120) and applied frame by frame on the stream."""
        if isinstance(self.corr_map, LinearMap):
            for i, p in zip(indices, dev_imgs):
                self._linear_device(i, p, stream)
            return
        hists = self.hist_device_batch(list(dev_imgs), stream)
        nb = _pixel_range(self.dtype)
        if getattr(self, "_dev_lut_n", 1) < len(indices):   # (the batch histogram above synchronised the stream)
            self._dev_lut.free()
            self._dev_lut = _lib.DeviceBuffer(3 * nb * self.dtype.itemsize * len(indices), self.device)
            self._dev_lut_n = len(indices)
        tabs = []
        for i, h in zip(indices, hists):
            correction = self.corr_map.correction(h)
            tabs.append(np.ascontiguousarray(np.stack(self.tables(correction)).astype(self.dtype)))
            self.corrections[i] = self.corr_map.correction_size(correction)
        allt = np.ascontiguousarray(np.stack(tabs))
        self._dev_lut.upload(allt)
        per = allt[0].nbytes
        for k, p in enumerate(dev_imgs):
            _lib.check(_lib.load().mi_apply_lut_device(self.device, stream, p, p, self._shape[0] * self._shape[1],
                                                       _lib.DTYPE_CODE[self.dtype], self._dev_lut.ptr + k * per, allt.shape[1]))

    def apply_correction_device(self, idx, dev_img, stream=None):
        """Balance the device frame in place."""
        if isinstance(self.corr_map, LinearMap):
            return self._linear_device(idx, dev_img, stream)
        correction = self.corr_map.correction(self.hist_device(dev_img, stream))
        t = np.ascontiguousarray(np.stack(self.tables(correction)).astype(self.dtype))
        self._dev_lut.upload(t)
        _lib.check(_lib.load().mi_apply_lut_device(self.device, stream, dev_img, dev_img,
                                                   self._shape[0] * self._shape[1],
                                                   _lib.DTYPE_CODE[self.dtype], self._dev_lut.ptr, t.shape[0]))
        self.corrections[idx] = self.corr_map.correction_size(correction)


class LumiCorrection(Correction):
    """One table from the luminance histogram, applied to B, G and R (balance.py:231-262, :34-36)."""
    hist_mode = _lib.HIST_LUMI
    channels = 1


class RGBCorrection(Correction):
    """One table per colour channel (balance.py:264-296, :44-50)."""
    hist_mode = _lib.HIST_BGR
    channels = 3


class Ch2Correction(Correction):
    """balance.py:304-338: the frame goes to a hue-based colour space, the two non-hue channels are balanced (one table
    each, from the histograms of those channels), the hue channel passes through, and the frame comes back to BGR.
    8-bit frames only, like cv2.cvtColor's HSV / HLS conversions."""
    hist_mode = _lib.HIST_BGR     # per-channel histograms of the converted image; channel 0 (hue) is dropped
    channels = 2
    to_code = from_code = None

    def _need_u8(self, dtype):
        if np.dtype(dtype) != np.uint8:
            raise InvalidOptionError("channel", type(self).__name__,
                                     "HSV / HLS balancing works on 8-bit frames (cv2.cvtColor has no 16-bit form)")

    def preprocess(self, image):
        self._need_u8(image.dtype)
        return _lib.cvt_color(image, self.to_code, self.device)

    def postprocess(self, image):
        return _lib.cvt_color(image, self.from_code, self.device)

    def get_hist(self, image, _idx=None):
        return Correction.get_hist(self, image, _idx)[1:]

    def begin(self, ref_image, size, ref_idx):
        Correction.begin(self, self.preprocess(ref_image), size, ref_idx)

    def tables(self, correction):
        m = self.corr_map
        ident = np.arange(_pixel_range(self.dtype)).astype(self.dtype)
        return [ident] + [m.table(correction[c], m.reference[c]) for c in range(2)]

    def apply_correction(self, idx, image):
        return self.postprocess(Correction.apply_correction(self, idx, self.preprocess(image)))

    # -- frames resident in HBM
    def begin_device(self, dev_ref, height, width, dtype, size):
        self._need_u8(dtype)
        self._cvt = _lib.DeviceBuffer(height * width * 3, self.device)
        _lib.check(_lib.load().mi_cvt_color_device(self.device, None, dev_ref, self._cvt.ptr, height * width,
                                                   _lib.MI_U8, self.to_code))
        Correction.begin_device(self, self._cvt.ptr, height, width, dtype, size)

    def hist_device(self, dev_img, stream=None):
        self.channels = 3
        try:
            return Correction.hist_device(self, dev_img, stream)[1:]
        finally:
            self.channels = 2

    def native_linear_opts(self, first_channel=1, cvt_to=-1, cvt_from=-1):
        return Correction.native_linear_opts(self, 1, self.to_code, self.from_code)

    def hist_device_batch(self, dev_imgs, stream=None):
        self.channels = 3
        try:
            return [h[1:] for h in Correction.hist_device_batch(self, dev_imgs, stream)]
        finally:
            self.channels = 2

    def apply_correction_device_batch(self, indices, dev_imgs, stream=None):
        lib, n = _lib.load(), self._shape[0] * self._shape[1]
        for p in dev_imgs:
            _lib.check(lib.mi_cvt_color_device(self.device, stream, p, p, n, _lib.MI_U8, self.to_code))
        if isinstance(self.corr_map, LinearMap):
            for i, p in zip(indices, dev_imgs):
                self._linear_device(i, p, stream, first_channel=1)
        else:
            Correction.apply_correction_device_batch(self, indices, dev_imgs, stream)
        for p in dev_imgs:
            _lib.check(lib.mi_cvt_color_device(self.device, stream, p, p, n, _lib.MI_U8, self.from_code))

    def apply_correction_device(self, idx, dev_img, stream=None):
        lib, n = _lib.load(), self._shape[0] * self._shape[1]
        _lib.check(lib.mi_cvt_color_device(self.device, stream, dev_img, dev_img, n, _lib.MI_U8, self.to_code))
        if isinstance(self.corr_map, LinearMap):
            self._linear_device(idx, dev_img, stream, first_channel=1)
        else:
            Correction.apply_correction_device(self, idx, dev_img, stream)
        _lib.check(lib.mi_cvt_color_device(self.device, stream, dev_img, dev_img, n, _lib.MI_U8, self.from_code))


class SVCorrection(Ch2Correction):
    """S and V of HSV (balance.py:340-350)."""
    to_code, from_code = _lib.CVT_BGR2HSV, _lib.CVT_HSV2BGR
    labels = ("H", "S", "V")


class LSCorrection(Ch2Correction):
    """L and S of HLS (balance.py:353-363)."""
    to_code, from_code = _lib.CVT_BGR2HLS, _lib.CVT_HLS2BGR
    labels = ("H", "L", "S")


class BalanceFrames(SubAction):
    """Sub-action of CombinedActions (balance.py:366-416)."""

    def __init__(self, enabled=True, **kwargs):
        super().__init__(enabled=enabled)
        self.process = None
        self.shape = None
        corr_map = kwargs.get('corr_map', constants.DEFAULT_CORR_MAP)
        subsample = kwargs.get('subsample', constants.DEFAULT_BALANCE_SUBSAMPLE)
        self.fast_subsampling = kwargs.get('fast_subsampling', constants.DEFAULT_BALANCE_FAST_SUBSAMPLING)
        channel = kwargs.pop('channel', constants.DEFAULT_CHANNEL)
        if subsample == -1:
            subsample = 1 if corr_map == constants.BALANCE_MATCH_HIST else constants.DEFAULT_BALANCE_SUBSAMPLE
        kwargs['subsample'] = subsample
        self.mask_size = kwargs.get('mask_size', 0)
        self.plot_summary = kwargs.get('plot_summary', False)
        if channel == constants.BALANCE_LUMI:
            self.correction = LumiCorrection(**kwargs)
        elif channel == constants.BALANCE_RGB:
            self.correction = RGBCorrection(**kwargs)
        elif channel == constants.BALANCE_HSV:
            self.correction = SVCorrection(**kwargs)
        elif channel == constants.BALANCE_HLS:
            self.correction = LSCorrection(**kwargs)
        else:
            raise InvalidOptionError("channel", channel)

    def begin(self, process):
        self.process = process
        self.correction.process = process
        img = read_img(self.process.input_full_path + "/" + self.process.filenames[process.ref_idx])
        self.shape = img.shape
        # per-frame tables are indexed by the GLOBAL frame index (a sharded process counts only its own block)
        self.correction.begin(img, len(self.process.filenames), process.ref_idx)

    def end(self):
        self.process.print_message(' ' * 60)
        self.correction.end(self.process.ref_idx)

    def run_frame(self, idx, _ref_idx, image):
        if idx != self.process.ref_idx:
            self.process.sub_message_r(': balance image')
            image = self.correction.apply_correction(idx, image)
        return image
