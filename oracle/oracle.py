"""oracle/oracle.py -- CPU checker for the HIP focus-stacking path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from shinestacker_amd/.

Two restatements of /root/reference/src/shinestacker/algorithms/pyramid.py:

* ``ref_shaped_*``  -- NumPy code that keeps the reference's *shape*: per-channel
  full-resolution 25-tap filter then ``[::2, ::2]`` (pyramid.py:27-32), zero-stuff
  then ``4*filter`` (:34-46), all frames' Laplacians resident, ``np.argmax`` over
  the frame axis and ``sum(where(best == i, lap_i, 0))`` (:48-55), collapse
  (:57-64), base-level entropy/deviation rule (:66-111), truncating cast (:179).
  Its only non-NumPy ingredients are the three cv2 primitives, taken from
  liboracle.so (pyramid_oracle.c).  gen_golden.py proves it equal, bit for bit,
  to the reference's own pyramid.py run with the same primitives as a cv2 shim.

* ``StreamingOracle`` -- the decimated/polyphase, O(1)-memory C implementation
  (same arithmetic, running first-max).  Fast enough for multi-megapixel checks
  and used as the ``cpu_baseline`` ("port") in bench.py.

PARITY STATUS: the cv2 primitives themselves are "parity unpinned" -- see the
header of pyramid_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("pyramid_oracle.c", "separable_oracle.c", "align_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(map(os.path.getmtime, srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota
    (OpenMP's default of one thread per visible core oversubscribes a quota-limited container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:  # cgroup v2
            quota, period = fh.read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, \
                    open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        n = int(os.environ["OMP_NUM_THREADS"])
    return n


_lib = None
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_num_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads(usable_cpus())
        L.orc_filter2d_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int]
        L.orc_filter2d_f64.argtypes = [_f64p, C.c_int, C.c_int, _f64p, _f64p, C.c_int]
        L.orc_bgr2gray_f32.argtypes = [_f32p, C.c_size_t, _f32p, C.c_int]
        L.orc_reduce_f32.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int]
        L.orc_expand_f32.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int,
                                     _f32p, C.c_int]
        L.orc_level_select_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int,
                                           _f32p, C.c_int, C.c_int, _f32p, _f32p, _i32p, _f32p,
                                           C.c_int]
        L.orc_collapse_level_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int,
                                             C.c_int, _f32p, C.c_int]
        L.orc_finalize_u8.argtypes = [_f32p, C.c_size_t, _u8p]
        L.orc_finalize_u16.argtypes = [_f32p, C.c_size_t, _u16p]
        L.orc_np_sum_f32.argtypes = [_f32p, C.c_int]
        L.orc_np_sum_f32.restype = C.c_float
        L.orc_base_features_f32.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p,
                                            _f32p, C.c_int]
        L.orc_base_select_f32.argtypes = [_f32p, _f32p, C.c_size_t, C.c_int, C.c_int, _f32p,
                                          _f32p, _i32p, _i32p]
        L.orc_base_fuse_f32.argtypes = [_f32p, C.c_size_t, _i32p, _i32p, _f32p]
        L.orc_synth_frame_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.orc_synth_crop_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                        C.c_int, C.c_int]
        L.orc_to_f32_u8.argtypes = [_u8p, C.c_size_t, _f32p]
        L.orc_to_f32_u16.argtypes = [_u16p, C.c_size_t, _f32p]
        L.orc_sep_reduce_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p]
        L.orc_sep_expand_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p]
        L.orc_sep_level_select_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_int,
                                               C.c_int, _f32p, _f32p, _i32p, _f32p]
        L.orc_sep_collapse_level_f32.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.orc_warp_affine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      _f64p, C.c_int, _f64p]
        L.orc_warp_perspective.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           _f64p, C.c_int, _f64p]
        L.orc_border_blur_composite.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_double]
        L.orc_gauss_kernel_f32.argtypes = [C.c_int, C.c_double, _f32p]
        L.orc_gauss_kernel_fixed.argtypes = [C.c_int, C.c_double, C.c_int, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]
        L.orc_invert_affine.argtypes = [_f64p, _f64p]
        _lib = L
    return _lib


# --------------------------------------------------------------------------
# constants of the path
# --------------------------------------------------------------------------
def gen_kernel_1d(a=0.4):
    """pyramid.py:19-20 -- the Burt-Adelson generating kernel, float64."""
    return np.array([0.25 - a / 2.0, 0.25, a, 0.25, 0.25 - a / 2.0])


def gen_kernel_2d(a=0.4):
    """pyramid.py:21 -- np.outer in float64 (what the reference hands cv2)."""
    k = gen_kernel_1d(a)
    return np.outer(k, k)


def k25_f32(a=0.4):
    """float32 taps as OpenCV uses them for a CV_32F image [from memory]."""
    return np.ascontiguousarray(gen_kernel_2d(a).astype(np.float32).ravel())


def k3_f32(a=0.4):
    """MI_ARITH_SEPARABLE: float32 of the 1-D generating kernel (k0, k1, k2), separable_oracle.c."""
    return np.ascontiguousarray(gen_kernel_1d(a)[:3].astype(np.float32))


def red_taps_f32(a=0.4):
    """MI_ARITH_SEPARABLE, taps of the reduce (w0, w1, w2, rs): the integers 20 k and rs = float32(1/400) when 20 k is
    integral, else float32(k) and 1 (separable_oracle.c; the same rule as csrc/common.hpp::red_taps)."""
    k = gen_kernel_1d(a)[:3].astype(np.float64)
    w = 20.0 * np.array([0.25 - a / 2.0, 0.25, a])
    r = np.round(w)      # (the library rounds half away from zero: the two differ on exact halves only, which are not integral)
    # integral taps whose partial sums stay exact for 16-bit input: (2 |w0| + |w2|) * 65535 < 2^24 (w1 = 5 comes last, the one
    # operation that may round); a = 0.4: 1 5 8; a = 0.7: -2 5 14 (a negative outer tap is fine); a = 6.7: -62 5 134 is over the bound
    if np.all(np.abs(w - r) < 1e-9) and 2 * abs(r[0]) + abs(r[2]) <= 255:
        return np.ascontiguousarray(np.array([r[0], 5.0, r[2], 1.0 / 400.0]).astype(np.float32))
    return np.ascontiguousarray(np.array([k[0], k[1], k[2], 1.0]).astype(np.float32))


def num_levels(h, w, min_size=32):
    """pyramid.py:165 (requested levels) + :129-130 (early stop when a side < 4)."""
    req = int(np.log2(min(h, w) / min_size))
    n = 0
    for _ in range(req):
        h, w = (h + 1) // 2, (w + 1) // 2
        if min(h, w) < 4:
            break
        n += 1
    return n


def level_shapes(h, w, levels):
    out = [(h, w)]
    for _ in range(levels):
        h, w = (h + 1) // 2, (w + 1) // 2
        out.append((h, w))
    return out


# --------------------------------------------------------------------------
# cv2 primitives (the three functions the reference's pyramid path calls)
# --------------------------------------------------------------------------
def filter2D(img, kernel2d, use_fma=True):
    """cv2.filter2D(img, -1, kernel, borderType=BORDER_REFLECT101), 2-D single channel."""
    assert img.ndim == 2
    h, w = img.shape
    if img.dtype == np.float32:
        src = np.ascontiguousarray(img)
        dst = np.empty_like(src)
        k = np.ascontiguousarray(np.asarray(kernel2d).astype(np.float32).ravel())
        lib().orc_filter2d_f32(src, h, w, k, dst, int(use_fma))
        return dst
    if img.dtype == np.float64:
        src = np.ascontiguousarray(img)
        dst = np.empty_like(src)
        k = np.ascontiguousarray(np.asarray(kernel2d, dtype=np.float64).ravel())
        lib().orc_filter2d_f64(src, h, w, k, dst, int(use_fma))
        return dst
    raise TypeError(f"filter2D oracle: unsupported dtype {img.dtype}")


def bgr2gray_f32(img, use_fma=True):
    """cv2.cvtColor(img.astype(float32), COLOR_BGR2GRAY)."""
    assert img.dtype == np.float32 and img.shape[-1] == 3
    src = np.ascontiguousarray(img)
    out = np.empty(img.shape[:-1], np.float32)
    lib().orc_bgr2gray_f32(src, out.size, out, int(use_fma))
    return out


def pad_reflect101(img, pad):
    """cv2.copyMakeBorder(img, pad, pad, pad, pad, BORDER_REFLECT101)."""
    return np.pad(img, pad, mode="reflect")


# --------------------------------------------------------------------------
# reference-shaped restatement
# --------------------------------------------------------------------------
class RefShaped:
    """Keeps the reference's structure; see module docstring."""

    def __init__(self, min_size=32, kernel_size=5, gen_kernel=0.4, float_type=np.float32,
                 use_fma=True):
        self.min_size = min_size
        self.pad = (kernel_size - 1) // 2
        self.k2d = gen_kernel_2d(gen_kernel)
        self.ft = float_type
        self.use_fma = use_fma

    def conv(self, plane):
        return filter2D(plane, self.k2d, self.use_fma)

    def reduce(self, layer):
        chans = [self.conv(np.ascontiguousarray(layer[:, :, c]))[::2, ::2]
                 for c in range(layer.shape[2])]
        return np.stack(chans, axis=-1)

    def expand(self, layer):
        h, w, nc = layer.shape
        out = np.zeros((2 * h, 2 * w, nc), layer.dtype)
        for c in range(nc):
            z = np.zeros((2 * h, 2 * w), layer.dtype)
            z[::2, ::2] = layer[:, :, c]
            out[:, :, c] = 4.0 * self.conv(z)
        return out

    def laplacian_pyramid(self, img, levels):
        gauss = [img.astype(self.ft)]
        for _ in range(levels):
            nxt = self.reduce(gauss[-1])
            if min(nxt.shape[:2]) < 4:
                break
            gauss.append(nxt)
        pyr = []
        for lev in range(len(gauss) - 1):
            h, w = gauss[lev].shape[:2]
            pyr.append(gauss[lev] - self.expand(gauss[lev + 1])[:h, :w])
        pyr.append(gauss[-1])
        return pyr, gauss

    def energy(self, lap):
        g = bgr2gray_f32(lap.astype(np.float32), self.use_fma)
        return self.conv(np.square(g))

    def fuse_level(self, laps):
        """laps: (N, h, w, 3). Returns fused, best, energies."""
        e = np.stack([self.energy(l) for l in laps])
        best = np.argmax(e, axis=0)
        fused = np.zeros_like(laps[0])
        for i, l in enumerate(laps):
            fused += np.where(best[:, :, None] == i, l, 0)
        return fused, best, e

    # -- base level ------------------------------------------------------
    def base_features(self, base, dtype):
        nlev = 256 if dtype == np.uint8 else 65536
        gray = bgr2gray_f32(base.astype(np.float32), self.use_fma).astype(dtype)
        levels, counts = np.unique(gray, return_counts=True)
        p = np.zeros(nlev, self.ft)
        p[levels] = counts.astype(self.ft) / counts.sum()
        f64 = self.ft == np.float64
        # float-32: correctly rounded float32 log (see pyramid_oracle.c: orc_base_features_f32);
        # float-64: logl rounded to double (NumPy's own log is CPU-dispatch dependent in both widths)
        logp = np.zeros(nlev, self.ft)
        nz = p > 0
        logp[nz] = np.log(p[nz].astype(np.longdouble)).astype(np.float64) if f64 \
            else np.log(p[nz].astype(np.float64)).astype(np.float32)
        pad = self.pad
        padded = pad_reflect101(gray, pad)
        hb, wb = gray.shape
        ent = np.empty((hb, wb), self.ft)
        dev = np.empty((hb, wb), self.ft)
        n = (2 * pad + 1) ** 2
        for y in range(hb):
            for x in range(wb):
                area = padded[y:y + 2 * pad + 1, x:x + 2 * pad + 1]
                lv = area.flatten()
                ent[y, x] = self.ft(-1.0 * (lv * logp[lv]).sum())
                mean = np.average(area).astype(self.ft)
                dev[y, x] = np.square(area - mean).sum() / n
        return ent, dev

    def fuse_base(self, bases, dtype):
        feats = [self.base_features(b, dtype) for b in bases]
        ent = np.stack([f[0] for f in feats])
        dev = np.stack([f[1] for f in feats])
        be, bd = np.argmax(ent, axis=0), np.argmax(dev, axis=0)
        fused = np.zeros(bases[0].shape, self.ft)
        for i, b in enumerate(bases):
            fused += np.where(be[:, :, None] == i, b, 0)
            fused += np.where(bd[:, :, None] == i, b, 0)
        return (fused / 2).astype(bases[0].dtype), be, bd, ent, dev

    def collapse(self, pyr, max_value):
        img = pyr[-1]
        for lap in pyr[-2::-1]:
            up = self.expand(img)[:lap.shape[0], :lap.shape[1]]
            img = up + lap
        return np.clip(np.abs(img), 0, max_value)

    def stack(self, frames, want_detail=False):
        """frames: list of HxWx3 uint8/uint16 arrays (BGR). Returns the fused image."""
        dtype = frames[0].dtype
        maxv = 255 if dtype == np.uint8 else 65535
        h, w = frames[0].shape[:2]
        levels = int(np.log2(min(h, w) / self.min_size))
        pyrs = [self.laplacian_pyramid(f, levels)[0] for f in frames]
        nl = len(pyrs[0]) - 1
        base, be, bd, ent, dev = self.fuse_base([p[-1] for p in pyrs], dtype)
        fused = [None] * nl + [base]
        detail = {"best": [None] * nl, "energy": [None] * nl, "be": be, "bd": bd,
                  "ent": ent, "dev": dev}
        for lev in range(nl - 1, -1, -1):
            fl, best, e = self.fuse_level(np.stack([p[lev] for p in pyrs]))
            fused[lev] = fl
            detail["best"][lev] = best
            detail["energy"][lev] = e.max(axis=0)
        img = self.collapse(fused, maxv)
        out = img.astype(dtype)
        if want_detail:
            detail.update(fused=fused, collapsed=img, pyramids=pyrs)
            return out, detail
        return out


# --------------------------------------------------------------------------
# streaming C restatement
# --------------------------------------------------------------------------
class StreamingOracle:
    """push_frame()/finish() over liboracle.so; O(1) memory in the frame count
    apart from the per-frame base images (tiny)."""

    def __init__(self, h, w, dtype=np.uint8, min_size=32, kernel_size=5, gen_kernel=0.4,
                 use_fma=True, levels=None, keep_gauss=True, arith="exact"):
        assert arith in ("exact", "separable")
        # "separable": the MI_ARITH_SEPARABLE arithmetic (separable_oracle.c) for the pyramid stencils; the base
        # level rule and the final cast are the same in both modes
        self.sep = arith == "separable"
        self.k3 = k3_f32(gen_kernel)
        self.rk = red_taps_f32(gen_kernel)
        self.keep_gauss = keep_gauss
        self.h, self.w, self.dtype = h, w, np.dtype(dtype)
        self.levels = num_levels(h, w, min_size) if levels is None else levels
        self.shapes = level_shapes(h, w, self.levels)
        self.k = k25_f32(gen_kernel)
        self.pad = (kernel_size - 1) // 2
        self.fma = int(use_fma)
        self.n = 0
        L = self.levels
        self.best_e = [np.zeros(s, np.float32) for s in self.shapes[:L]]
        self.best_idx = [np.zeros(s, np.int32) for s in self.shapes[:L]]
        self.best_lap = [np.zeros(s + (3,), np.float32) for s in self.shapes[:L]]
        hb, wb = self.shapes[L]
        self.b_ent = np.zeros((hb, wb), np.float32)
        self.b_dev = np.zeros((hb, wb), np.float32)
        self.idx_e = np.zeros((hb, wb), np.int32)
        self.idx_d = np.zeros((hb, wb), np.int32)
        self.bases = []
        # every per-frame buffer is allocated and touched once, so a timed push measures
        # arithmetic, not first-touch page faults
        self._scratch = np.zeros(h * w * 4 + ((h + 1) // 2) * ((w + 1) // 2), np.float32)
        self._g = [np.zeros(s + (3,), np.float32) for s in self.shapes]
        for a in self.best_e + self.best_lap + self.best_idx:
            a.fill(0)

    def gaussians(self, frame):
        g = self._g
        src = np.ascontiguousarray(frame)
        if src.dtype == np.uint8:
            lib().orc_to_f32_u8(src, src.size, g[0])
        elif src.dtype == np.uint16:
            lib().orc_to_f32_u16(src, src.size, g[0])
        elif src.dtype == np.float32:
            g = [src] + g[1:]     # float-32 frames are level 0 as they are (read only from here on): no copy
        else:
            g[0][...] = src
        for lv in range(self.levels):
            h, w = self.shapes[lv]
            if self.sep:
                lib().orc_sep_reduce_f32(g[lv], h, w, self.rk, g[lv + 1])
            else:
                lib().orc_reduce_f32(g[lv], h, w, 3, self.k, g[lv + 1], self.fma)
        return g

    def push_frame(self, frame):
        assert frame.shape == (self.h, self.w, 3)
        g = self.gaussians(frame)
        first = int(self.n == 0)
        for lv in range(self.levels):
            h, w = self.shapes[lv]
            hs, ws = self.shapes[lv + 1]
            if self.sep:
                lib().orc_sep_level_select_f32(g[lv], h, w, g[lv + 1], hs, ws, self.k3, self.n, first,
                                               self.best_e[lv], self.best_lap[lv], self.best_idx[lv], self._scratch)
            else:
                lib().orc_level_select_f32(g[lv], h, w, g[lv + 1], hs, ws, self.k, self.n, first,
                                           self.best_e[lv], self.best_lap[lv], self.best_idx[lv],
                                           self._scratch, self.fma)
        hb, wb = self.shapes[self.levels]
        ent = np.empty((hb, wb), np.float32)
        dev = np.empty((hb, wb), np.float32)
        nlev = 256 if self.dtype == np.uint8 else 65536
        lib().orc_base_features_f32(g[-1], hb, wb, nlev, self.pad, ent, dev, self.fma)
        lib().orc_base_select_f32(ent, dev, hb * wb, self.n, first, self.b_ent, self.b_dev,
                                  self.idx_e, self.idx_d)
        self.bases.append(g[-1].copy())
        self.n += 1
        return [a.copy() for a in g] if self.keep_gauss else None

    def fused_base(self):
        hb, wb = self.shapes[self.levels]
        bases = np.ascontiguousarray(np.stack(self.bases))
        out = np.empty((hb, wb, 3), np.float32)
        lib().orc_base_fuse_f32(bases, hb * wb, self.idx_e, self.idx_d, out)
        return out

    def collapse(self):
        img = self.fused_base()
        for lv in range(self.levels - 1, -1, -1):
            h, w = self.shapes[lv]
            hs, ws = self.shapes[lv + 1]
            out = np.empty((h, w, 3), np.float32)
            if self.sep:
                lib().orc_sep_collapse_level_f32(img, hs, ws, self.k3, self.best_lap[lv], h, w, out)
            else:
                lib().orc_collapse_level_f32(img, hs, ws, self.k, self.best_lap[lv], h, w, out,
                                             self.fma)
            img = out
        return img

    def finish(self):
        img = self.collapse()
        out = np.empty((self.h, self.w, 3), self.dtype)
        if self.dtype == np.uint8:
            lib().orc_finalize_u8(img, img.size, out)
        else:
            lib().orc_finalize_u16(img, img.size, out)
        return out


def synth_frame_u8(h, w, f, n, seed=20250824):
    """SURVEY.md 8(d) config-2 generator (C version)."""
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_synth_frame_u8(out, h, w, f, n, seed)
    return out


def synth_crop_u8(H, W, f, n, y0, x0, h, w, seed=20250824):
    """rows [y0, y0+h) x columns [x0, x0+w) of frame f of the H x W config-2 generator."""
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_synth_crop_u8(out, H, W, f, n, seed, y0, x0, h, w)
    return out


def synth_frame_numpy(h, w, f, n, seed=20250824, dtype=np.uint8):
    """Same generator in NumPy integer arithmetic (no libm), for cross-checks."""
    y = np.arange(h, dtype=np.uint32)[:, None, None]
    x = np.arange(w, dtype=np.uint32)[None, :, None]
    c = np.arange(3, dtype=np.uint32)[None, None, :]
    with np.errstate(over="ignore"):
        hsh = (np.uint32(seed) ^ (np.uint32(f) * np.uint32(0x9E3779B1)) ^
               (y * np.uint32(0x85EBCA77)) ^ (x * np.uint32(0xC2B2AE3D)) ^ c)
        hsh ^= hsh >> 16
        hsh *= np.uint32(0x7feb352d)
        hsh ^= hsh >> 15
        hsh *= np.uint32(0x846ca68b)
        hsh ^= hsh >> 16
    noise = (hsh >> 24).astype(np.int32) - 128
    band = (np.arange(h, dtype=np.int64) * n // h).astype(np.int32)[:, None, None]
    d = np.abs(band - f)
    amp = 64 >> np.minimum(d, 6)
    base = ((3 * x.astype(np.int32) + 5 * y.astype(np.int32) + 17 * c.astype(np.int32)) & 127) + 64
    v = np.clip(base + ((noise * amp) >> 7), 0, 255)
    if np.dtype(dtype) == np.uint16:
        return (v * 257).astype(np.uint16)
    return v.astype(np.uint8)


# --------------------------------------------------------------------------
# alignment apply step (align.py:238-251), see align_oracle.c
# --------------------------------------------------------------------------
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REPLICATE_BLUR = 0, 1, 2


def warp_affine(img, M, border_mode=BORDER_REPLICATE_BLUR, border_value=(0, 0, 0, 0),
                blur_ksize=21, blur_sigma=50.0, want_mask=False):
    """warpAffine (+ mask + border blur for BORDER_REPLICATE_BLUR) of an HxWx3 uint8/uint16 image."""
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    dt = 0 if img.dtype == np.uint8 else 1
    M = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(6))
    bv = np.ascontiguousarray(np.asarray(list(border_value) + [0, 0, 0, 0], dtype=np.float64)[:4])
    warp = np.empty_like(img)
    valid = np.empty((h, w), np.uint8)
    mode = 0 if border_mode == BORDER_CONSTANT else 1
    lib().orc_warp_affine(img.ctypes.data, warp.ctypes.data, valid.ctypes.data, h, w, dt, M, mode, bv)
    out = warp
    if border_mode == BORDER_REPLICATE_BLUR:
        out = np.empty_like(img)
        lib().orc_border_blur_composite(warp.ctypes.data, valid.ctypes.data, out.ctypes.data, h, w, dt,
                                        blur_ksize, float(blur_sigma))
    return (out, valid) if want_mask else out


def _warp_raw(fn, img, M, n, mode, border_value):
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    M = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(n))
    bv = np.ascontiguousarray(np.asarray(list(border_value) + [0, 0, 0, 0], dtype=np.float64)[:4])
    warp = np.empty_like(img)
    fn(img.ctypes.data, warp.ctypes.data, None, h, w, 0 if img.dtype == np.uint8 else 1, M,
       0 if mode == BORDER_CONSTANT else 1, bv)
    return warp


def warp_affine_raw(img, M, mode, border_value=(0, 0, 0, 0)):
    """cv2.warpAffine alone (no mask, no composite): the primitive oracle/ref_import.py hands the reference's align.py."""
    return _warp_raw(lib().orc_warp_affine, img, M, 6, mode, border_value)


def warp_perspective_raw(img, M, mode, border_value=(0, 0, 0, 0)):
    return _warp_raw(lib().orc_warp_perspective, img, M, 9, mode, border_value)


def gaussian_blur_fixed(img, ksize=21, sigma=50.0):
    """cv2.GaussianBlur(img, (ksize, ksize), sigmaX=sigma) of an HxWx3 uint8 / uint16 image: OpenCV's bit-exact fixed-point
    path as align_oracle.c restates it [from memory, parity unpinned] (the blur behind align.py:249)."""
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    out = np.empty_like(img)
    nothing_valid = np.zeros((h, w), np.uint8)
    lib().orc_border_blur_composite(img.ctypes.data, nothing_valid.ctypes.data, out.ctypes.data, h, w,
                                    0 if img.dtype == np.uint8 else 1, int(ksize), float(sigma))
    return out


def gauss_kernel_fixed(ksize, sigma, bits):
    k = np.zeros(ksize, np.uint32)
    lib().orc_gauss_kernel_fixed(int(ksize), float(sigma), int(bits), k)
    return k


def warp_perspective(img, M, border_mode=BORDER_REPLICATE_BLUR, border_value=(0, 0, 0, 0),
                     blur_ksize=21, blur_sigma=50.0, want_mask=False):
    """warpPerspective (+ mask + border blur for BORDER_REPLICATE_BLUR), M: 3x3 src->dst (align.py:231-237)."""
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    dt = 0 if img.dtype == np.uint8 else 1
    M = np.ascontiguousarray(np.asarray(M, dtype=np.float64).reshape(9))
    bv = np.ascontiguousarray(np.asarray(list(border_value) + [0, 0, 0, 0], dtype=np.float64)[:4])
    warp = np.empty_like(img)
    valid = np.empty((h, w), np.uint8)
    mode = 0 if border_mode == BORDER_CONSTANT else 1
    lib().orc_warp_perspective(img.ctypes.data, warp.ctypes.data, valid.ctypes.data, h, w, dt, M, mode, bv)
    out = warp
    if border_mode == BORDER_REPLICATE_BLUR:
        out = np.empty_like(img)
        lib().orc_border_blur_composite(warp.ctypes.data, valid.ctypes.data, out.ctypes.data, h, w, dt,
                                        blur_ksize, float(blur_sigma))
    return (out, valid) if want_mask else out


# --------------------------------------------------------------------------
# BalanceFrames device steps (balance.py:158-180 histogram, :30-50 table apply) -- NumPy restatement
# --------------------------------------------------------------------------
def bgr2gray_int(img):
    """cv2.cvtColor(BGR2GRAY) on uint8/uint16 [from memory: 14-bit fixed point, parity unpinned]."""
    a = img.astype(np.uint64)
    g = (a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + (1 << 13)) >> 14
    return g.astype(img.dtype)


def resize_area_int(img, s):
    """cv2.resize(img, (0,0), fx=1/s, fy=1/s, INTER_AREA) for an integer factor [from memory; parity unpinned]: output
    size round-half-even(dim / s); whole s x s blocks -> (sum + 2) >> 2 for s == 2, else round-half-even of the float32
    product sum * float32(1 / s^2); the partial blocks of a last row / column -> round-half-even of float32 sum / count
    over the pixels that exist.  (The same rules as shinestacker_amd.align.img_subsample, written independently: whole
    blocks by reshape, the hanging row / column block by block.)"""
    if s == 1:
        return img
    h, w = img.shape[:2]
    dh, dw = int(np.rint(h * (1.0 / s))), int(np.rint(w * (1.0 / s)))
    tail = img.shape[2:]
    out = np.zeros((dh, dw) + tail, np.int64)
    fh, fw = min(h // s, dh), min(w // s, dw)          # whole blocks
    if fh and fw:
        blk = img[:fh * s, :fw * s].reshape(fh, s, fw, s, -1).astype(np.uint32).sum(axis=(1, 3))
        whole = (blk + 2) >> 2 if s == 2 else np.rint(blk.astype(np.float32) * np.float32(1.0 / (s * s)))
        out[:fh, :fw] = whole.reshape((fh, fw) + tail)

    def partial(by, bx):
        y0, y1, x0, x1 = by * s, min(by * s + s, h), bx * s, min(bx * s + s, w)
        if y0 >= h or x0 >= w:
            return
        tot = img[y0:y1, x0:x1].astype(np.uint32).reshape((-1,) + tail).sum(axis=0)
        out[by, bx] = np.rint(np.float32(tot) / np.float32((y1 - y0) * (x1 - x0)))

    for by in range(fh, dh):
        for bx in range(dw):
            partial(by, bx)
    for bx in range(fw, dw):
        for by in range(fh):
            partial(by, bx)
    return out.astype(img.dtype)


def balance_hist(img, lumi=False, subsample=1, fast=True, mask_size=0.0):
    """What Correction.calc_hist_1ch returns for the luminance (lumi) or for each of B, G, R
    (balance.py:158-180, :235-236, :264-266): int64 [nch][nbins]."""
    nb = 256 if img.dtype == np.uint8 else 65536
    chans = [bgr2gray_int(img)] if lumi else [img[..., c] for c in range(3)]
    out = []
    for ch in chans:
        if subsample > 1:
            ch = ch[::subsample, ::subsample] if fast else resize_area_int(ch[..., None], subsample)[..., 0]
        if mask_size > 0:
            hh, ww = ch.shape
            xv, yv = np.meshgrid(np.linspace(0, ww - 1, ww), np.linspace(0, hh - 1, hh))
            r = min(ww, hh) * mask_size / 2
            ch = ch[(xv - ww / 2) ** 2 + (yv - hh / 2) ** 2 <= r ** 2]
        out.append(np.bincount(ch.ravel(), minlength=nb).astype(np.int64))
    return np.stack(out)


def apply_lut(img, luts):
    """cv2.LUT / np.take per channel (one table: all channels)."""
    luts = np.asarray(luts).reshape(-1, 256 if img.dtype == np.uint8 else 65536)
    out = np.empty_like(img)
    for c in range(3):
        out[..., c] = luts[0 if luts.shape[0] == 1 else c][img[..., c]]
    return out


# --------------------------------------------------------------------------
# 8-bit BGR <-> HSV / HLS (cv2.cvtColor, reference algorithms/balance.py:340-363) -- NumPy restatement of OpenCV's
# color_hsv code [from memory; parity unpinned like every other cv2 primitive here].  cv2 supports these conversions
# for CV_8U and CV_32F only: a 16-bit frame raises in the reference (cv2.error), and in the mirror.
#   BGR2HSV_b : integer arithmetic with the 12-bit reciprocal tables sdiv / hdiv180; H in [0, 180)
#   HSV2BGR_b : through float: (h, s/255, v/255) -> sector tables -> round(x * 255)
#   BGR2HLS_b : through float: x/255 -> h (degrees * 0.5), l, s -> round(h), round(l * 255), round(s * 255)
#   HLS2BGR_b : through float
# float32 operations one by one in the order written (no fused multiply-add), round half to even, saturate.
# --------------------------------------------------------------------------
_HSV_SHIFT = 12
_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def _hsv_tables():
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    sdiv[1:] = np.rint((255 << _HSV_SHIFT) / (1.0 * i)).astype(np.int64)     # saturate_cast<int> = cvRound
    hdiv[1:] = np.rint((180 << _HSV_SHIFT) / (6.0 * i)).astype(np.int64)
    return sdiv, hdiv


def _sat_u8(x):
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def bgr2hsv_u8(img):
    assert img.dtype == np.uint8
    sdiv, hdiv = _hsv_tables()
    b, g, r = (img[..., c].astype(np.int64) for c in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * sdiv[v] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


def _sector_pick(tab, sector):
    """b, g, r = tab[sector_data[sector]]"""
    t = np.stack(tab, axis=-1)                                   # (..., 4)
    idx = _SECTOR[sector]                                        # (..., 3)
    return [np.take_along_axis(t, idx[..., k:k + 1], axis=-1)[..., 0] for k in range(3)]


def hsv2bgr_u8(img):
    assert img.dtype == np.uint8
    f = np.float32
    h = img[..., 0].astype(f)
    s = img[..., 1].astype(f) * f(1.0 / 255.0)
    v = img[..., 2].astype(f) * f(1.0 / 255.0)
    hh = h * f(6.0 / 180.0)
    sector = np.floor(hh).astype(np.int64)
    hh = hh - sector.astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hh = np.where(bad, f(0), hh)
    one = f(1.0)
    tab = [v, v * (one - s), v * (one - s * hh), v * (one - s * (one - hh))]
    b, g, r = _sector_pick(tab, sector)
    gray = s == 0
    out = [np.where(gray, v, c) for c in (b, g, r)]
    return np.stack([_sat_u8(c * f(255.0)) for c in out], axis=-1)


def bgr2hls_u8(img):
    assert img.dtype == np.uint8
    f = np.float32
    b, g, r = (img[..., c].astype(f) * f(1.0 / 255.0) for c in range(3))
    vmax = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = vmax - vmin
    l = (vmax + vmin) * f(0.5)
    ok = diff > np.finfo(f).eps
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(l < f(0.5), diff / (vmax + vmin), diff / (f(2.0) - vmax - vmin))
        d60 = f(60.0) / diff
        h = np.where(vmax == r, (g - b) * d60, np.where(vmax == g, (b - r) * d60 + f(120.0), (r - g) * d60 + f(240.0)))
    h = np.where(h < 0, h + f(360.0), h)
    h = np.where(ok, h, f(0)) * f(180.0 / 360.0)
    s = np.where(ok, s, f(0))
    return np.stack([_sat_u8(h), _sat_u8(l * f(255.0)), _sat_u8(s * f(255.0))], axis=-1)


def hls2bgr_u8(img):
    assert img.dtype == np.uint8
    f = np.float32
    h = img[..., 0].astype(f)
    l = img[..., 1].astype(f) * f(1.0 / 255.0)
    s = img[..., 2].astype(f) * f(1.0 / 255.0)
    one = f(1.0)
    p2 = np.where(l <= f(0.5), l * (one + s), l + s - l * s)
    p1 = f(2.0) * l - p2
    hh = h * f(6.0 / 180.0)
    sector = np.floor(hh).astype(np.int64)
    hh = hh - sector.astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hh = np.where(bad, f(0), hh)
    tab = [p2, p1, p1 + (p2 - p1) * (one - hh), p1 + (p2 - p1) * hh]
    b, g, r = _sector_pick(tab, sector)
    gray = s == 0
    out = [np.where(gray, l, c) for c in (b, g, r)]
    return np.stack([_sat_u8(c * f(255.0)) for c in out], axis=-1)


CVT_BGR2HSV, CVT_HSV2BGR, CVT_BGR2HLS, CVT_HLS2BGR = range(4)
_CVT = {CVT_BGR2HSV: bgr2hsv_u8, CVT_HSV2BGR: hsv2bgr_u8, CVT_BGR2HLS: bgr2hls_u8, CVT_HLS2BGR: hls2bgr_u8}


def cvt_color_u8(img, code):
    if img.dtype != np.uint8:
        raise ValueError("cv2.cvtColor(BGR <-> HSV / HLS) supports 8-bit (and float32) images only")
    return _CVT[code](np.ascontiguousarray(img))
