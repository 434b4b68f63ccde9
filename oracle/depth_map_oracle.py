"""oracle/depth_map_oracle.py -- CPU restatement of the reference's DepthMapStack
(algorithms/depth_map.py:10-123).  TEST INFRASTRUCTURE ONLY: imported by tests/, by
oracle/gen_golden.py and by oracle/ref_import.py's cv2 shim; never by the product.

PARITY UNPINNED for the OpenCV primitives: OpenCV is not installed in this image, so Sobel,
GaussianBlur, Laplacian, bilateralFilter, pyrDown and pyrUp below are restated from OpenCV's
published algorithms (imgproc: deriv.cpp, smooth.dispatch.cpp, bilateral_filter.dispatch.cpp,
pyramids.cpp), with one fixed operation order each, written down in the docstrings.  What IS
pinned: the reference's own control flow -- gen_golden.py runs the reference's DepthMapStack
class itself over these primitives (ref_import.reference_depth_map) and records the result; and
`depth_map_stack` below (the streaming restatement the GPU tests compare with on the GPU box,
where /root/reference does not exist) must reproduce those recordings bit for bit.

Where the order does not matter: with the default parameters the energy maps are EXACT --
gray levels are integers, the 5-tap Gaussian [1 4 6 4 1]/16 and the integer derivative kernels
keep every intermediate inside float32's / float64's exact integer range -- and so is the
image's Gaussian pyramid for 8-bit frames.  Operation order only enters through the bilateral
smoothing, the weights and the weighted sums.
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- borders
def reflect101(idx, n):
    """cv2.BORDER_REFLECT_101 index map, any overshoot (borderInterpolate)."""
    idx = np.asarray(idx)
    if n == 1:
        return np.zeros_like(idx)
    p = 2 * n - 2
    m = np.mod(idx, p)
    return np.where(m >= n, p - m, m)


def _pad101(img, r):
    h, w = img.shape[:2]
    return img[reflect101(np.arange(-r, h + r), h)][:, reflect101(np.arange(-r, w + r), w)]


# --------------------------------------------------------------------------- kernels
_SMALL_GAUSS = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
                7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}


def gaussian_kernel(ksize, dtype=F32, sigma=0.0):
    """cv2.getGaussianKernel(ksize, sigma, CV_32F / CV_64F): fixed table for ksize <= 7 with sigma <= 0,
    else exp(-x^2 / 2 sigma^2) with sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8, evaluated in double, (CV_32F: rounded
    to float32,) normalised by the sum of the stored values."""
    dtype = np.dtype(dtype).type
    if sigma <= 0 and ksize in _SMALL_GAUSS:
        return np.array(_SMALL_GAUSS[ksize], dtype)
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    scale = -0.5 / (sigma * sigma)
    k = np.array([math.exp(scale * (i - (ksize - 1) * 0.5) ** 2) for i in range(ksize)], np.float64).astype(dtype)
    s = 0.0
    for v in k:
        s += float(v)
    s = 1.0 / s
    return np.array([dtype(float(v) * s) for v in k], dtype)


def gaussian_kernel_f32(ksize, sigma=0.0):
    return gaussian_kernel(ksize, F32, sigma)


def sobel_kernels(dx, dy, ksize):
    """cv2.getDerivKernels for the Sobel family (ksize 3, 5, 7, ...): ksize - order - 1 smoothing
    steps [1 1] and `order` difference steps [-1 1].  Integer kernels (float64)."""
    def one(order):
        k = np.array([1.0])
        for _ in range(ksize - order - 1):
            k = np.convolve(k, [1.0, 1.0])
        for _ in range(order):
            k = np.convolve(k, [1.0, -1.0])
        return k[::-1].copy() if order % 2 else k
    return one(dx), one(dy)


def laplacian_kernel2d(ksize):
    """The 2-D kernel cv2.Laplacian applies: fixed 3x3 apertures for ksize 1 / 3, else
    d2/dx2 + d2/dy2 from the Sobel kernels (deriv.cpp: sepFilter2D(kd, ks) + sepFilter2D(ks, kd))."""
    if ksize == 1:
        return np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], np.float64)
    if ksize == 3:
        return np.array([[2, 0, 2], [0, -8, 0], [2, 0, 2]], np.float64)
    kd, ks = sobel_kernels(2, 0, ksize)
    return np.outer(ks, kd) + np.outer(kd, ks)


# --------------------------------------------------------------------------- primitives
def filter2d_f64(img, k2d):
    """Correlation of a float32 / float64 image with a 2-D kernel, accumulated in float64 in row-major
    tap order, REFLECT_101.  (Sobel / Laplacian with ddepth CV_64F work in double.)"""
    kh, kw = k2d.shape
    ry, rx = kh // 2, kw // 2
    h, w = img.shape
    yy = reflect101(np.arange(-ry, h + ry), h)
    xx = reflect101(np.arange(-rx, w + rx), w)
    p = img.astype(np.float64)[yy][:, xx]
    out = np.zeros((h, w), np.float64)
    for i in range(kh):
        for j in range(kw):
            if k2d[i, j] != 0.0:
                out = out + k2d[i, j] * p[i:i + h, j:j + w]
    return out


def sobel_energy(gray, float_type=F32):
    """depth_map.py:32-33: |Sobel_x| + |Sobel_y| (3x3, CV_64F), stored in float_type."""
    kx, ky = sobel_kernels(1, 0, 3)       # x-derivative: row kernel [-1 0 1], column kernel [1 2 1]
    gx = filter2d_f64(gray, np.outer(ky, kx))
    gy = filter2d_f64(gray, np.outer(kx, ky))
    return (np.abs(gx) + np.abs(gy)).astype(float_type)


def gaussian_blur(img, ksize):
    """cv2.GaussianBlur(img, (ksize, ksize), 0) on float32 / float64: separable, rows then columns, each as
    k[c]*S[0] + sum_j k[c+j]*(S[-j] + S[j]) in the image's type (the symmetric row / column filters)."""
    ft = img.dtype.type
    k = gaussian_kernel(ksize, ft)
    r = ksize // 2
    h, w = img.shape

    def sym(p, axis, n):
        sl = [slice(None), slice(None)]

        def at(o):
            sl[axis] = slice(r + o, r + o + n)
            return p[tuple(sl)]
        acc = k[r] * at(0)
        for j in range(1, r + 1):
            acc = acc + k[r + j] * (at(-j) + at(j))
        return acc
    p = img[:, reflect101(np.arange(-r, w + r), w)]
    rows = sym(p, 1, w)
    p = rows[reflect101(np.arange(-r, h + r), h)]
    return sym(p, 0, h)


def gaussian_blur_f32(img, ksize):
    return gaussian_blur(np.asarray(img, F32), ksize)


def laplacian_energy(gray, blur_size, kernel_size, float_type=F32):
    """depth_map.py:39-40."""
    blurred = gaussian_blur(gray, blur_size)
    return np.abs(filter2d_f64(blurred, laplacian_kernel2d(kernel_size))).astype(float_type)


def bilateral_tables(d, sigma_color, sigma_space):
    """(radius, offsets [(dy, dx)], space weights float32, gauss_color_coeff)."""
    radius = max(d // 2, 1) if d > 0 else max(int(round(sigma_space * 1.5)), 1)
    cs = -0.5 / (sigma_space * sigma_space)
    offs, sw = [], []
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            r = math.sqrt(float(i) * i + float(j) * j)
            if r > radius:
                continue
            offs.append((i, j))
            sw.append(F32(math.exp(r * r * cs)))
    return radius, offs, np.array(sw, F32), -0.5 / (sigma_color * sigma_color)


def bilateral_lut(vmin, vmax, color_coeff, nbins=4096):
    """expLUT of the float32 bilateral filter: nbins + 2 entries, entry i = exp((i/scale)^2 * coeff)
    until it underflows to 0; scale_index = nbins / (max - min) in float32."""
    length = F32(F32(float(vmax) - float(vmin)))
    scale_index = F32(F32(nbins) / length)
    lut = np.zeros(nbins + 2, F32)
    last = F32(1.0)
    for i in range(nbins + 2):
        if last > 0:
            val = float(F32(F32(i) / scale_index))
            lut[i] = F32(math.exp(val * val * color_coeff))
            last = lut[i]
    return lut, scale_index


def bilateral_f32(img, d, sigma_color=25.0, sigma_space=25.0):
    """cv2.bilateralFilter on a 1-channel float32 image: disc of radius d/2 in raster order,
    range weight from the 4096-bin interpolated exp table, float32 accumulation
    sum += val*w, wsum += w, result sum / wsum.  A constant image is returned unchanged."""
    img = np.ascontiguousarray(img, F32)
    vmin, vmax = float(img.min()), float(img.max())
    if abs(vmin - vmax) < np.finfo(F32).eps:
        return img.copy()
    radius, offs, sw, cc = bilateral_tables(d, sigma_color, sigma_space)
    lut, scale_index = bilateral_lut(vmin, vmax, cc)
    h, w = img.shape
    p = _pad101(img, radius)
    s = np.zeros((h, w), F32)
    ws = np.zeros((h, w), F32)
    for (dy, dx), swk in zip(offs, sw):
        val = p[radius + dy:radius + dy + h, radius + dx:radius + dx + w]
        alpha = np.abs(val - img) * scale_index
        idx = np.floor(alpha).astype(np.int32)
        alpha = alpha - idx.astype(F32)
        wk = swk * (lut[idx] + alpha * (lut[idx + 1] - lut[idx]))
        s = s + val * wk
        ws = ws + wk
    return s / ws


def _ft(img):
    a = np.asarray(img)
    return a if a.dtype in (np.float32, np.float64) else a.astype(F32)


def pyr_down(img):
    """cv2.pyrDown on float32 / float64 (any channel count): 5-tap [1 4 6 4 1] rows
    (s[2x]*6 + (s[2x-1] + s[2x+1])*4 + s[2x-2] + s[2x+2]), the same across rows, then * 1/256.
    REFLECT_101 on the source indices; destination ((h+1)//2, (w+1)//2)."""
    a = _ft(img)
    T = a.dtype.type
    h, w = a.shape[:2]
    ho, wo = (h + 1) // 2, (w + 1) // 2
    cols = [reflect101(2 * np.arange(wo) + o, w) for o in (-2, -1, 0, 1, 2)]
    row = a[:, cols[2]] * T(6) + (a[:, cols[1]] + a[:, cols[3]]) * T(4) + a[:, cols[0]] + a[:, cols[4]]
    rws = [reflect101(2 * np.arange(ho) + o, h) for o in (-2, -1, 0, 1, 2)]
    out = row[rws[2]] * T(6) + (row[rws[1]] + row[rws[3]]) * T(4) + row[rws[0]] + row[rws[4]]
    return out * T(1.0 / 256.0)


def _up_axis(a, n_dst, axis):
    """One axis of pyrUp (unnormalised): even = s[i-1] + 6 s[i] + s[i+1], odd = 4 (s[i] + s[i+1]);
    ends: first even 6 s[0] + 2 s[1], last even s[n-2] + 7 s[n-1], last odd 8 s[n-1]; an odd destination
    repeats its last sample, one of 2n - 1 drops the last odd sample."""
    T = a.dtype.type
    a = np.moveaxis(a, axis, 0)
    n = a.shape[0]
    out = np.empty((max(n_dst, 2 * n),) + a.shape[1:], a.dtype)
    if n == 1:
        out[0] = a[0] * T(8)
        out[1] = a[0] * T(8)
    else:
        ev = np.empty_like(a)
        od = np.empty_like(a)
        ev[0] = a[0] * T(6) + a[1] * T(2)
        ev[1:-1] = a[:-2] + a[1:-1] * T(6) + a[2:]
        ev[-1] = a[-2] + a[-1] * T(7)
        od[:-1] = (a[:-1] + a[1:]) * T(4)
        od[-1] = a[-1] * T(8)
        out[0:2 * n:2] = ev
        out[1:2 * n:2] = od
    if n_dst > 2 * n:
        out[2 * n] = out[2 * n - 1]
    return np.moveaxis(out[:n_dst], 0, axis)   # a destination of 2n - 1 samples drops the last odd one


def pyr_up(img, dstsize):
    """cv2.pyrUp(img, dstsize=(w, h)) on float32 / float64: columns then rows by _up_axis, then * 1/64."""
    a = _ft(img)
    wd, hd = dstsize
    h, w = a.shape[:2]
    assert abs(wd - 2 * w) == wd % 2 and abs(hd - 2 * h) == hd % 2, "pyrUp: bad dstsize"
    t = _up_axis(a, wd, 1)
    t = _up_axis(t, hd, 0)
    return t * a.dtype.type(1.0 / 64.0)


# --------------------------------------------------------------------------- the stacker
def depth_map_stack(frames, map_type="average", energy="laplacian", kernel_size=5, blur_size=5,
                    smooth_size=15, temperature=0.1, levels=3, float_type="float-32", gray_fn=None):
    """DepthMapStack.focus_stack (depth_map.py:64-123), frame at a time, for float_type 'float-32' / 'float-64'.
    `frames`: list of H x W x 3 uint8 / uint16 BGR arrays.  Returns the fused frame.
    float-64: gray / energy planes and the image pyramids are float64; the bilateral filter still runs on
    float32 copies and its output array is float32 (depth_map.py:46-51), so with smoothing the weights are float32."""
    from . import oracle as orc
    gray_fn = gray_fn or orc.bgr2gray_int
    FT = {"float-32": np.float32, "float-64": np.float64}[float_type]
    dtype = frames[0].dtype
    grays = [gray_fn(f).astype(FT) for f in frames]                               # :70-71, :77
    if energy == "sobel":
        en = [sobel_energy(g, FT) for g in grays]                                 # :28-34
    elif energy == "laplacian":
        en = [laplacian_energy(g, blur_size, kernel_size, FT) for g in grays]     # :36-41
    else:
        raise ValueError("energy")
    mx = max(e.max() for e in en)                                                 # :88
    if mx > 0:
        en = [e / mx for e in en]                                                 # :90
    if smooth_size > 0:
        en = [bilateral_f32(e.astype(F32), smooth_size, 25, 25) for e in en]      # :43-52 (float32 result array)
    WT = en[0].dtype.type
    if map_type == "average":                                                     # :55-57
        tot = np.zeros_like(en[0])
        for e in en:
            tot = tot + e
        # where the sum is 0 the reference leaves np.divide's output uninitialised; 0 here
        weights = [np.divide(e, tot, out=np.zeros_like(e), where=tot != 0) for e in en]
    elif map_type == "max":                                                       # :58-61
        m = en[0]
        for e in en[1:]:
            m = np.maximum(m, e)
        rel = [exp_rounded((e - m) / WT(temperature)) for e in en]
        tot = np.zeros_like(rel[0])
        for r in rel:
            tot = tot + r
        weights = [r / tot for r in rel]
    else:
        raise ValueError("map_type")
    blended = None
    for f, wgt in zip(frames, weights):                                           # :94-112
        gp_img, gp_w = [f.astype(FT)], [wgt]
        for _ in range(levels - 1):
            gp_img.append(pyr_down(gp_img[-1]))
            gp_w.append(pyr_down(gp_w[-1]))
        lp = [gp_img[-1]]
        for j in range(levels - 1, 0, -1):
            size = (gp_img[j - 1].shape[1], gp_img[j - 1].shape[0])
            lp.append(gp_img[j - 1] - pyr_up(gp_img[j], size))
        cur = [lp[j] * gp_w[levels - 1 - j][..., None] for j in range(levels)]
        blended = cur if blended is None else [b + c for b, c in zip(blended, cur)]
    result = blended[0]                                                           # :117-121
    for j in range(1, levels):
        size = (blended[j].shape[1], blended[j].shape[0])
        result = pyr_up(result, size) + blended[j]
    n_values = 255 if dtype == np.uint8 else 65535
    return np.clip(np.absolute(result), 0, n_values).astype(dtype)               # :122-123


def exp_rounded(x):
    """exp in x's type as the correctly rounded value (NumPy's SIMD exp is CPU-dispatch dependent within 1 ulp, so it
    cannot be a parity target; ref_import patches the reference's np.exp to this one when recording golden vectors):
    float32 through float64, float64 through the x87 long double."""
    x = np.asarray(x)
    if x.dtype == np.float32:
        return np.exp(x.astype(np.float64)).astype(F32)
    return np.exp(x.astype(np.longdouble)).astype(np.float64)


def exp_f32(x):
    return exp_rounded(np.asarray(x, F32))
