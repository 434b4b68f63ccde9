"""oracle/cv2_standin.py -- a STAND-IN for the parts of OpenCV the alignment ESTIMATOR uses (feature detectors,
descriptors, matchers, model fits).  TEST INFRASTRUCTURE ONLY (see oracle.py): imported by tests/, by
oracle/ref_import.py (to run the reference's own align.py here) and by oracle/gen_golden.py -- never by the product.

There is no OpenCV in this image and its SIFT / FLANN / RANSAC arithmetic is not a parity target (SURVEY.md 8(a) A2-A3:
"CPU by plan"); what IS a target is everything `algorithms/align.py:48-252` decides around those calls.  So the stand-in
makes the estimator's outputs a simple, exactly reproducible function of the images, and both the reference's
`align_images` (through ref_import.load_align_module) and `shinestacker_amd.align_images` run over the same stand-in:

* images carry their keypoints: a pixel whose 8-bit gray value is 200 + k, and whose left and upper neighbours hold
  a different value, is keypoint k (so a 2 x 2 block of the marker value on the even grid is ONE keypoint at every
  sub-sampling: the s = 2 INTER_AREA average of the block is the marker value again);
* descriptors encode k (a one-hot float vector for SIFT, a bit pattern for the binary ones), matchers are brute force;
* the model fits are closed-form least squares in plain Python floats in a fixed order (no LAPACK, no SIMD
  reductions), so the matrix they return is bit-identical on every IEEE-754 machine -- the frozen `M` of
  tests/golden/align.npz is reproduced on the GPU box.

Every call is appended to `log` (a list) so a test can assert which OpenCV entry points ran, on what, with which
arguments.
"""
import types

import numpy as np


class KeyPoint:
    def __init__(self, x, y, k):
        self.pt, self.k = (float(x), float(y)), k


class DMatch:
    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.distance = q, t, float(d)


def solve_linear(a, b):
    """Gaussian elimination with partial pivoting on Python floats (deterministic order)."""
    n = len(b)
    a = [list(map(float, row)) + [float(b[i])] for i, row in enumerate(a)]
    for c in range(n):
        p = max(range(c, n), key=lambda r: (abs(a[r][c]), -r))
        if a[p][c] == 0.0:
            raise np.linalg.LinAlgError("singular system")
        a[c], a[p] = a[p], a[c]
        for r in range(c + 1, n):
            f = a[r][c] / a[c][c]
            if f != 0.0:
                for k in range(c, n + 1):
                    a[r][k] -= f * a[c][k]
    x = [0.0] * n
    for r in range(n - 1, -1, -1):
        s = a[r][n]
        for k in range(r + 1, n):
            s -= a[r][k] * x[k]
        x[r] = s / a[r][r]
    return x


def fit_similarity(src, dst):
    """4-DoF least squares  x' = a x - b y + tx,  y' = b x + a y + ty  (closed form about the centroids)."""
    s = [(float(p[0]), float(p[1])) for p in np.asarray(src).reshape(-1, 2)]
    d = [(float(p[0]), float(p[1])) for p in np.asarray(dst).reshape(-1, 2)]
    n = len(s)
    sx, sy = sum(p[0] for p in s) / n, sum(p[1] for p in s) / n
    dx, dy = sum(p[0] for p in d) / n, sum(p[1] for p in d) / n
    num_a = num_b = den = 0.0
    for (x, y), (u, v) in zip(s, d):
        xc, yc, uc, vc = x - sx, y - sy, u - dx, v - dy
        num_a += xc * uc + yc * vc
        num_b += xc * vc - yc * uc
        den += xc * xc + yc * yc
    a, b = num_a / den, num_b / den
    tx, ty = dx - (a * sx - b * sy), dy - (b * sx + a * sy)
    return np.array([[a, -b, tx], [b, a, ty]], np.float64)


def fit_homography(src, dst):
    """Least-squares homography with h22 = 1 (normal equations of the 2n x 8 DLT system, Gaussian elimination)."""
    s = [(float(p[0]), float(p[1])) for p in np.asarray(src).reshape(-1, 2)]
    d = [(float(p[0]), float(p[1])) for p in np.asarray(dst).reshape(-1, 2)]
    rows, rhs = [], []
    for (x, y), (u, v) in zip(s, d):
        rows.append([x, y, 1.0, 0.0, 0.0, 0.0, -u * x, -u * y])
        rhs.append(u)
        rows.append([0.0, 0.0, 0.0, x, y, 1.0, -v * x, -v * y])
        rhs.append(v)
    ata = [[sum(r[i] * r[j] for r in rows) for j in range(8)] for i in range(8)]
    atb = [sum(r[i] * t for r, t in zip(rows, rhs)) for i in range(8)]
    h = solve_linear(ata, atb)
    return np.array(h + [1.0], np.float64).reshape(3, 3)


def get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform [from memory of imgwarp.cpp, parity unpinned]: the 8 x 8 system
    (x, y, 1, 0, 0, 0, -x u, -y u | 0, 0, 0, x, y, 1, -x v, -y v) solved by LU with partial pivoting in double, h22 = 1."""
    s, d = np.asarray(src, np.float64).reshape(4, 2), np.asarray(dst, np.float64).reshape(4, 2)
    a, b = [[0.0] * 8 for _ in range(8)], [0.0] * 8
    for i in range(4):
        x, y, u, v = s[i][0], s[i][1], d[i][0], d[i][1]
        a[i][0] = a[i + 4][3] = x
        a[i][1] = a[i + 4][4] = y
        a[i][2] = a[i + 4][5] = 1.0
        a[i][6], a[i][7] = -x * u, -y * u
        a[i + 4][6], a[i + 4][7] = -x * v, -y * v
        b[i], b[i + 4] = u, v
    return np.array(solve_linear(a, b) + [1.0], np.float64).reshape(3, 3)


def install_features(cv2, log):
    """Adds the estimator's entry points (align.py:48-151) to the cv2-like module `cv2`."""
    cv2.RANSAC, cv2.LMEDS, cv2.NORM_HAMMING = 8, 4, 6

    class Feature2D:
        binary = False

        def __init__(self, name):
            self.name = name
            log.append(("create", name))

        def detect(self, img, mask):
            assert img.dtype == np.uint8 and img.ndim == 2 and mask is None
            log.append(("detect", self.name))
            left = np.zeros_like(img)
            left[:, 1:] = img[:, :-1]
            up = np.zeros_like(img)
            up[1:, :] = img[:-1, :]
            ys, xs = np.nonzero((img >= 200) & (left != img) & (up != img))
            return [KeyPoint(x, y, int(img[y, x]) - 200) for y, x in zip(ys, xs)]

        def compute(self, img, kps):
            log.append(("compute", self.name))
            if self.binary:
                d = np.zeros((len(kps), 32), np.uint8)
                for i, kp in enumerate(kps):
                    d[i] = np.unpackbits(np.array([kp.k * 37 + 11], np.uint32).view(np.uint8)).repeat(8)[:256].reshape(32, 8) \
                        .dot(1 << np.arange(8)[::-1]).astype(np.uint8)
            else:
                d = np.zeros((len(kps), 128), np.float32)
                for i, kp in enumerate(kps):
                    d[i, kp.k % 128] = 1.0
                    d[i, (kp.k * 7 + 3) % 128] += 0.25 * (kp.k // 128)
            return kps, d

        def detectAndCompute(self, img, mask):
            log.append(("detectAndCompute", self.name))
            return self.compute(img, self.detect(img, mask))

    def factory(name, binary):
        def create():
            f = Feature2D(name)
            f.binary = binary
            return f
        return create
    cv2.SIFT_create = factory("SIFT", False)
    cv2.ORB_create = factory("ORB", True)
    cv2.AKAZE_create = factory("AKAZE", True)
    cv2.BRISK_create = factory("BRISK", True)
    cv2.FastFeatureDetector_create = factory("FAST", True)

    class FlannBasedMatcher:
        def __init__(self, index_params, search_params):
            log.append(("flann", dict(index_params), dict(search_params)))

        def knnMatch(self, d0, d1, k):
            assert k == 2
            out = []
            for q in range(len(d0)):
                dist = np.sqrt(((d1.astype(np.float64) - d0[q]) ** 2).sum(axis=1))
                order = np.argsort(dist, kind="stable")[:2]
                out.append((DMatch(q, order[0], dist[order[0]]), DMatch(q, order[1], dist[order[1]])))
            return out
    cv2.FlannBasedMatcher = FlannBasedMatcher

    class BFMatcher:
        def __init__(self, norm, crossCheck=False):
            log.append(("bf", norm, crossCheck))
            assert norm == cv2.NORM_HAMMING and crossCheck is True

        def match(self, d0, d1):
            ham = np.unpackbits(d0[:, None, :] ^ d1[None, :, :], axis=2).sum(axis=2)
            fwd, bwd = ham.argmin(axis=1), ham.argmin(axis=0)
            return [DMatch(q, t, ham[q, t]) for q, t in enumerate(fwd) if bwd[t] == q][::-1]   # unsorted on purpose
    cv2.BFMatcher = BFMatcher

    def estimateAffinePartial2D(src, dst, method=None, ransacReprojThreshold=None, confidence=None, refineIters=None):
        log.append(("estimateAffinePartial2D", src.shape, src.dtype, method, ransacReprojThreshold, confidence, refineIters))
        return fit_similarity(src, dst), np.ones((len(src.reshape(-1, 2)), 1), np.uint8)
    cv2.estimateAffinePartial2D = estimateAffinePartial2D

    def findHomography(src, dst, method=None, ransacReprojThreshold=None, maxIters=None):
        log.append(("findHomography", src.shape, src.dtype, method, ransacReprojThreshold, maxIters))
        return fit_homography(src, dst), np.ones((len(src.reshape(-1, 2)), 1), np.uint8)
    cv2.findHomography = findHomography
    return cv2


def make_cv2(log, resize=None, gray=None):
    """A `cv2` module holding ONLY what `opencv_estimator` / `img_subsample` touch.  `resize(img, s)` / `gray(img)`
    default to a strided pick and the channel maximum (marker pixels are gray, so any gray conversion keeps them);
    the golden-fixture tests pass the oracle's INTER_AREA / BGR2GRAY restatements instead."""
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2GRAY, cv2.INTER_AREA = 6, 3

    def cvtColor(im, code):
        assert code == cv2.COLOR_BGR2GRAY and im.dtype == np.uint8 and im.ndim == 3
        log.append(("cvtColor", im.shape))
        return gray(im) if gray else im.max(axis=2)
    cv2.cvtColor = cvtColor

    def resize_(img, dsize, fx=None, fy=None, interpolation=None):
        log.append(("resize", fx, fy, interpolation))
        s = int(round(1 / fx))
        return resize(img, s) if resize else img[::s, ::s]
    cv2.resize = resize_
    return install_features(cv2, log)


def marker_scene(M, n=40, h=240, w=320, dtype=np.uint8, seed=3, block=1, texture=False, parity=(0, 0), grid=2):
    """(moving, reference, src points, dst points): keypoint k is a `block` x `block` patch of value 200 + k whose top-left
    corner sits at p_k (coordinates = parity mod `grid`) in the moving image and at M p_k rounded to the `grid` lattice in
    the reference image.  `texture`: the rest of both images is a smooth pattern plus noise below 180 (so warps of it
    have something to interpolate)."""
    rng = np.random.default_rng(seed)
    scale = 257 if np.dtype(dtype) == np.uint16 else 1
    if texture:
        yy, xx = np.mgrid[0:h, 0:w]
        tex = (60 + 50 * np.sin(xx / 11.0) * np.cos(yy / 7.0))[..., None] * np.array([0.9, 1.0, 0.8]) + \
            rng.integers(0, 40, (h, w, 3))
        mov = (np.clip(tex, 0, 179).astype(np.int64) * scale + (rng.integers(0, 200, (h, w, 3)) if scale > 1 else 0)).astype(dtype)
        # the reference frame: the same content seen through M, nearest neighbour (the estimator stand-in never
        # looks at it; the warp under test acts on `mov`)
        Mi = np.array(M, float)
        Mi = np.vstack([Mi, [0, 0, 1.0]]) if Mi.shape == (2, 3) else Mi
        inv = np.linalg.inv(Mi)
        q = np.stack([xx, yy, np.ones_like(xx)], -1) @ inv.T
        qx = np.clip(np.rint(q[..., 0] / q[..., 2]), 0, w - 1).astype(int)
        qy = np.clip(np.rint(q[..., 1] / q[..., 2]), 0, h - 1).astype(int)
        ref = mov[qy, qx].copy()
    else:
        mov, ref = np.zeros((h, w, 3), dtype), np.zeros((h, w, 3), dtype)
    pts = set()
    tries = 0
    while len(pts) < n and tries < 100000:
        tries += 1
        p = (grid * int(rng.integers(8 // grid + 1, (w - 8) // grid - 1)) + parity[0],
             grid * int(rng.integers(8 // grid + 1, (h - 8) // grid - 1)) + parity[1])
        if all(abs(p[0] - q_[0]) > 2 * grid or abs(p[1] - q_[1]) > 2 * grid for q_ in pts):
            pts.add(p)
    src, dst, taken = [], [], []
    for (x, y) in sorted(pts):
        v = np.array(M, float) @ [x, y, 1.0]
        if len(v) == 3:
            v = v[:2] / v[2]
        u, t = grid * int(round(v[0] / grid)), grid * int(round(v[1] / grid))
        if not (8 <= u < w - 8 and 8 <= t < h - 8) or any(abs(u - a) <= 2 * grid and abs(t - b) <= 2 * grid for a, b in taken):
            continue
        if len(src) >= 56:
            break
        val = (200 + len(src)) * scale
        mov[y:y + block, x:x + block] = val
        ref[t:t + block, u:u + block] = val
        taken.append((u, t))
        src.append((x, y))
        dst.append((u, t))
    # nothing but the markers may reach 200 (8-bit gray)
    return mov, ref, np.array(src, float), np.array(dst, float)
