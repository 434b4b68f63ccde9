"""CPU: the stated tolerance of MI_ARITH_SEPARABLE (oracle/separable_oracle.c == csrc/kernels_sep.hpp bit for bit, see
tests/test_gpu_separable.py) against a float64 evaluation of the reference's algorithm (reference-shaped: 2-D
float64 outer-product kernel, per-channel filter at full size, zero-stuffed expand -- oracle.RefShaped(float64),
algorithms/pyramid.py:20-55).

Stated bounds (SURVEY.md section 7 "hard parts", section 8(d) "parity reporting"); u = 2^-24, maxv = 255 / 65535:
  * one 25-term float32 dot product:            |x - x64| <= 32 u sum|w_i x_i|  (= 32 u maxv for a convex stencil)
  * Gaussian level l (l convolutions deep):     |G - G64|   <= l * 32 u maxv
  * Laplacian level l (G_l - expand(G_{l+1})):  |L - L64|   <= (2 l + 2) * 32 u maxv   =: eps_l
  * energy E = blur(gray(L)^2):                 |E - E64|   <= 2 eps_l sqrt(E64) + eps_l^2 + 64 u E64   =: tol_E
  * selection: wherever the arg-max differs from the float64 arg-max, the two candidates are a near tie:
                                                E64[true] - E64[chosen] <= 2 tol_E(E64[true])
  * final image: equal to the float64 result except where a near tie flipped the winner or the float value
    sits within the Laplacian bound of an integer boundary of the truncating cast: |diff| <= 1 count there.
"""
import numpy as np
import pytest

U = 2.0 ** -24


def make_frames(rng, h, w, n, dtype):
    hi = 65535 if dtype == np.uint16 else 255
    yy, xx = np.mgrid[0:h, 0:w]
    frames = []
    for f in range(n):
        base = (0.5 + 0.5 * np.sin(xx / 9.0 + f) * np.cos(yy / 7.0 - f)) * hi * 0.6
        sharp = rng.integers(0, hi + 1, (h, w, 3)) * (0.15 + 0.25 * ((yy * n // h) == f))[:, :, None]
        frames.append(np.clip(base[:, :, None] + sharp, 0, hi).astype(dtype))
    return frames


def gray64(lap):
    c = np.array([np.float32(0.114), np.float32(0.587), np.float32(0.299)], np.float64)
    return lap @ c


@pytest.mark.parametrize("dtype,shape,n,min_size", [(np.uint8, (150, 221), 5, 16), (np.uint16, (97, 130), 4, 8)])
def test_separable_within_stated_tolerance_of_float64(oracle, dtype, shape, n, min_size, capsys):
    rng = np.random.default_rng(7)
    h, w = shape
    maxv = 65535.0 if dtype == np.uint16 else 255.0
    frames = make_frames(rng, h, w, n, dtype)
    ref = oracle.RefShaped(min_size=min_size, float_type=np.float64)
    levels = oracle.num_levels(h, w, min_size)
    pyr64 = [ref.laplacian_pyramid(f, levels) for f in frames]          # (laplacians + base, gaussians), float64
    k2d = oracle.gen_kernel_2d()

    # per-frame separable results: a fresh streaming oracle per frame (its first frame wins everywhere, so the
    # running state IS that frame's energy and Laplacian)
    e32, lap32, g32 = [], [], []
    for f in frames:
        so = oracle.StreamingOracle(h, w, dtype, min_size=min_size, arith="separable")
        g32.append(so.push_frame(f))
        e32.append([a.copy() for a in so.best_e])
        lap32.append([a.copy() for a in so.best_lap])

    report = []
    for lv in range(levels):
        eps = (2 * lv + 2) * 32 * U * maxv
        # pyramid coefficients
        for fi in range(n):
            if lv >= 1:
                dg = np.abs(g32[fi][lv] - pyr64[fi][1][lv]).max()
                assert dg <= lv * 32 * U * maxv, (lv, fi, dg)
            dl = np.abs(lap32[fi][lv] - pyr64[fi][0][lv])
            assert dl.max() <= eps, (lv, fi, dl.max(), eps)
        # energies and the selection
        E64 = np.stack([oracle.filter2D(np.square(gray64(pyr64[fi][0][lv])), k2d) for fi in range(n)])
        E32 = np.stack([e32[fi][lv] for fi in range(n)]).astype(np.float64)
        tolE = 2 * eps * np.sqrt(E64) + eps * eps + 64 * U * E64
        assert np.all(np.abs(E32 - E64) <= tolE), (lv, (np.abs(E32 - E64) / tolE).max())
        i64, i32 = np.argmax(E64, axis=0), np.argmax(E32, axis=0)       # first maximum in both
        mism = i64 != i32
        if mism.any():
            yy, xx = np.nonzero(mism)
            top, got = E64[i64[yy, xx], yy, xx], E64[i32[yy, xx], yy, xx]
            assert np.all(top - got <= 2 * tolE[i64[yy, xx], yy, xx]), "an arg-max flip that is not a near tie"
        report.append((lv, float(np.abs(E32 - E64).max()), float(mism.mean())))

    # whole stacks: float64 reference-shaped result vs the separable streaming oracle
    want = ref.stack(frames)
    so = oracle.StreamingOracle(h, w, dtype, min_size=min_size, arith="separable")
    for f in frames:
        so.push_frame(f)
    got = so.finish()
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    hist = np.bincount(np.minimum(d.ravel(), 3), minlength=4)
    with capsys.disabled():
        print(f"\n[separable vs float64] {np.dtype(dtype).name} {h}x{w} N={n}: per level (max |E-E64|, selection "
              f"mismatch rate) {report}; final-image |diff| histogram 0/1/2/3+: {hist.tolist()}")
    # the float-32 exact-order mode differs from float64 in the same way (truncating cast): the separable mode must
    # not be worse than a few times that
    exact = oracle.StreamingOracle(h, w, dtype, min_size=min_size)
    for f in frames:
        exact.push_frame(f)
    de = np.abs(exact.finish().astype(np.int64) - want.astype(np.int64))
    assert (d > 1).mean() <= max(2 * (de > 1).mean(), 1e-4)
    assert (d > 0).mean() <= max(3 * (de > 0).mean(), 2e-3)
