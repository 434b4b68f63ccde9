#!/usr/bin/env python3
"""SURVEY.md 8(d) "Parity reporting" between the two arithmetics of the library at ANY size, on the GPU:

    MI_ARITH_EXACT      the reference's evaluation order (pyramid.py:20-55 on cv2.filter2D's row-major 25-tap chain)
    MI_ARITH_SEPARABLE  the 5 + 5 tap separable / polyphase form (csrc/kernels_sep.hpp)

run on the SAME device-resident stack.  Reported per pyramid level, in input-LSB units:
  * Gaussian (last frame) max / 99.9-percentile |difference| against the stated bound 2 * l * 32 u maxv;
  * running-max energy max / 99.9-percentile |difference| and the largest ratio to the stated bound 2 tol_E
    (tests/test_sep_tolerance.py: each arithmetic within tol_E = 2 eps_l sqrt(E) + eps_l^2 + 64 u E of float64,
    eps_l = (2 l + 2) 32 u maxv, u = 2^-24);
  * selection-mismatch rate (pixels whose arg-max frame differs), and a NEAR-TIE PROOF for every such pixel: both
    candidate frames a (exact's winner) and b (separable's winner) are pushed ALONE through both arithmetics and their
    energies read back at the pixel; the flip is a near tie iff
        E_exact[a] - E_exact[b] <= 4 tol_E   and   E_sep[b] - E_sep[a] <= 4 tol_E
    (each is the true gap T[a] - T[b], itself <= 2 tol_E in absolute value, seen through two errors of tol_E), and the
    two arithmetics agree on each candidate within 2 tol_E;
  * final image: |difference| histogram, and every value that differs by >= 2 counts accounted for: it lies in the
    collapse footprint of a Laplacian-level pixel whose selection flipped (the base level's own flips -- a discontinuity
    of the reference's truncating base rule -- are bounded separately through the fused base images' difference).

Test infrastructure (used by tests/test_gpu_fullsize.py and by bench.py after its timed region); no oracle involved: both
sides are the product's own kernels, each of which is separately bit-exact against its CPU restatement.
"""
import numpy as np

U = 2.0 ** -24


def _tol_e(E, lv, maxv):
    eps = (2 * lv + 2) * 32 * U * maxv
    return 2 * eps * np.sqrt(E) + eps * eps + 64 * U * E


def _dilate3(m):
    """3 x 3 binary dilation"""
    p = np.pad(m, 1)
    out = np.zeros_like(m)
    for dy in range(3):
        for dx in range(3):
            out |= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
    return out


def _up(m, shape):
    """influence of level l+1 pixels on level l through expand: pixel (y, x) reads (y >> 1) + {-1, 0, 1}"""
    d = _dilate3(m)
    return np.repeat(np.repeat(d, 2, axis=0), 2, axis=1)[:shape[0], :shape[1]]


def report(L, dev_ptr, n, H, W, dtype, device=0, max_proof_pixels=2_000_000, log=None):
    """`dev_ptr`: n contiguous H x W x 3 frames of `dtype` resident on `device`.  Returns the report dict; the key
    "ok" is True when every statement above holds."""
    say = log or (lambda *_: None)
    dt = np.dtype(dtype)
    fb = H * W * 3 * dt.itemsize
    maxv = 65535.0 if dt == np.uint16 else 255.0
    out_dt = np.uint16 if dt == np.uint16 else np.uint8
    kw = dict(in_dtype=dt, out_dtype=out_dt, device=device)
    st = {}
    for arith in ("exact", "separable"):
        st[arith] = L.Stack(H, W, arith=arith, **kw)
        st[arith].push_frames_device(dev_ptr, n, fb)
    levels = st["exact"].levels
    rep = {"frames": n, "height": H, "width": W, "dtype": dt.name, "levels": []}
    ok = True
    mism = []          # per level: (flat pixel index, a, b)
    masks = []
    for lv in range(levels):
        Ee, Es = st["exact"].tap(L.TAP_ENERGY, lv), st["separable"].tap(L.TAP_ENERGY, lv)
        Ie, Is = st["exact"].tap(L.TAP_INDEX, lv), st["separable"].tap(L.TAP_INDEX, lv)
        dE = np.abs(Ee.astype(np.float64) - Es)
        tol = _tol_e(np.maximum(Ee, Es).astype(np.float64), lv, maxv)
        ratio = float((dE / (2 * tol)).max())
        m = Ie != Is
        row = {"level": lv, "shape": list(Ee.shape),
               "energy_abs_diff_max": float(dE.max()), "energy_abs_diff_p999": float(np.quantile(dE, 0.999)),
               "energy_diff_over_bound_max": ratio, "selection_mismatch_rate": float(m.mean()),
               "selection_mismatches": int(m.sum())}
        if lv >= 1:
            Ge, Gs = st["exact"].tap(L.TAP_GAUSS, lv), st["separable"].tap(L.TAP_GAUSS, lv)
            dG = np.abs(Ge.astype(np.float64) - Gs)
            row.update(gauss_abs_diff_max_lsb=float(dG.max()), gauss_abs_diff_p999_lsb=float(np.quantile(dG, 0.999)),
                       gauss_bound_lsb=2 * lv * 32 * U * maxv)
            ok &= row["gauss_abs_diff_max_lsb"] <= row["gauss_bound_lsb"]
        ok &= ratio <= 1.0
        rep["levels"].append(row)
        idx = np.flatnonzero(m.ravel())
        mism.append((idx, Ie.ravel()[idx], Is.ravel()[idx], Ee.ravel()[idx], Es.ravel()[idx]))
        masks.append(m)
        say(f"level {lv}: {row}")
    # base level (pyramid.py:95-111): the features are computed on gray(G_L) TRUNCATED to an integer (:99-101), so a G_L
    # difference of 1e-4 LSB next to an integer changes a histogram bin and with it every entropy of that frame -- a
    # flip here is a discontinuity of the reference's own rule, not an energy near tie, and where the frames' base images
    # are as alike as the generator's it happens on a good part of the base pixels.  What it can do to the result is
    # bounded by the difference of the two FUSED base images: expand is an averaging operator (non-negative taps that
    # sum to one), so a fused-base difference below one LSB moves no collapsed value by a full count on its own.
    base_mask = (st["exact"].tap(L.TAP_BASE_IDX_E) != st["separable"].tap(L.TAP_BASE_IDX_E)) | \
                (st["exact"].tap(L.TAP_BASE_IDX_D) != st["separable"].tap(L.TAP_BASE_IDX_D))
    dB = np.abs(st["exact"].tap(L.TAP_GAUSS, levels).astype(np.float64) - st["separable"].tap(L.TAP_GAUSS, levels))
    fe, fs = st["exact"].finish(), st["separable"].finish()     # the fused base exists once the stack is finished
    dF = np.abs(st["exact"].tap(L.TAP_FUSED_BASE).astype(np.float64) - st["separable"].tap(L.TAP_FUSED_BASE))
    lsb = 257.0 if dt == np.uint16 else 1.0
    rep["base"] = {"shape": list(base_mask.shape), "selection_mismatches": int(base_mask.sum()),
                   "gauss_abs_diff_max_lsb": float(dB.max()), "gauss_bound_lsb": 2 * levels * 32 * U * maxv,
                   "fused_base_abs_diff_max_lsb": float(dF.max()) / lsb}
    rep["base"]["ok_below_0.9_lsb"] = bool(rep["base"]["fused_base_abs_diff_max_lsb"] < 0.9)
    ok &= rep["base"]["gauss_abs_diff_max_lsb"] <= rep["base"]["gauss_bound_lsb"]
    ok &= rep["base"]["ok_below_0.9_lsb"]
    for s in st.values():
        s.close()
    d = np.abs(fe.astype(np.int32) - fs.astype(np.int32))
    rep["final_abs_diff_counts_0_1_2_3plus"] = [int(x) for x in np.bincount(np.minimum(d.ravel(), 3), minlength=4)]
    rep["final_max_abs_diff"] = int(d.max())
    if out_dt == np.uint16:   # the same histogram in 8-bit-equivalent counts (1 count = 257): what the gates of an 8-bit run mean here
        rep["final_abs_diff_counts_0_1_2_3plus_lsb8"] = [int(x) for x in np.bincount(np.minimum(d.ravel() // 257, 3), minlength=4)]

    # ---- every value off by >= 2 counts lies in the collapse footprint of a flipped LAPLACIAN selection: elsewhere the
    # collapsed floats differ by the fused-base difference (< 0.9 LSB, checked above) plus the two arithmetics' Laplacian
    # differences of the SAME frame (<= 2 eps_l per level, some 1e-3 LSB in all) -- less than one count
    big = (d >= 2 * (257 if out_dt == np.uint16 else 1)).any(axis=2)
    infl = masks[levels - 1]
    for lv in range(levels - 2, -1, -1):
        infl = _up(infl, masks[lv].shape) | masks[lv]
    rep["flip_footprint_fraction_of_image"] = float(infl.mean())
    rep["final_pixels_off_by_2plus"] = int(big.sum())
    rep["final_pixels_off_by_2plus_outside_a_flip_footprint"] = int((big & ~infl).sum())
    ok &= rep["final_pixels_off_by_2plus_outside_a_flip_footprint"] == 0

    # ---- near-tie proof: candidate frames alone through both arithmetics
    total = sum(len(x[0]) for x in mism)
    rep["near_tie"] = {"pixels": total, "checked": 0}
    if total:
        keep = 1.0 if total <= max_proof_pixels else max_proof_pixels / total
        rng = np.random.default_rng(0)
        sel = []
        for (idx, a, b, ee, es) in mism:
            k = np.ones(len(idx), bool) if keep >= 1.0 else rng.random(len(idx)) < keep
            sel.append(tuple(x[k] for x in (idx, a, b, ee, es)))
        frames = np.unique(np.concatenate([np.concatenate([s[1], s[2]]) for s in sel]))
        # energies of (frame a, frame b) under (exact, separable) at every selected pixel
        got = [{k: np.full(len(s[0]), np.nan) for k in ("ea", "eb", "sa", "sb")} for s in sel]
        probe = {arith: L.Stack(H, W, arith=arith, **kw) for arith in ("exact", "separable")}
        for f in frames:
            f = int(f)
            need = [np.flatnonzero((s[1] == f) | (s[2] == f)) for s in sel]
            for arith, (ka, kb) in (("exact", ("ea", "eb")), ("separable", ("sa", "sb"))):
                p = probe[arith]
                p.reset()
                p.set_first_index(f)
                p.push_frames_device(dev_ptr + f * fb, 1, fb)
                for lv, (s, nd) in enumerate(zip(sel, need)):
                    if not len(nd):
                        continue
                    E = p.tap(L.TAP_ENERGY, lv).ravel()
                    v = E[s[0][nd]]
                    isa = s[1][nd] == f
                    got[lv][ka][nd[isa]] = v[isa]
                    isb = s[2][nd] == f
                    got[lv][kb][nd[isb]] = v[isb]
        for p in probe.values():
            p.close()
        worst_gap, worst_cross, bad, self_check = 0.0, 0.0, 0, True
        for lv, (s, g) in enumerate(zip(sel, got)):
            if not len(s[0]):
                continue
            # the single-frame energies of the winners ARE the running maxima of the full run
            self_check &= bool(np.array_equal(g["ea"].astype(np.float32), s[3]) and np.array_equal(g["sb"].astype(np.float32), s[4]))
            tol = _tol_e(np.maximum(g["ea"], g["sb"]), lv, maxv)
            gap_e, gap_s = g["ea"] - g["eb"], g["sb"] - g["sa"]
            cross = np.maximum(np.abs(g["ea"] - g["sa"]), np.abs(g["eb"] - g["sb"]))
            worst_gap = max(worst_gap, float((np.maximum(gap_e, gap_s) / (4 * tol)).max()))
            worst_cross = max(worst_cross, float((cross / (2 * tol)).max()))
            bad += int(((gap_e > 4 * tol) | (gap_s > 4 * tol) | (gap_e < 0) | (gap_s < 0) | (cross > 2 * tol)).sum())
            rep["near_tie"]["checked"] += len(s[0])
        rep["near_tie"].update(frames_probed=int(len(frames)), winner_energy_reproduced=self_check,
                               gap_over_bound_max=worst_gap, cross_arith_diff_over_bound_max=worst_cross,
                               not_a_near_tie=bad, sampled=keep < 1.0)
        ok &= bad == 0 and self_check
    rep["ok"] = bool(ok)
    return rep


def real_crop_frames():
    """the six frames of tests/golden/img_jpg_crop (crops of the reference's examples/input/img-jpg), BGR uint8"""
    import os
    from PIL import Image
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "img_jpg_crop")
    return [np.ascontiguousarray(np.array(Image.open(os.path.join(d, n)))[:, :, ::-1]) for n in sorted(os.listdir(d))]


def ramp_frames(H, W, n, orc=None):
    """the config-2 generator with a per-frame exposure ramp (gain 0.70 .. 1.30 and an offset), so that the frames' LOW-PASS
    content differs and the base-level winners (pyramid.py:95-111) matter to the result"""
    if orc is None:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle as orc
    out = []
    for f in range(n):
        v = orc.synth_frame_numpy(H, W, f, n).astype(np.float64)
        gain = 0.70 + 0.60 * f / max(n - 1, 1)
        out.append(np.clip(np.rint(v * gain + 3.0 * ((f * 5) % 7)), 0, 255).astype(np.uint8))
    return out


def defocus_frames(H, W, n, dtype=np.uint8, seed=7):
    """A simulated FOCUS STACK of a natural-looking scene -- what the reference is for -- instead of the bench generator's
    independent noise: one sharp scene (octaves of smooth random fields = 1/f texture, hard edges, a few clipped highlights,
    dark areas) over a depth map (a tilted plane with steps); frame f is focused at depth f / (n - 1): every pixel is the
    scene blurred by a Gaussian whose sigma grows with its distance from the focal plane (interpolated between six
    pre-blurred copies), times a slight exposure drift, plus sensor noise, quantised to `dtype`.  Blur decays monotonically
    towards the focal plane, so the frames' energies order the way real stacks do (coherent winners, smooth energy
    landscapes with genuine near ties BETWEEN neighbouring frames -- the hard case for an arg-max)."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    scene = np.zeros((H, W, 3), np.float32)
    for k in range(1, 8):
        g = rng.standard_normal((H // 2 ** k + 2, W // 2 ** k + 2, 3)).astype(np.float32)
        scene += ndimage.zoom(g, (2 ** k, 2 ** k, 1), order=1)[:H, :W] * (2.0 ** (k - 4))
    scene = (scene - scene.min()) / (scene.max() - scene.min())
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    scene *= 0.55 + 0.45 * (((xx // 97 + yy // 61) % 2)[..., None])            # hard-edged patches
    scene[(xx - 0.3 * W) ** 2 + (yy - 0.6 * H) ** 2 < (0.05 * H) ** 2] = 1.6     # a highlight that clips
    scene[int(0.8 * H):, : int(0.25 * W)] *= 0.04                                # a nearly black corner
    depth = np.clip(0.1 + 0.8 * (0.6 * xx / W + 0.4 * yy / H) + 0.15 * ((xx // 401) % 2) - 0.075, 0.0, 1.0)
    sigmas = [0.0, 0.6, 1.2, 2.2, 3.6, 5.5]
    blurred = [scene if sg == 0 else ndimage.gaussian_filter(scene, (sg, sg, 0)) for sg in sigmas]
    hi = 255.0 if np.dtype(dtype) == np.uint8 else 65535.0
    out = []
    for f in range(n):
        d = np.abs(depth - f / max(n - 1, 1)) * 9.0                   # blur sigma of this frame at every pixel
        idx = np.clip(np.searchsorted(sigmas, d, side="right") - 1, 0, len(sigmas) - 2)
        t = np.clip((d - np.take(sigmas, idx)) / (np.take(sigmas, idx + 1) - np.take(sigmas, idx)), 0.0, 1.0)[..., None]
        img = np.zeros_like(scene)
        for k in range(len(sigmas) - 1):
            m = (idx == k)[..., None]
            img += m * ((1.0 - t) * blurred[k] + t * blurred[k + 1])
        gain = 1.0 + 0.03 * np.sin(1.7 * f)
        noisy = img * gain * (0.8 * hi) + rng.normal(0.0, 0.004 * hi, img.shape).astype(np.float32)
        out.append(np.clip(np.rint(noisy), 0, hi).astype(dtype))
    return out


def report_host_frames(L, frames, **kw):
    """report() on frames given as host arrays (uploaded into one device buffer)"""
    n = len(frames)
    H, W = frames[0].shape[:2]
    dt = frames[0].dtype
    fb = H * W * 3 * dt.itemsize
    buf = L.DeviceBuffer(fb * n)
    for i, f in enumerate(frames):
        buf.upload(np.ascontiguousarray(f), offset=i * fb)
    try:
        return report(L, buf.ptr, n, H, W, dt, **kw)
    finally:
        buf.free()


if __name__ == "__main__":
    import argparse
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from shinestacker_amd import _lib as L
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--height", type=int, default=4000)
    ap.add_argument("--width", type=int, default=6000)
    ap.add_argument("--dtype", default="f32", choices=["u8", "u16", "f32"])
    ap.add_argument("--source", default="generator", choices=["generator", "crops", "ramp"],
                    help="crops: the six real frames of tests/golden/img_jpg_crop; ramp: the generator with an exposure ramp")
    a = ap.parse_args()
    dt = {"u8": np.uint8, "u16": np.uint16, "f32": np.float32}[a.dtype]
    L.require_device()
    log = lambda s: print(s, file=sys.stderr)   # noqa: E731
    if a.source == "crops":
        print(json.dumps(report_host_frames(L, real_crop_frames(), log=log)))
    elif a.source == "ramp":
        print(json.dumps(report_host_frames(L, ramp_frames(a.height, a.width, a.frames), log=log)))
    else:
        buf = L.DeviceBuffer(a.height * a.width * 3 * np.dtype(dt).itemsize * a.frames)
        L.synth_frames_device(buf.ptr, dt, a.height, a.width, 0, a.frames, a.frames)
        print(json.dumps(report(L, buf.ptr, a.frames, a.height, a.width, dt, log=log)))
