"""CPU: the oracle restatements reproduce every golden vector frozen from the
reference's own control flow (oracle/gen_golden.py), plus known-answer checks of
the conventions the restatement assumes."""
import json
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, f64_cases, fusion_cases, load_golden, stack_kwargs


@pytest.mark.parametrize("case", fusion_cases())
def test_ref_shaped_matches_golden(oracle, case):
    g = load_golden(case)
    kw = stack_kwargs(g["params"])
    rs = oracle.RefShaped(**kw)
    out, d = rs.stack(list(g["frames"]), want_detail=True)
    assert np.array_equal(out, g["final"])
    assert np.array_equal(d["collapsed"], g["collapsed"])
    for lv in range(int(g["levels"])):
        assert np.array_equal(d["fused"][lv], g[f"fused_{lv}"])
        assert np.array_equal(d["best"][lv], g[f"best_{lv}"])
        assert np.array_equal(d["energy"][lv], g[f"energy_{lv}"])
    assert np.array_equal(d["fused"][-1], g["fused_base"])
    assert np.array_equal(d["ent"].max(axis=0), g["base_ent"].max(axis=0))


@pytest.mark.parametrize("case", f64_cases())
def test_ref_shaped_float64_matches_golden(oracle, case):
    """float_type='float-64': the reference's own run (oracle/gen_golden.py f64_case) vs RefShaped(float64)."""
    g = load_golden(case)
    rs = oracle.RefShaped(float_type=np.float64, **stack_kwargs(g["params"]))
    out, d = rs.stack(list(g["frames"]), want_detail=True)
    assert np.array_equal(out, g["final"])
    assert d["collapsed"].dtype == np.float64 and np.array_equal(d["collapsed"], g["collapsed"])
    for lv in range(int(g["levels"])):
        assert np.array_equal(d["fused"][lv], g[f"fused_{lv}"])
        assert np.array_equal(d["best"][lv], g[f"best_{lv}"])
        assert d["energy"][lv].dtype == np.float32 and np.array_equal(d["energy"][lv], g[f"energy_{lv}"])
    assert np.array_equal(d["fused"][-1], g["fused_base"])
    assert d["ent"].dtype == np.float64 and np.array_equal(d["ent"], g["base_ent"])
    assert np.array_equal(d["dev"], g["base_dev"])


@pytest.mark.parametrize("case", fusion_cases())
def test_streaming_matches_golden(oracle, case):
    g = load_golden(case)
    kw = stack_kwargs(g["params"])
    fr = g["frames"]
    so = oracle.StreamingOracle(fr.shape[1], fr.shape[2], fr.dtype, **kw)
    gs = [so.push_frame(f) for f in fr]
    assert so.levels == int(g["levels"])
    assert np.array_equal(so.finish(), g["final"])
    for lv in range(so.levels):
        assert np.array_equal(so.best_lap[lv], g[f"fused_{lv}"])
        assert np.array_equal(so.best_idx[lv], g[f"best_{lv}"])
        assert np.array_equal(so.best_e[lv], g[f"energy_{lv}"])
    assert np.array_equal(so.fused_base(), g["fused_base"])
    assert np.array_equal(so.idx_e, g["base_idx_e"])
    assert np.array_equal(so.idx_d, g["base_idx_d"])
    if case == "g1_u8":
        for f in range(len(fr)):
            for lv in range(1, so.levels + 1):
                assert np.array_equal(gs[f][lv], g[f"gauss_f{f}_l{lv}"])


def test_ties_pick_first_frame(oracle):
    g = load_golden("g3_ties")
    # frames 1 and 2 are identical, 0 and 3 are constant: index 2 and 3 never win
    for lv in range(int(g["levels"])):
        assert set(np.unique(g[f"best_{lv}"])) <= {0, 1}


def test_primitives_golden(oracle):
    g = load_golden("g5_primitives")
    k = oracle.k25_f32()
    for key in [n[3:] for n in g if n.startswith("in_")]:
        src = np.ascontiguousarray(g["in_" + key])
        h, w = src.shape[:2]
        red = np.empty(((h + 1) // 2, (w + 1) // 2, 3), np.float32)
        oracle.lib().orc_reduce_f32(src, h, w, 3, k, red, 1)
        assert np.array_equal(red, g["reduce_" + key]), key
        ex = np.empty((2 * h, 2 * w, 3), np.float32)
        oracle.lib().orc_expand_f32(src, h, w, 3, k, 2 * h, 2 * w, ex, 1)
        assert np.array_equal(ex, g["expand_" + key]), key


def test_expand_border_taps_analytic(oracle):
    """REFLECT101 acts on the zero-stuffed grid: an impulse in the last row/col is
    seen twice by the far border (index 2h reflects to 2h-2), once by the near one."""
    k1 = oracle.gen_kernel_1d().astype(np.float64)
    img = np.zeros((4, 4, 3), np.float32)
    img[3, 3] = 1.0
    ex = np.empty((8, 8, 3), np.float32)
    oracle.lib().orc_expand_f32(img, 4, 4, 3, oracle.k25_f32(), 8, 8, ex, 1)
    # along one axis the response at out=7 (odd) is k[1]*V[3] + k[3]*V[4], V[4]=V[3]
    col = 2.0 * np.array([k1[0], k1[1], k1[2] + k1[4], k1[1] + k1[3]])  # out = 4,5,6,7
    expect = np.outer(col, col)
    assert np.allclose(ex[4:, 4:, 0], expect, rtol=1e-6, atol=0)
    assert np.all(ex[:4, :, 0] == 0) and np.all(ex[:, :4, 0] == 0)


def test_base_golden(oracle):
    g = load_golden("g7_base")
    for ks in (3, 5, 7):
        for dt, hi in (("uint8", 256), ("uint16", 65536)):
            tag = f"k{ks}_{dt}"
            imgs = g["in_" + tag]
            rs = oracle.RefShaped(kernel_size=ks)
            fused, _be, _bd, ent, dev = rs.fuse_base(list(imgs), np.dtype(dt).type)
            assert np.array_equal(fused, g["fused_" + tag])
            for i in range(len(imgs)):
                e = np.empty((7, 9), np.float32)
                d = np.empty((7, 9), np.float32)
                oracle.lib().orc_base_features_f32(np.ascontiguousarray(imgs[i]), 7, 9, hi,
                                                   (ks - 1) // 2, e, d, 1)
                assert np.array_equal(e, g["ent_" + tag][i])
                assert np.array_equal(d, g["dev_" + tag][i])


def test_numpy_sum_order(oracle):
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 8, 9, 24, 25, 49, 121, 128, 129, 300):
        for _ in range(200):
            a = (rng.standard_normal(n) * 1000).astype(np.float32)
            assert np.float32(a.sum()) == oracle.lib().orc_np_sum_f32(a, n)


def test_levels_and_shapes(oracle):
    with open(os.path.join(GOLDEN, "plumbing.json")) as fh:
        lv = json.load(fh)["levels"]
    for key, want in lv.items():
        h, w = map(int, key.split("x"))
        assert oracle.num_levels(h, w) == want
    assert oracle.level_shapes(4000, 6000, 6)[-1] == (63, 94)
    assert oracle.level_shapes(5760, 8640, 7)[-1] == (45, 68)
    assert oracle.num_levels(40, 40, 8) == 2
    assert oracle.num_levels(24, 500, 2) == 2  # early stop: a side would drop below 4


def test_synth_generator(oracle):
    with open(os.path.join(GOLDEN, "synth_crc.json")) as fh:
        meta = json.load(fh)
    for f, crc in meta["crc32"].items():
        a = oracle.synth_frame_u8(meta["h"], meta["w"], int(f), meta["n"], meta["seed"])
        b = oracle.synth_frame_numpy(meta["h"], meta["w"], int(f), meta["n"], meta["seed"])
        assert np.array_equal(a, b)
        assert zlib.crc32(a.tobytes()) == crc


def test_nofma_differs_from_fma(oracle):
    """The two arithmetic modes are genuinely different roundings."""
    g = load_golden("g1_u8")
    fr = g["frames"][0].astype(np.float32)
    k = oracle.k25_f32()
    a = np.empty((34, 51, 3), np.float32)
    b = np.empty_like(a)
    oracle.lib().orc_reduce_f32(np.ascontiguousarray(fr), 67, 101, 3, k, a, 1)
    oracle.lib().orc_reduce_f32(np.ascontiguousarray(fr), 67, 101, 3, k, b, 0)
    assert not np.array_equal(a, b)
    assert np.allclose(a, b, rtol=1e-5)


@pytest.mark.parametrize("h,w,exact", [(64, 96, True), (66, 100, True), (63, 96, False), (64, 95, False)])
def test_sep_natural_extension(h, w, exact):
    """What the edge tiles of csrc/kernels_sep.hpp rely on: run on the REFLECT101-padded image, the separable arithmetic
    reproduces -- bit for bit, inside the original frame -- what it computes on the image itself (where REFLECT101 acts on
    the zero-stuffed expand grid and on Q), provided the far edges have even sizes; an odd far edge does not."""
    from oracle import oracle as orc
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    P = 12
    pad = np.pad(img, ((P, P), (P, P), (0, 0)), mode="reflect")
    out = []
    for im in (img, pad):
        so = orc.StreamingOracle(im.shape[0], im.shape[1], np.uint8, arith="separable", levels=1)
        g = so.push_frame(im)
        out.append((so.best_e[0].copy(), g[1].copy()))
    (e, g1), (ep, g1p) = out
    assert np.array_equal(g1p[P // 2:P // 2 + g1.shape[0], P // 2:P // 2 + g1.shape[1]], g1)
    assert np.array_equal(ep[P:P + h, P:P + w], e) == exact


def test_fixed_point_gaussian_blur_restatement():
    """cv2.GaussianBlur on 8- / 16-bit images as oracle/align_oracle.c restates it (OpenCV's bit-exact fixed-point path,
    [from memory], the blur of align.py:249): taps sum to exactly 1.0 in fixed point, are symmetric, the centre takes the
    remainder of the error diffusion; the blurred image stays within one count of the float64 Gaussian blur; a constant
    image stays constant (sum of taps == 1.0 and round-half-up)."""
    from oracle import oracle as orc
    for ks, sg, bits in [(21, 50.0, 8), (21, 50.0, 16), (5, 1.0, 8), (21, 3.0, 8), (31, 12.5, 16), (1, 2.0, 8)]:
        k = orc.gauss_kernel_fixed(ks, sg, bits).astype(np.int64)
        assert k.sum() == 1 << bits and np.array_equal(k, k[::-1]) and (k >= 0).all()
        x = np.arange(ks) - (ks - 1) / 2
        ideal = np.exp(-x * x / (2 * sg * sg))
        ideal /= ideal.sum()
        assert np.abs(k / float(1 << bits) - ideal).max() <= 1.5 / (1 << bits)
    assert list(orc.gauss_kernel_fixed(5, 1.0, 8)) == [14, 62, 104, 62, 14]
    rng = np.random.default_rng(4)
    for dt, hi in ((np.uint8, 256), (np.uint16, 65536)):
        img = rng.integers(0, hi, (37, 53, 3)).astype(dt)
        got = orc.gaussian_blur_fixed(img, 21, 50.0).astype(np.float64)
        x = np.arange(21) - 10.0
        k = np.exp(-x * x / 5000.0)
        k /= k.sum()
        pad = np.pad(img.astype(np.float64), ((10, 10), (10, 10), (0, 0)), mode="reflect")
        rows = sum(k[i] * pad[:, i:i + 53] for i in range(21))
        want = sum(k[i] * rows[i:i + 37] for i in range(21))
        assert np.abs(got - want).max() <= (1.5 if dt == np.uint8 else 160.0)   # the taps have 8 / 16 fractional bits
        flat = np.full((30, 40, 3), hi - 1, dt)
        assert np.array_equal(orc.gaussian_blur_fixed(flat, 21, 50.0), flat)
