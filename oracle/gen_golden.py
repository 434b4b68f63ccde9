"""oracle/gen_golden.py -- freeze golden vectors under tests/golden/.

Run ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Every expected output below is produced by the reference's OWN modules
(pyramid.py, stack.py, stack_framework.py, core/framework.py) imported through
oracle/ref_import.py, with oracle.py's cv2 primitives as the shim.  The script
also asserts that oracle.RefShaped and oracle.StreamingOracle reproduce those
outputs bit for bit before writing anything, so a fixture on disk certifies
reference control flow == both restatements.

Fixtures are data only (inputs + expected outputs); no reference source text.
"""
import io
import json
import os
import shutil
import sys
import types
import zlib

import numpy as np

from . import oracle as orc
from . import ref_import as ri

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def _check_restatements(frames, out_ref, det, use_fma=True, **kw):
    """RefShaped and StreamingOracle must equal the reference run."""
    if kw.get("float_type") == "float-64":
        kw = {k: v for k, v in kw.items() if k != "float_type"}
        rs = orc.RefShaped(use_fma=use_fma, float_type=np.float64, **kw)
        out_rs, d = rs.stack(frames, want_detail=True)
        assert np.array_equal(out_rs, out_ref), "RefShaped(float64) final != reference"
        for a, b in zip(d["fused"], det["fused"]):
            assert a.dtype == np.float64 and np.array_equal(a, b), "RefShaped(float64) fused pyramid != reference"
        return d, None, None
    rs = orc.RefShaped(use_fma=use_fma, **kw)
    out_rs, d = rs.stack(frames, want_detail=True)
    assert np.array_equal(out_rs, out_ref), "RefShaped final != reference"
    for a, b in zip(d["fused"], det["fused"]):
        assert np.array_equal(a, b), "RefShaped fused pyramid != reference"
    h, w = frames[0].shape[:2]
    so = orc.StreamingOracle(h, w, frames[0].dtype, use_fma=use_fma,
                             min_size=kw.get("min_size", 32),
                             kernel_size=kw.get("kernel_size", 5),
                             gen_kernel=kw.get("gen_kernel", 0.4))
    gs = [so.push_frame(f) for f in frames]
    assert so.levels == len(det["fused"]) - 1
    assert np.array_equal(so.finish(), out_ref), "StreamingOracle final != reference"
    for lv in range(so.levels):
        assert np.array_equal(so.best_lap[lv], det["fused"][lv])
        assert np.array_equal(so.best_idx[lv], d["best"][lv])
    assert np.array_equal(so.fused_base(), det["fused"][-1])
    return d, so, gs


def fusion_case(name, frames, store_pyramids=False, use_fma=True, **kw):
    out_ref, det = ri.reference_stack(frames, use_fma=use_fma, exact_log=True, **kw)
    d, so, gs = _check_restatements(frames, out_ref, det, use_fma, **kw)
    arrays = {"frames": np.stack(frames), "final": out_ref, "collapsed": det["collapsed"],
              "params": np.array(json.dumps({"use_fma": use_fma, **kw}))}
    nl = len(det["fused"]) - 1
    arrays["levels"] = np.array(nl)
    for lv in range(nl):
        arrays[f"fused_{lv}"] = det["fused"][lv]
        arrays[f"best_{lv}"] = d["best"][lv].astype(np.int16)
        arrays[f"energy_{lv}"] = d["energy"][lv]
    arrays["fused_base"] = det["fused"][-1]
    arrays["base_idx_e"] = d["be"].astype(np.int16)
    arrays["base_idx_d"] = d["bd"].astype(np.int16)
    arrays["base_ent"] = d["ent"]
    arrays["base_dev"] = d["dev"]
    if store_pyramids:
        for f, pyr in enumerate(det["pyramids"]):
            for lv, a in enumerate(pyr):
                arrays[f"lap_f{f}_l{lv}"] = a
            for lv in range(1, nl + 1):
                arrays[f"gauss_f{f}_l{lv}"] = gs[f][lv]
    _save(name, **arrays)
    return out_ref


def primitive_cases():
    """G5: border behaviour of reduce / expand on impulse images, through the
    reference's reduce_layer / expand_layer."""
    mod = ri.load_pyramid_module()
    algo = mod.PyramidStack()
    arrays = {}
    for (h, w) in [(8, 8), (9, 7), (6, 11)]:
        for (py, px) in [(0, 0), (h - 1, w - 1), (1, w - 2), (h // 2, w // 2)]:
            img = np.zeros((h, w, 3), np.float32)
            img[py, px] = (1.0, 2.0, 3.0)
            key = f"{h}x{w}_{py}_{px}"
            arrays["in_" + key] = img
            arrays["reduce_" + key] = algo.reduce_layer(img)
            arrays["expand_" + key] = algo.expand_layer(img)
    rng = np.random.default_rng(5)
    img = (rng.random((13, 10, 3)) * 255).astype(np.float32)
    arrays["in_rand"] = img
    arrays["reduce_rand"] = algo.reduce_layer(img)
    arrays["expand_rand"] = algo.expand_layer(img)
    # both restatements agree with the reference functions
    k = orc.k25_f32()
    for key in [k_[3:] for k_ in arrays if k_.startswith("in_")]:
        src = np.ascontiguousarray(arrays["in_" + key])
        h, w = src.shape[:2]
        red = np.empty(((h + 1) // 2, (w + 1) // 2, 3), np.float32)
        orc.lib().orc_reduce_f32(src, h, w, 3, k, red, 1)
        assert np.array_equal(red, arrays["reduce_" + key]), key
        ex = np.empty((2 * h, 2 * w, 3), np.float32)
        orc.lib().orc_expand_f32(src, h, w, 3, k, 2 * h, 2 * w, ex, 1)
        assert np.array_equal(ex, arrays["expand_" + key]), key
    _save("g5_primitives", **arrays)


def base_case():
    """G7: the base-level rule on hand-checkable sizes, several window sizes."""
    rng = np.random.default_rng(7)
    arrays = {}
    for ks in (3, 5, 7):
        for dtype, hi in ((np.uint8, 256), (np.uint16, 65536)):
            mod = ri.load_pyramid_module(exact_log=True)
            algo = mod.PyramidStack(kernel_size=ks)
            algo.dtype = dtype
            algo.num_pixel_values = hi
            imgs = (rng.random((3, 7, 9, 3)) * (hi - 1)).astype(np.float32)
            if dtype == np.uint8:
                imgs[1] = np.round(imgs[1] / 16) * 16  # few distinct levels, ties
            fused = algo.get_fused_base(imgs)
            tag = f"k{ks}_{np.dtype(dtype).name}"
            arrays["in_" + tag] = imgs
            arrays["fused_" + tag] = fused
            rs = orc.RefShaped(kernel_size=ks)
            f2, be, bd, ent, dev = rs.fuse_base(list(imgs), dtype)
            assert np.array_equal(f2, fused), tag
            arrays["ent_" + tag] = ent
            arrays["dev_" + tag] = dev
            # C restatement
            for i in range(3):
                e = np.empty((7, 9), np.float32)
                d = np.empty((7, 9), np.float32)
                orc.lib().orc_base_features_f32(np.ascontiguousarray(imgs[i]), 7, 9, hi,
                                                (ks - 1) // 2, e, d, 1)
                assert np.array_equal(e, ent[i]) and np.array_equal(d, dev[i]), (tag, i)
    _save("g7_base", **arrays)


# ---------------------------------------------------------------------------
# config-1 plumbing: the reference's StackJob / FocusStack / FocusStackBunch on
# real image content (a crop of examples/input/img-jpg, frozen as PNG).
# ---------------------------------------------------------------------------
def _install_io_shim(cv2):
    from PIL import Image

    def imread(path, flags=None):
        im = Image.open(path)
        a = np.array(im)
        if a.ndim == 3:
            a = a[:, :, ::-1]  # RGB -> BGR like cv2
        return np.ascontiguousarray(a)

    def imwrite(path, img, params=None):
        a = img[:, :, ::-1] if img.ndim == 3 else img
        Image.fromarray(np.ascontiguousarray(a)).save(path)
        return True

    cv2.imread = imread
    cv2.imwrite = imwrite


def make_crop_inputs():
    from PIL import Image
    src_dir = "/root/reference/examples/input/img-jpg"
    dst_dir = os.path.join(OUT, "img_jpg_crop")
    os.makedirs(dst_dir, exist_ok=True)
    names = sorted(os.listdir(src_dir))
    for n in names:
        im = Image.open(os.path.join(src_dir, n))
        a = np.array(im)
        y0, x0 = 500, 800
        crop = a[y0:y0 + 256, x0:x0 + 384]
        Image.fromarray(crop).save(os.path.join(dst_dir, n.replace(".jpg", ".png")),
                                   optimize=True)
    return dst_dir, [n.replace(".jpg", ".png") for n in names]


def plumbing_case():
    dst_dir, names = make_crop_inputs()
    ri.load_pyramid_module(exact_log=True)
    cv2 = sys.modules["cv2"]
    _install_io_shim(cv2)
    for stub in ("shinestacker.algorithms.exif", "shinestacker.algorithms.denoise"):
        m = types.ModuleType(stub)
        m.copy_exif_from_file_to_file = lambda *a, **k: None
        m.denoise = lambda img, *a, **k: img
        sys.modules[stub] = m
    import importlib
    cfg = importlib.import_module("shinestacker.config.config").config
    try:
        cfg.init(DISABLE_TQDM=True)
    except Exception:
        pass
    sf = importlib.import_module("shinestacker.algorithms.stack_framework")
    st = importlib.import_module("shinestacker.algorithms.stack")
    pyr = sys.modules["shinestacker.algorithms.pyramid"]

    work = "/tmp/_golden_work"
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "input"))
    for n in names:
        shutil.copy(os.path.join(dst_dir, n), os.path.join(work, "input", n))

    trace = []

    def cb(key):
        def _f(*args):
            trace.append([key] + [a if isinstance(a, (int, str)) else str(a) for a in args])
            return None
        return _f
    callbacks = {k: cb(k) for k in ("before_action", "after_action", "step_counts",
                                    "begin_steps", "end_steps", "after_step", "save_plot",
                                    "check_running")}
    job = sf.StackJob("job", work, input_path="input", callbacks=callbacks)
    job.add_action(st.FocusStack("stack-pyramid", pyr.PyramidStack(), output_path="out-stack",
                                 prefix="pyr_"))
    job.run()
    out_files = sorted(os.listdir(os.path.join(work, "out-stack")))
    from PIL import Image
    stack_img = np.array(Image.open(os.path.join(work, "out-stack", out_files[0])))[:, :, ::-1]
    trace_stack = list(trace)
    trace.clear()

    job = sf.StackJob("job", work, input_path="input", callbacks=callbacks)
    job.add_action(st.FocusStackBunch("bunches", pyr.PyramidStack(), output_path="out-bunch",
                                      frames=3))
    job.run()
    bunch_files = sorted(os.listdir(os.path.join(work, "out-bunch")))
    bunch_imgs = [np.array(Image.open(os.path.join(work, "out-bunch", f)))[:, :, ::-1]
                  for f in bunch_files]
    trace_bunch = list(trace)

    # the in-memory oracle agrees with what the reference job wrote
    frames = [cv2.imread(os.path.join(work, "input", n)) for n in names]
    rs = orc.RefShaped()
    assert np.array_equal(rs.stack(frames), stack_img)

    bunches = {f"{n}_{fr}_{ov}": st.get_bunches(list(range(n)), fr, ov)
               for (n, fr, ov) in [(6, 3, 2), (1024, 10, 2), (10, 10, 2), (3, 10, 2), (7, 4, 1)]}
    with open(os.path.join(OUT, "plumbing.json"), "w") as fh:
        json.dump({"input_names": names, "stack_out_files": out_files,
                   "bunch_out_files": bunch_files, "trace_stack": trace_stack,
                   "trace_bunch": trace_bunch, "get_bunches": bunches,
                   "levels": {"4000x6000": int(np.log2(4000 / 32)),
                              "5760x8640": int(np.log2(5760 / 32)),
                              "1300x2000": int(np.log2(1300 / 32)),
                              "825x1280": int(np.log2(825 / 32))}}, fh, indent=1)
    _save("plumbing_outputs", stack=np.ascontiguousarray(stack_img),
          **{f"bunch_{i}": np.ascontiguousarray(b) for i, b in enumerate(bunch_imgs)})
    shutil.rmtree(work, ignore_errors=True)


def balance_case():
    """The reference's own correction classes (balance.py) on small frames: histograms, corrections,
    look-up tables and corrected frames for LUMI / RGB x LINEAR / GAMMA / MATCH_HIST, uint8 and uint16,
    with and without sub-sampling / mask.  cv2.LUT / split / merge are exact in the shim; BGR2GRAY and
    the INTER_AREA resize are the restatements of oracle.py (parity unpinned for those two)."""
    bal = ri.load_balance_module()
    rng = np.random.default_rng(11)
    arrays, meta = {}, []
    k = 0
    for dtype, hi in ((np.uint8, 256), (np.uint16, 65536)):
        yy, xx = np.mgrid[0:48, 0:64]
        base = (0.25 + 0.5 * (np.sin(xx / 9.0) * np.cos(yy / 7.0) * 0.5 + 0.5))[..., None] * \
            np.array([0.9, 1.0, 0.8])
        ref = np.clip(base * hi + rng.normal(0, hi / 40, base.shape), 0, hi - 1).astype(dtype)
        mov = np.clip((base ** 1.15) * hi * 0.85 + rng.normal(0, hi / 40, base.shape), 0, hi - 1).astype(dtype)
        for channel, cls in (("LUMI", bal.LumiCorrection), ("RGB", bal.RGBCorrection), ("HSV", bal.SVCorrection),
                             ("HLS", bal.LSCorrection)):
            if channel in ("HSV", "HLS") and dtype != np.uint8:
                continue   # cv2.cvtColor has no 16-bit HSV / HLS: the reference raises there
            for cmap in ("LINEAR", "GAMMA", "MATCH_HIST"):
                for opts in ({"subsample": 1}, {"subsample": 2, "fast_subsampling": True},
                             {"subsample": 4, "fast_subsampling": False, "mask_size": 0.8},
                             {"subsample": 1, "intensity_interval": {"min": 10, "max": hi // 2}}):
                    if cmap == "MATCH_HIST" and "intensity_interval" in opts:
                        continue
                    if dtype == np.uint16 and (opts != {"subsample": 1} or (channel == "RGB" and cmap != "MATCH_HIST")):
                        continue   # 16-bit tables are 128 KiB each: a few cases keep the fixture small
                    corr = cls(corr_map=cmap, **opts)
                    corr.begin(ref, 2, 0)
                    hist_ref = np.stack(corr.get_hist(corr.preprocess(ref), 0))
                    hist_mov = np.stack(corr.get_hist(corr.preprocess(mov), 1))
                    out = corr.apply_correction(1, mov.copy())
                    c = corr.corr_map.correction(list(hist_mov))
                    luts = np.stack([corr.corr_map.lut(c[i], corr.corr_map.reference[i])
                                     for i in range(corr.channels)])
                    tag = f"c{k}"
                    arrays[f"{tag}_hist_ref"] = hist_ref.astype(np.int32)
                    arrays[f"{tag}_hist_mov"] = hist_mov.astype(np.int32)
                    arrays[f"{tag}_luts"] = luts
                    arrays[f"{tag}_out"] = out
                    arrays[f"{tag}_size"] = np.asarray(corr.corrections[1], dtype=np.float64)
                    meta.append({"tag": tag, "dtype": np.dtype(dtype).name, "channel": channel,
                                 "corr_map": cmap, "opts": opts})
                    # the NumPy restatement of the device steps agrees with the reference run
                    o = {"subsample": opts.get("subsample", 1), "fast": opts.get("fast_subsampling", False),
                         "mask_size": opts.get("mask_size", 0)}
                    if channel in ("HSV", "HLS"):
                        to, back = ((orc.CVT_BGR2HSV, orc.CVT_HSV2BGR) if channel == "HSV" else (orc.CVT_BGR2HLS, orc.CVT_HLS2BGR))
                        pre = orc.cvt_color_u8(mov, to)
                        arrays[f"{tag}_pre"] = pre
                        assert np.array_equal(orc.balance_hist(pre, False, **o)[1:], hist_mov)
                        ident = np.arange(hi).astype(dtype)[None]
                        assert np.array_equal(orc.cvt_color_u8(orc.apply_lut(pre, np.concatenate([ident, luts])), back), out)
                    else:
                        assert np.array_equal(orc.balance_hist(mov, channel == "LUMI", **o), hist_mov)
                        assert np.array_equal(orc.apply_lut(mov, luts), out)
                    k += 1
        arrays[f"ref_{np.dtype(dtype).name}"] = ref
        arrays[f"mov_{np.dtype(dtype).name}"] = mov
    arrays["meta"] = np.array(json.dumps(meta))
    _save("balance", **arrays)


def f64_case():
    """float_type='float-64' (base_stack_algo.py:14-17): float64 pyramids and base features."""
    rng = np.random.default_rng(64)
    frames = [rng.integers(0, 256, (70, 93, 3), dtype=np.uint8) for _ in range(3)]
    frames[1][20:50, 30:70] = frames[0][20:50, 30:70] // 2 + 60
    fusion_case("g9_f64", frames, min_size=16, float_type="float-64")
    frames16 = [rng.integers(0, 65536, (41, 67, 3), dtype=np.uint16) for _ in range(3)]
    fusion_case("g9_f64_u16", frames16, min_size=8, float_type="float-64", use_fma=False)


def nolevels_case():
    """Frames smaller than 2*min_size: levels = int(log2(40/32)) = 0 (pyramid.py:165), the pyramid is the
    base alone -- entropy/deviation fusion of the frames themselves, then clip/abs/cast."""
    rng = np.random.default_rng(8)
    frames = [rng.integers(0, 256, (40, 52, 3), dtype=np.uint8) for _ in range(3)]
    frames[2][10:30, 5:40] = frames[0][10:30, 5:40] // 2
    fusion_case("g8_nolevels", frames)


def _dm_frames(rng, n, h, w, dtype):
    """A textured scene whose sharp region moves from frame to frame (so the weights differ)."""
    top = 255 if dtype == np.uint8 else 65535
    yy, xx = np.mgrid[0:h, 0:w]
    scene = rng.random((h, w, 3)) * 0.6 + 0.2 * np.sin(xx / 3.0)[..., None] + 0.2 * np.cos(yy / 4.0)[..., None]
    out = []
    for i in range(n):
        focus = np.exp(-((xx - w * (i + 0.5) / n) / (w / n)) ** 2)[..., None]
        smooth = (scene + np.roll(scene, 1, 0) + np.roll(scene, 1, 1) + np.roll(scene, -1, 0) + np.roll(scene, -1, 1)) / 5
        img = focus * scene + (1 - focus) * smooth
        out.append(np.clip(img * top, 0, top).astype(dtype))
    return out


def depth_map_case():
    """The reference's DepthMapStack.focus_stack itself (depth_map.py:64-123) over the cv2 shim; the streaming
    restatement oracle/depth_map_oracle.depth_map_stack must reproduce every recording bit for bit."""
    from . import depth_map_oracle as dmo
    rng = np.random.default_rng(61)
    cases = [
        ("dm_default_u8", np.uint8, (5, 45, 70), {}),
        ("dm_max_u8", np.uint8, (4, 37, 51), {"map_type": "max"}),
        ("dm_sobel_l4_u16", np.uint16, (3, 64, 49), {"energy": "sobel", "levels": 4}),
        ("dm_nosmooth_k3_u8", np.uint8, (3, 33, 40), {"smooth_size": 0, "kernel_size": 3, "blur_size": 3}),
        ("dm_max_t05_l1_u16", np.uint16, (3, 30, 44), {"map_type": "max", "temperature": 0.5, "levels": 1,
                                                       "smooth_size": 5}),
        ("dm_k7_b9_u8", np.uint8, (3, 40, 40), {"kernel_size": 7, "blur_size": 9, "smooth_size": 9, "levels": 2}),
        ("dm_f64_default_u8", np.uint8, (4, 45, 70), {"float_type": "float-64"}),
        ("dm_f64_max_nosmooth_u16", np.uint16, (3, 37, 51), {"float_type": "float-64", "map_type": "max",
                                                             "smooth_size": 0, "blur_size": 9}),
        ("dm_f64_sobel_l4_u16", np.uint16, (3, 64, 49), {"float_type": "float-64", "energy": "sobel", "levels": 4,
                                                         "map_type": "max", "temperature": 0.3}),
    ]
    store = {}
    meta = {}
    for name, dt, (n, h, w), kw in cases:
        frames = _dm_frames(rng, n, h, w, dt)
        if name == "dm_default_u8":
            frames[1][:, :20] = frames[0][:, :20] = 60   # a flat patch in every frame: zero energy, zero total
            for f in frames[2:]:
                f[:, :20] = 60
        out, trace = ri.reference_depth_map(frames, **kw)
        mine = dmo.depth_map_stack(frames, **kw)
        flat = None
        if name == "dm_default_u8":
            # where every energy is 0 the reference's weights are uninitialised memory (np.divide where=)
            flat = np.zeros(out.shape[:2], bool)
            flat[:, :20] = True   # zero totals up to column 8, spread by two pyrDown / pyrUp rounds to column 18
            assert np.array_equal(out[~flat], mine[~flat]), name
        else:
            assert np.array_equal(out, mine), name
        store[name + "_frames"] = np.stack(frames)
        store[name + "_out"] = out
        if flat is not None:
            store[name + "_undefined"] = flat
        meta[name] = {"kwargs": kw, "trace_len": len(trace),
                      "trace_head": [list(map(str, t)) for t in trace[:4]]}
        print("  ", name, out.shape, out.dtype, "oracle == reference run")
    np.savez_compressed(os.path.join(OUT, "depth_map.npz"), **store)
    with open(os.path.join(OUT, "depth_map.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


ALIGN_CASES = [
    # name, dtype, (h, w), true transform, scene kwargs, alignment_config, feature_config, matching_config
    ("rigid_sub2_area_blur_u8", np.uint8, (120, 160), [[0.9995, 0.0314, 3.0], [-0.0314, 0.9995, -5.0]],
     {"block": 2}, {"min_good_matches": 10}, None, None),                                   # the defaults of align.py
    ("rigid_sub2_fast_replicate_u8", np.uint8, (121, 163), [[1.002, -0.01, -4.0], [0.01, 1.002, 6.0]],
     {"block": 1}, {"min_good_matches": 10, "fast_subsampling": True, "border_mode": "BORDER_REPLICATE"}, None, None),
    ("rigid_sub1_constant_u8", np.uint8, (96, 130), [[1.0, 0.02, 7.0], [-0.02, 1.0, 2.0]],
     {"block": 1}, {"subsample": 1, "border_mode": "BORDER_CONSTANT", "border_value": [10, 200, 30, 0]}, None, None),
    ("rigid_sub2_area_blur_u16", np.uint16, (100, 140), [[0.998, 0.02, -6.0], [-0.02, 0.998, 4.0]],
     {"block": 2}, {"min_good_matches": 10, "border_blur": 20}, None, None),
    ("homography_sub1_blur_u8", np.uint8, (110, 150), [[1.0, 0.02, 4.0], [-0.015, 1.0, 2.0], [4e-5, -2e-5, 1.0]],
     {"block": 1}, {"subsample": 1, "transform": "ALIGN_HOMOGRAPHY"}, None, None),
    ("homography_sub2_area_replicate_u8", np.uint8, (120, 160), [[1.01, 0.0, -3.0], [0.01, 0.99, 5.0], [-3e-5, 2e-5, 1.0]],
     {"block": 2}, {"min_good_matches": 10, "transform": "ALIGN_HOMOGRAPHY", "border_mode": "BORDER_REPLICATE"}, None, None),
    ("homography_sub2_fast_constant_u16", np.uint16, (101, 141), [[1.0, 0.01, 5.0], [-0.01, 1.0, -3.0], [2e-5, 1e-5, 1.0]],
     {"block": 1}, {"min_good_matches": 10, "transform": "ALIGN_HOMOGRAPHY", "fast_subsampling": True,
                    "border_mode": "BORDER_CONSTANT", "border_value": [1000, 60000, 3, 0]}, None, None),
    ("rigid_retry_without_subsampling_u8", np.uint8, (120, 160), [[1.0, 0.0, 4.0], [0.0, 1.0, -6.0]],
     {"block": 1, "parity": (1, 1)}, {"min_good_matches": 10, "fast_subsampling": True}, None, None),
    ("rigid_too_few_matches_u8", np.uint8, (64, 80), [[1.0, 0.0, 2.0], [0.0, 1.0, 2.0]],
     {"block": 2, "n": 2}, {"min_good_matches": 1}, None, None),
    ("homography_three_matches_u8", np.uint8, (64, 80), [[1.0, 0.0, 2.0], [0.0, 1.0, 2.0], [0.0, 0.0, 1.0]],
     {"block": 2, "n": 3}, {"min_good_matches": 1, "transform": "ALIGN_HOMOGRAPHY"}, None, None),
    ("rigid_sub4_area_ragged_blur_u8", np.uint8, (123, 162), [[1.0, 0.01, 8.0], [-0.01, 1.0, -4.0]],
     {"block": 4, "grid": 4}, {"min_good_matches": 5, "subsample": 4}, None, None),
    ("rigid_orb_hamming_lmeds_u8", np.uint8, (96, 128), [[1.0, 0.0, -3.0], [0.0, 1.0, 5.0]],
     {"block": 2}, {"min_good_matches": 10, "align_method": "LMEDS", "border_mode": "BORDER_REPLICATE"},
     {"detector": "ORB", "descriptor": "ORB"}, {"match_method": "NORM_HAMMING"}),
]


def align_case():
    """The reference's OWN `align_images` (align.py:154-252) run here through ref_import.load_align_module: estimator
    calls on the stand-in (oracle/cv2_standin.py), numeric cv2 calls on oracle.py's raw primitives.  Asserts, before
    freezing anything, that the restatements the GPU tests compare with -- oracle.warp_affine / warp_perspective
    (warp + `valid` rule + composite in one call) -- reproduce what the reference's own lines (mask warp, cvtColor,
    `mask == 0`, GaussianBlur, argument order, float32 cast of the rescaled rigid matrix, getPerspectiveTransform
    conjugation) produced, bit for bit."""
    from . import cv2_standin as cs
    arrays, meta = {}, []
    for name, dtype, (h, w), m_true, skw, acfg, fcfg, mcfg in ALIGN_CASES:
        log = []
        al = ri.load_align_module(log)
        mov, ref, src, _dst = cs.marker_scene(m_true, h=h, w=w, dtype=dtype, texture=True, seed=len(meta) + 5,
                                              **{"n": 30, **skw})
        trace = []
        callbacks = {k: (lambda *a, _k=k: trace.append([_k] + [str(x) for x in a]))
                     for k in ("message", "matches_message", "align_message", "ecc_message", "blur_message", "warning",
                               "save_plot")}
        n_good, m, img_warp = al.align_images(ref, mov.copy(), feature_config=fcfg, matching_config=mcfg,
                                              alignment_config=acfg, callbacks=callbacks)
        cfg = {**al._DEFAULT_ALIGNMENT_CONFIG, **(acfg or {})}
        calls = [e[0] for e in log]
        entry = {"name": name, "alignment_config": acfg, "feature_config": fcfg, "matching_config": mcfg,
                 "n_good_matches": int(n_good), "callbacks": trace,
                 "standin_calls": [c for c in calls if c in ("detectAndCompute", "detect", "flann", "bf",
                                                             "estimateAffinePartial2D", "findHomography")],
                 "fit_args": [[str(x) for x in e[1:]] for e in log if e[0] in ("estimateAffinePartial2D", "findHomography")]}
        arrays[f"{name}_mov"], arrays[f"{name}_ref"] = mov, ref
        if m is None:
            assert img_warp is None
            entry["aligned"] = False
        else:
            entry["aligned"] = True
            entry["m_dtype"] = str(m.dtype)
            mode = {"BORDER_CONSTANT": orc.BORDER_CONSTANT, "BORDER_REPLICATE": orc.BORDER_REPLICATE,
                    "BORDER_REPLICATE_BLUR": orc.BORDER_REPLICATE_BLUR}[cfg["border_mode"]]
            fn = orc.warp_perspective if m.shape == (3, 3) else orc.warp_affine
            mine = fn(mov, m, mode, cfg["border_value"], 21, cfg["border_blur"])
            assert np.array_equal(mine, img_warp), f"{name}: oracle apply restatement != the reference's align_images"
            arrays[f"{name}_m"], arrays[f"{name}_warp"] = m, img_warp
            filled = int((orc.warp_affine(mov, m, want_mask=True)[1] == 0).sum()) if m.shape == (2, 3) else \
                int((orc.warp_perspective(mov, m, want_mask=True)[1] == 0).sum())
            entry["out_of_frame_pixels"] = filled
        meta.append(entry)
        print("  ", name, "n_good", n_good, "aligned" if m is not None else "not aligned", entry["standin_calls"])
    arrays["meta"] = np.array(json.dumps(meta))
    _save("align", **arrays)


def depth_map_steps_case():
    """The reference's DepthMapStack step methods (depth_map.py:28-62: get_sobel_map, get_laplacian_map, smooth_energy,
    get_focus_map -- public, and called by its own tests) run here over the cv2 shim on small gray stacks, float-32 and
    float-64, both map types; the GPU test compares `shinestacker_amd.DepthMapStack`'s methods of the same names."""
    import importlib
    ri.load_pyramid_module()
    mod = importlib.import_module("shinestacker.algorithms.depth_map")
    mod.np = ri._NumpyWithExactExp()
    rng = np.random.default_rng(77)
    arrays, meta = {}, []
    for tag, ft, shape, kw in (("f32", "float-32", (3, 45, 70), {}),
                               ("f32_k3", "float-32", (2, 40, 52), {"kernel_size": 3, "blur_size": 3, "smooth_size": 9}),
                               ("f64", "float-64", (3, 37, 51), {}),
                               ("f64_max_nosmooth", "float-64", (3, 33, 40), {"map_type": "max", "smooth_size": 0, "temperature": 0.3}),
                               ("f32_max", "float-32", (4, 30, 44), {"map_type": "max", "temperature": 0.5})):
        algo = mod.DepthMapStack(float_type=ft, **kw)
        n, h, w = shape
        yy, xx = np.mgrid[0:h, 0:w]
        gray = np.stack([np.clip(120 + 60 * np.sin(xx / (3.0 + i)) * np.cos(yy / (4.0 + i)) + rng.normal(0, 12, (h, w)), 0, 255)
                         .astype(np.uint8) for i in range(n)]).astype(algo.float_type)
        sob = algo.get_sobel_map(gray)
        lap = algo.get_laplacian_map(gray)
        en = lap / lap.max()
        sm = algo.smooth_energy(en)
        fm = algo.get_focus_map(sm)
        arrays.update({f"{tag}_gray": gray, f"{tag}_sobel": sob, f"{tag}_laplacian": lap, f"{tag}_energy": en,
                       f"{tag}_smoothed": sm, f"{tag}_focus": fm})
        meta.append({"tag": tag, "float_type": ft, "kwargs": kw})
        print("  ", tag, sob.dtype, lap.dtype, sm.dtype, fm.dtype)
    arrays["meta"] = np.array(json.dumps(meta))
    _save("depth_map_steps", **arrays)


def pyramid_steps_case():
    """The reference's PyramidStack methods one at a time (pyramid.py:24-148: convolve, reduce_layer, expand_layer,
    process_single_image, fuse_laplacian, get_fused_base, fuse_pyramids, collapse) on a small stack, over the cv2 shim; the
    GPU test compares `shinestacker_amd.PyramidStack`'s methods of the same names with these recordings."""
    rng = np.random.default_rng(99)
    arrays = {}
    for tag, dtype, (n, h, w), levels, use_fma in (("u8", np.uint8, (3, 44, 61), 2, True), ("u16", np.uint16, (3, 37, 50), 2, True),
                                                   ("u8_nofma", np.uint8, (2, 40, 40), 1, False)):
        mod = ri.load_pyramid_module(use_fma=use_fma, exact_log=True)
        algo = mod.PyramidStack(min_size=8)
        algo.process = ri.FakeProcess()
        algo.dtype = dtype
        hi = 256 if dtype == np.uint8 else 65536
        algo.num_pixel_values, algo.max_pixel_value = hi, hi - 1
        yy, xx = np.mgrid[0:h, 0:w]
        frames = []
        for i in range(n):
            f = (0.5 + 0.4 * np.sin(xx / (2.5 + i)) * np.cos(yy / (3.0 + i)))[..., None] * np.array([0.9, 1.0, 0.8]) * (hi - 1)
            frames.append(np.clip(f + rng.normal(0, hi / 30, f.shape), 0, hi - 1).astype(dtype))
        f32 = frames[0].astype(np.float32)
        arrays[f"{tag}_frames"] = np.stack(frames)
        arrays[f"{tag}_convolve3"] = algo.convolve(f32)
        arrays[f"{tag}_convolve1"] = algo.convolve(np.ascontiguousarray(f32[..., 1]))
        arrays[f"{tag}_reduce3"] = algo.reduce_layer(f32)
        arrays[f"{tag}_reduce1"] = algo.reduce_layer(np.ascontiguousarray(f32[..., 0]))
        arrays[f"{tag}_expand3"] = algo.expand_layer(f32)
        arrays[f"{tag}_expand1"] = algo.expand_layer(np.ascontiguousarray(f32[..., 2]))
        pyrs = [algo.process_single_image(f, levels) for f in frames]
        for i, p in enumerate(pyrs):
            for lv, a in enumerate(p):
                arrays[f"{tag}_pyr{i}_{lv}"] = a
        arrays[f"{tag}_fuse_lap0"] = algo.fuse_laplacian(np.stack([p[0] for p in pyrs], axis=0))
        arrays[f"{tag}_fused_base"] = algo.get_fused_base(np.stack([p[-1] for p in pyrs], axis=0))
        fused = algo.fuse_pyramids(pyrs)
        for lv, a in enumerate(fused):
            arrays[f"{tag}_fused{lv}"] = a
        arrays[f"{tag}_collapsed"] = algo.collapse(fused)
        arrays[f"{tag}_meta"] = np.array(json.dumps({"levels": levels, "use_fma": use_fma, "dtype": np.dtype(dtype).name, "n": n}))
        print("  ", tag, [a.shape for a in fused], arrays[f"{tag}_collapsed"].dtype)
    _save("pyramid_steps", **arrays)


def api_case():
    """The drop-in boundary as the reference declares it: constructor / function signatures -- parameter
    names, order, kinds and defaults -- of the classes the mirrors stand in for, read from the reference's OWN modules with
    `inspect` and frozen as tests/golden/api_surface.json (tests/test_host_logic.py compares the mirrors with it)."""
    import importlib
    import inspect
    al = ri.load_align_module()
    sf = importlib.import_module("shinestacker.algorithms.stack_framework")
    st = importlib.import_module("shinestacker.algorithms.stack")
    bal = importlib.import_module("shinestacker.algorithms.balance")
    pyr = sys.modules["shinestacker.algorithms.pyramid"]
    dm = importlib.import_module("shinestacker.algorithms.depth_map")
    objs = {"StackJob": sf.StackJob, "FocusStack": st.FocusStack, "FocusStackBunch": st.FocusStackBunch,
            "CombinedActions": sf.CombinedActions, "AlignFrames": al.AlignFrames, "BalanceFrames": bal.BalanceFrames,
            "PyramidStack": pyr.PyramidStack, "DepthMapStack": dm.DepthMapStack, "align_images": al.align_images,
            "detect_and_compute": al.detect_and_compute, "get_good_matches": al.get_good_matches,
            "find_transform": al.find_transform, "validate_align_config": al.validate_align_config,
            "get_bunches": st.get_bunches, "img_subsample": importlib.import_module("shinestacker.algorithms.utils").img_subsample}
    out = {}
    for name, obj in objs.items():
        params = []
        for k, v in inspect.signature(obj).parameters.items():
            params.append({"name": k, "kind": v.kind.name,
                           "default": None if v.default is inspect.Parameter.empty else repr(v.default),
                           "has_default": v.default is not inspect.Parameter.empty})
        entry = {"params": params}
        if inspect.isclass(obj):
            entry["protocol"] = sorted(n for n in ("name", "steps_per_frame", "focus_stack", "print_message", "run", "run_core",
                                                   "run_frame", "run_step", "begin", "end", "add_action", "init", "callback",
                                                   "sub_message", "sub_message_r", "img_ref", "process_frame", "align_images",
                                                   "convolve", "reduce_layer", "expand_layer", "process_single_image",
                                                   "fuse_laplacian", "get_fused_base", "fuse_pyramids", "collapse",
                                                   "get_sobel_map", "get_laplacian_map", "smooth_energy", "get_focus_map",
                                                   "time", "set_terminator", "folder_list_str")
                                       if callable(getattr(obj, n, None)))
        out[name] = entry
    with open(os.path.join(OUT, "api_surface.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("  wrote api_surface.json", {k: len(v["params"]) for k, v in out.items()})


def main():
    assert ri.available(), "needs /root/reference"
    os.makedirs(OUT, exist_ok=True)
    orc.build()
    if "--only-align" in sys.argv:
        align_case()
        return
    if "--only-api" in sys.argv:
        api_case()
        return
    if "--only-pyramid-steps" in sys.argv:
        pyramid_steps_case()
        return
    if "--only-depth-map-steps" in sys.argv:
        depth_map_steps_case()
        return
    if "--only-depth-map" in sys.argv:
        depth_map_case()
        return
    if "--only-nolevels" in sys.argv:
        nolevels_case()
        return
    if "--only-f64" in sys.argv:
        f64_case()
        return
    if "--only-balance" in sys.argv:
        balance_case()
        return
    rng = np.random.default_rng(20250824)
    print("G1 u8 odd sizes")
    fusion_case("g1_u8", [rng.integers(0, 256, (67, 101, 3), dtype=np.uint8) for _ in range(4)],
                store_pyramids=True, min_size=8)
    print("G2 u16")
    fusion_case("g2_u16", [rng.integers(0, 65536, (45, 70, 3), dtype=np.uint16)
                           for _ in range(3)], min_size=8)
    print("G3 ties")
    const = np.full((40, 48, 3), 77, np.uint8)
    a = rng.integers(0, 256, (40, 48, 3), dtype=np.uint8)
    fusion_case("g3_ties", [const, a, a.copy(), const.copy()], min_size=8)
    print("G1b non-fma arithmetic, gen_kernel 0.3, kernel_size 3")
    fusion_case("g1b_nofma", [rng.integers(0, 256, (50, 37, 3), dtype=np.uint8)
                              for _ in range(3)], use_fma=False, min_size=8, gen_kernel=0.3,
                kernel_size=3)
    print("G1c smooth content (near-tie energies), default min_size")
    yy, xx = np.mgrid[0:96, 0:130]
    smooth = []
    for f in range(5):
        blur = 1.0 + 3.0 * abs(f - 2)
        img = 128 + 100 * np.sin(xx / blur / 2.0) * np.cos(yy / blur / 3.0)
        smooth.append(np.repeat(img[:, :, None], 3, 2).clip(0, 255).astype(np.uint8))
    fusion_case("g1c_smooth", smooth)
    print("G8 no Laplacian levels")
    nolevels_case()
    print("G9 float-64")
    f64_case()
    print("balance")
    balance_case()
    print("depth map stacker")
    depth_map_case()
    print("alignment (the reference's align_images)")
    align_case()
    print("API surface")
    api_case()
    print("depth map steps")
    depth_map_steps_case()
    print("pyramid steps")
    pyramid_steps_case()
    print("G5 primitives")
    primitive_cases()
    print("G7 base")
    base_case()
    print("config-1 plumbing")
    plumbing_case()
    crcs = {}
    for f in (0, 127, 255):
        fr = orc.synth_frame_numpy(64, 96, f, 256)
        assert np.array_equal(fr, orc.synth_frame_u8(64, 96, f, 256))
        crcs[str(f)] = zlib.crc32(fr.tobytes())
    with open(os.path.join(OUT, "synth_crc.json"), "w") as fh:
        json.dump({"h": 64, "w": 96, "n": 256, "seed": 20250824, "crc32": crcs}, fh)
    print("done")


if __name__ == "__main__":
    main()
