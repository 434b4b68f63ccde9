"""GPU: parity of the HIP path (through the C ABI) with the oracle and the golden vectors.
Bit-exact: `np.array_equal` on float32 tensors (the only slack is -0.0 == +0.0)."""
import json
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN, f64_cases, fusion_cases, load_golden, stack_kwargs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    hiplib.require_device()
    return hiplib


def impls(L):
    return [L.IMPL_SIMPLE, L.IMPL_TILED]


def run_stack(L, frames, impl, **kw):
    fr0 = frames[0]
    st = L.Stack(fr0.shape[0], fr0.shape[1], in_dtype=fr0.dtype, impl=impl, **kw)
    for f in frames:
        st.push_frame(f)
    return st


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("case", fusion_cases())
def test_golden_fusion(L, case, impl):
    g = load_golden(case)
    kw = stack_kwargs(g["params"])
    st = run_stack(L, list(g["frames"]), impl, **kw)
    assert st.levels == int(g["levels"])
    for lv in range(st.levels):
        assert np.array_equal(st.tap(L.TAP_ENERGY, lv), g[f"energy_{lv}"]), f"energy {lv}"
        assert np.array_equal(st.tap(L.TAP_INDEX, lv), g[f"best_{lv}"]), f"index {lv}"
        assert np.array_equal(st.tap(L.TAP_FUSED_LAP, lv), g[f"fused_{lv}"]), f"lap {lv}"
    assert np.array_equal(st.tap(L.TAP_BASE_IDX_E), g["base_idx_e"])
    assert np.array_equal(st.tap(L.TAP_BASE_IDX_D), g["base_idx_d"])
    assert np.array_equal(st.tap(L.TAP_BASE_ENT), g["base_ent"].max(axis=0))
    assert np.array_equal(st.tap(L.TAP_BASE_DEV), g["base_dev"].max(axis=0))
    out = st.finish()
    assert np.array_equal(st.tap(L.TAP_FUSED_BASE), g["fused_base"])
    assert np.array_equal(st.tap(L.TAP_COLLAPSED), g["collapsed"])
    assert out.dtype == g["final"].dtype and np.array_equal(out, g["final"])
    st.close()


@pytest.mark.parametrize("case", f64_cases())
def test_golden_fusion_float64(L, case):
    """float_type='float-64' against the reference's own float-64 run: float64 Laplacians / base / collapse,
    float32 energies, float64 entropy and deviation."""
    g = load_golden(case)
    fr = g["frames"]
    st = L.Stack(fr.shape[1], fr.shape[2], in_dtype=fr.dtype, float_type=L.MI_F64, **stack_kwargs(g["params"]))
    for f in fr:
        st.push_frame(f)
    assert st.levels == int(g["levels"])
    for lv in range(st.levels):
        assert np.array_equal(st.tap(L.TAP_ENERGY, lv), g[f"energy_{lv}"]), f"energy {lv}"
        assert np.array_equal(st.tap(L.TAP_INDEX, lv), g[f"best_{lv}"]), f"index {lv}"
        lap = st.tap(L.TAP_FUSED_LAP, lv)
        assert lap.dtype == np.float64 and np.array_equal(lap, g[f"fused_{lv}"]), f"lap {lv}"
    assert np.array_equal(st.tap(L.TAP_BASE_IDX_E), g["base_idx_e"])
    assert np.array_equal(st.tap(L.TAP_BASE_IDX_D), g["base_idx_d"])
    assert np.array_equal(st.tap(L.TAP_BASE_ENT), g["base_ent"].max(axis=0))
    assert np.array_equal(st.tap(L.TAP_BASE_DEV), g["base_dev"].max(axis=0))
    out = st.finish()
    assert np.array_equal(st.tap(L.TAP_FUSED_BASE), g["fused_base"])
    assert np.array_equal(st.tap(L.TAP_COLLAPSED), g["collapsed"])
    assert out.dtype == g["final"].dtype and np.array_equal(out, g["final"])
    st.close()


@pytest.mark.parametrize("shape,dtype,n,kw", [
    ((90, 131), np.uint16, 4, {"min_size": 16}),
    ((40, 52), np.uint8, 3, {}),                         # no Laplacian levels
    ((77, 64), np.uint8, 2, {"min_size": 8, "kernel_size": 3, "use_fma": False}),
])
def test_float64_random_vs_ref_shaped(L, oracle, shape, dtype, n, kw):
    from shinestacker_amd.pyramid import PyramidStack
    rng = np.random.default_rng(shape[0] * 3 + n)
    hi = 256 if dtype == np.uint8 else 65536
    frames = [rng.integers(0, hi, shape + (3,)).astype(dtype) for _ in range(n)]
    want = oracle.RefShaped(float_type=np.float64, **kw).stack(frames)
    got = PyramidStack(float_type="float-64", **kw).focus_stack_arrays(frames)
    assert got.dtype == want.dtype and np.array_equal(got, want)
    # and it is not the float-32 result in general (different rounding of the pyramid)
    f32 = PyramidStack(**kw).focus_stack_arrays(frames)
    assert f32.shape == got.shape


@pytest.mark.parametrize("impl", [1, 2])
def test_golden_gaussian_levels_per_frame(L, impl):
    g = load_golden("g1_u8")
    kw = stack_kwargs(g["params"])
    fr = g["frames"]
    st = L.Stack(fr.shape[1], fr.shape[2], in_dtype=fr.dtype, impl=impl, batch_frames=1, **kw)
    for f in range(len(fr)):
        st.push_frame(fr[f])
        for lv in range(1, st.levels + 1):
            assert np.array_equal(st.tap(L.TAP_GAUSS, lv), g[f"gauss_f{f}_l{lv}"]), (f, lv)
    st.close()


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("shape,dtype,n,kw", [
    ((389, 517), np.uint8, 5, {}),
    ((256, 384), np.uint16, 4, {}),
    ((130, 67), np.uint8, 3, {"min_size": 4}),          # deep pyramid, odd sizes everywhere
    ((300, 444), np.uint8, 3, {"gen_kernel": 0.5, "kernel_size": 7}),
    ((211, 199), np.uint16, 3, {"use_fma": False, "min_size": 16}),
    ((64, 1030), np.uint8, 2, {}),                       # one level, ragged tile edge
])
def test_seeded_random_vs_streaming_oracle(L, oracle, impl, shape, dtype, n, kw):
    rng = np.random.default_rng(hash((shape, n)) % (2 ** 32))
    hi = 256 if dtype == np.uint8 else 65536
    # low-pass + noise so that selection is not pure noise
    frames = []
    for f in range(n):
        base = rng.integers(0, hi, (shape[0] // 8 + 2, shape[1] // 8 + 2, 3))
        up = np.kron(base, np.ones((8, 8, 1)))[:shape[0], :shape[1]]
        noise = rng.integers(-hi // 16, hi // 16, shape + (3,))
        frames.append(np.clip(up + noise * (f + 1) // n, 0, hi - 1).astype(dtype))
    so = oracle.StreamingOracle(shape[0], shape[1], dtype, **kw)
    for f in frames:
        so.push_frame(f)
    want = so.finish()
    st = run_stack(L, frames, impl, **kw)
    assert st.levels == so.levels
    for lv in range(st.levels):
        assert np.array_equal(st.tap(L.TAP_INDEX, lv), so.best_idx[lv]), f"index {lv}"
        assert np.array_equal(st.tap(L.TAP_ENERGY, lv), so.best_e[lv]), f"energy {lv}"
        assert np.array_equal(st.tap(L.TAP_FUSED_LAP, lv), so.best_lap[lv]), f"lap {lv}"
    got = st.finish()
    assert np.array_equal(st.tap(L.TAP_FUSED_BASE), so.fused_base())
    assert np.array_equal(got, want)
    st.close()


@pytest.mark.parametrize("impl", [1, 2])
def test_f32_input_equals_u8_input(L, impl):
    """config 2 feeds fp32 frames holding integer values (img.astype(float32), pyramid.py:126)."""
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (200, 300, 3), dtype=np.uint8) for _ in range(3)]
    a = run_stack(L, frames, impl)
    out_a = a.finish()
    st = L.Stack(200, 300, in_dtype=np.float32, out_dtype=np.uint8, impl=impl)
    for f in frames:
        st.push_frame(f.astype(np.float32))
    assert np.array_equal(st.finish(), out_a)


@pytest.mark.parametrize("impl", [1, 2])
def test_reset_and_reuse_handle(L, impl):
    rng = np.random.default_rng(12)
    fa = [rng.integers(0, 256, (96, 128, 3), dtype=np.uint8) for _ in range(3)]
    fb = [rng.integers(0, 256, (96, 128, 3), dtype=np.uint8) for _ in range(2)]
    st = run_stack(L, fa, impl)
    out_a = st.finish()
    with pytest.raises(L.DeviceError):
        st.push_frame(fa[0])  # push after finish is a state error
    st.reset()
    for f in fb:
        st.push_frame(f)
    out_b = st.finish()
    ref = run_stack(L, fb, impl)
    assert np.array_equal(out_b, ref.finish())
    assert not np.array_equal(out_a[:8], out_b[:8])


@pytest.mark.parametrize("impl", [1, 2])
def test_duplicate_frame_is_a_noop(L, impl):
    """Size-independent property of first-max selection: appending a copy of an
    earlier frame never changes the result."""
    rng = np.random.default_rng(13)
    frames = [rng.integers(0, 256, (150, 220, 3), dtype=np.uint8) for _ in range(4)]
    a = run_stack(L, frames, impl).finish()
    b = run_stack(L, frames + [frames[1].copy(), frames[3].copy()], impl).finish()
    assert np.array_equal(a, b)


def test_device_resident_frames_and_synth_generator(L, oracle):
    """push_frames_device on frames made by the device generator == host path on the
    NumPy generator (bit-exact generator + same result)."""
    H, W, N = 192, 256, 6
    per = H * W * 3
    buf = L.DeviceBuffer(per * N)
    L.synth_frames_device(buf.ptr, np.uint8, H, W, 0, N, N)
    dev_frames = buf.download((N, H, W, 3), np.uint8)
    for f in range(N):
        assert np.array_equal(dev_frames[f], oracle.synth_frame_numpy(H, W, f, N))
    outs = []
    for impl in impls(L):
        st = L.Stack(H, W, impl=impl)
        st.push_frames_device(buf.ptr, N)
        outs.append(st.finish())
        idx0 = st.tap(L.TAP_INDEX, 0)
        st.close()
    assert np.array_equal(outs[0], outs[1])
    host = run_stack(L, list(dev_frames), L.IMPL_SIMPLE).finish()
    assert np.array_equal(host, outs[0])
    # built-in sanity check of the generator: frame f is sharp in band f
    band = (np.arange(H) * N // H)[:, None]
    assert (idx0 == band).mean() > 0.9


def test_stack_job_on_gpu_matches_reference_outputs(L, tmp_path):
    """config 1 plumbing: StackJob + FocusStack / FocusStackBunch with the HIP PyramidStack
    reproduce the files and callback trace the reference's own job produced."""
    from shinestacker_amd import FocusStack, FocusStackBunch, PyramidStack, StackJob
    from shinestacker_amd.imageio import read_img
    with open(os.path.join(GOLDEN, "plumbing.json")) as fh:
        gold = json.load(fh)
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "input"))
    for n in gold["input_names"]:
        shutil.copy(os.path.join(GOLDEN, "img_jpg_crop", n), os.path.join(work, "input", n))
    trace = []

    def cb(key):
        return lambda *a: trace.append([key] + [x if isinstance(x, (int, str)) else str(x)
                                                for x in a])
    keys = ("before_action", "after_action", "step_counts", "begin_steps", "end_steps",
            "after_step", "save_plot", "check_running")
    job = StackJob("job", work, input_path="input", callbacks={k: cb(k) for k in keys})
    job.add_action(FocusStack("stack-pyramid", PyramidStack(arith="exact"), output_path="out-stack",
                              prefix="pyr_"))
    job.add_action(FocusStackBunch("bunches", PyramidStack(arith="exact"), input_path="input",
                                   output_path="out-bunch", frames=3))
    job.run()
    outs = load_golden("plumbing_outputs")
    got = read_img(os.path.join(work, "out-stack", gold["stack_out_files"][0]))
    assert np.array_equal(got, outs["stack"])
    files = sorted(os.listdir(os.path.join(work, "out-bunch")))
    assert files == gold["bunch_out_files"]
    for i, f in enumerate(files):
        assert np.array_equal(read_img(os.path.join(work, "out-bunch", f)), outs[f"bunch_{i}"])
    norm = lambda tr: [[x.replace(work, "<W>").replace("/tmp/_golden_work", "<W>")
                        if isinstance(x, str) else x for x in t] for t in tr]
    n_stack = len(gold["trace_stack"]) - 1  # golden ends with the job's own after_action
    assert norm(trace[:n_stack]) == norm(gold["trace_stack"][:n_stack])


def test_default_arithmetic_on_the_reference_example_job(L, tmp_path):
    """The one-import swap as a user gets it: `PyramidStack()` with no arguments (constants.DEFAULT_PY_ARITH = "separable")
    in the same StackJob, against the files the reference's own job wrote (tests/golden/plumbing_outputs.npz: real image
    content, crops of examples/input/img-jpg).  Same files, same callback trace; pixel values within the stated
    tolerance: a value differs from the recording by at most ONE count, and fewer than 0.05 % of the values do (the
    truncating cast at an integer boundary, pyramid.py:179) -- measured: 18 of 294 912 in the full stack."""
    from shinestacker_amd import FocusStack, FocusStackBunch, PyramidStack, StackJob
    from shinestacker_amd.imageio import read_img
    with open(os.path.join(GOLDEN, "plumbing.json")) as fh:
        gold = json.load(fh)
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "input"))
    for n in gold["input_names"]:
        shutil.copy(os.path.join(GOLDEN, "img_jpg_crop", n), os.path.join(work, "input", n))
    assert PyramidStack().arith == "separable"
    job = StackJob("job", work, input_path="input")
    job.add_action(FocusStack("stack-pyramid", PyramidStack(), output_path="out-stack", prefix="pyr_"))
    job.add_action(FocusStackBunch("bunches", PyramidStack(), input_path="input", output_path="out-bunch", frames=3))
    job.run()
    outs = load_golden("plumbing_outputs")
    pairs = [(read_img(os.path.join(work, "out-stack", gold["stack_out_files"][0])), outs["stack"])]
    files = sorted(os.listdir(os.path.join(work, "out-bunch")))
    assert files == gold["bunch_out_files"]
    pairs += [(read_img(os.path.join(work, "out-bunch", f)), outs[f"bunch_{i}"]) for i, f in enumerate(files)]
    for got, want in pairs:
        assert got.dtype == want.dtype and got.shape == want.shape
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 5e-4, (int(d.max()), float((d != 0).mean()))


def test_large_frame_properties(L, oracle):
    """BASELINE config-2 geometry (4000x6000, 6 levels + 63x94 base) with a short stack:
    properties that hold at any size + an oracle check on a cropped corner region."""
    H, W, N = 4000, 6000, 3
    per = H * W * 3
    buf = L.DeviceBuffer(per * N)
    L.synth_frames_device(buf.ptr, np.uint8, H, W, 0, N, N)
    res = {}
    for impl in impls(L):
        st = L.Stack(H, W, impl=impl)
        assert st.levels == 6 and st.shapes[-1] == (63, 94)
        st.push_frames_device(buf.ptr, N)
        res[impl] = (st.finish(), st.tap(L.TAP_INDEX, 0), st.tap(L.TAP_ENERGY, 3))
        st.close()
    a, b = res[L.IMPL_SIMPLE], res[L.IMPL_TILED]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    band = (np.arange(H) * N // H)[:, None]
    assert (a[1] == band).mean() > 0.9
    # level-0 selection of the top-left corner depends only on a bounded neighbourhood:
    # the oracle on a 256x256 crop agrees away from the crop's own borders.
    frames = [oracle.synth_frame_numpy(H, W, f, N)[:256, :256] for f in range(N)]
    so = oracle.StreamingOracle(256, 256, np.uint8, levels=1)
    for f in frames:
        so.push_frame(f)
    assert np.array_equal(so.best_idx[0][:200, :200], a[1][:200, :200])
    assert np.array_equal(so.best_e[0][:200, :200],
                          L_energy_crop(L, buf, H, W, N)[:200, :200])


def L_energy_crop(L, buf, H, W, N):
    st = L.Stack(H, W, impl=L.IMPL_TILED)
    st.push_frames_device(buf.ptr, N)
    e = st.tap(L.TAP_ENERGY, 0)
    st.close()
    return e[:256, :256]


def test_sharded_state_combine_on_one_gpu(L):
    """Two handles play two ranks (frame blocks [0,3) and [3,6)); their state, combined with
    mi_combine_select in rank order, equals one handle that saw all six frames -- including a
    duplicate frame that sits in the second block (the first copy must win)."""
    import ctypes as C
    rng = np.random.default_rng(21)
    frames = [rng.integers(0, 256, (160, 224, 3), dtype=np.uint8) for _ in range(5)]
    frames.insert(4, frames[1].copy())
    whole = run_stack(L, frames, L.IMPL_TILED)
    parts = []
    for r, blk in enumerate((frames[:3], frames[3:])):
        st = L.Stack(160, 224, impl=L.IMPL_TILED)
        st.set_first_index(3 * r)
        for f in blk:
            st.push_frame(f)
        parts.append(st)
    lib = L.load()
    for level in range(whole.levels + 2):
        ptrs = [p.state_ptrs(level) for p in parts]
        n = ptrs[0][3]
        cand_e, cand_l, cand_i = L.DeviceBuffer(2 * n * 4), L.DeviceBuffer(2 * n * 12), L.DeviceBuffer(2 * n * 4)
        host = []
        for r, (e, l, i, _n) in enumerate(ptrs):
            for src, dst, sz in ((e, cand_e, n * 4), (l, cand_l, n * 12), (i, cand_i, n * 4)):
                tmp = np.empty(sz, np.uint8)
                L.check(lib.mi_memcpy_d2h(0, tmp.ctypes.data, src, sz))
                dst.upload(tmp, r * sz)
        out_e, out_l, out_i = L.DeviceBuffer(n * 4), L.DeviceBuffer(n * 12), L.DeviceBuffer(n * 4)
        L.check(lib.mi_combine_select(0, None, 2, cand_e.ptr, cand_l.ptr, cand_i.ptr, n,
                                      out_e.ptr, out_l.ptr, out_i.ptr))
        L.check(lib.mi_device_synchronize(0))
        we, wl, wi, _ = whole.state_ptrs(level)
        for got, want_ptr, sz, dt in ((out_e, we, n, np.float32), (out_l, wl, 3 * n, np.float32),
                                      (out_i, wi, n, np.int32)):
            want = np.empty(sz, dt)
            L.check(lib.mi_memcpy_d2h(0, want.ctypes.data, want_ptr, want.nbytes))
            assert np.array_equal(got.download((sz,), dt), want), f"level {level}"
    for p in parts:
        p.close()
    whole.close()


def test_combiner_under_rccl_world_1(tmp_path):
    """Combiner.combine() (flat all-levels exchange over torch.distributed / RCCL) in a fresh process
    -- torch's HIP runtime has to be loaded before libmi355stack.so -- with a world of one rank: the
    exchange, the HIP select and the copy back must leave the stack's result unchanged."""
    import subprocess
    import sys
    script = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
torch.cuda.init()
sys.path.insert(0, os.getcwd())
from shinestacker_amd import _lib as L
from shinestacker_amd.multigpu import Combiner
dist.init_process_group("nccl", rank=0, world_size=1)
rng = np.random.default_rng(5)
frames = [rng.integers(0, 256, (200, 296, 3), dtype=np.uint8) for _ in range(5)]
ref = L.Stack(200, 296)
for f in frames: ref.push_frame(f)
want = ref.finish()
st = L.Stack(200, 296)
st.set_first_index(0)
for f in frames: st.push_frame(f)
Combiner(st).combine()
got = st.finish()
assert np.array_equal(got, want), int((got != want).sum())
for lv in range(st.levels):
    assert np.array_equal(st.tap(L.TAP_INDEX, lv), ref.tap(L.TAP_INDEX, lv))
dist.destroy_process_group()
print("COMBINE_OK")
'''
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "COMBINE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bunches_of_16bit_tiff_frames(L, oracle, tmp_path):
    """BASELINE config 5 in miniature: 16-bit frames on disk, FocusStackBunch (frames=5, overlap=2) reusing
    one stacker handle for every bunch, frames decoded on the host and pushed through the pinned async
    upload path -- every bunch's output file equals the oracle's stack of that bunch, bit for bit."""
    from shinestacker_amd import FocusStackBunch, PyramidStack, StackJob, get_bunches
    from shinestacker_amd.imageio import read_img, write_img
    rng = np.random.default_rng(50)
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "in16"))
    base = rng.integers(0, 65536, (136, 200, 3)).astype(np.uint16)
    frames, names = [], []
    for f in range(12):
        fr = base.copy()
        y0 = 10 * f
        band = fr[y0:y0 + 30]
        band[...] = rng.integers(0, 65536, band.shape)   # a band of fresh detail per frame
        fr[: max(y0 - 5, 0)] //= 3
        frames.append(fr)
        names.append(f"f{f:02d}.tif")
        write_img(os.path.join(work, "in16", names[-1]), fr)
        assert np.array_equal(read_img(os.path.join(work, "in16", names[-1])), fr)   # lossless 16-bit round trip
    job = StackJob("job", work, input_path="in16")
    job.add_action(FocusStackBunch("bunches", PyramidStack(min_size=16, batch_frames=4, arith="exact"), output_path="out16",
                                   frames=5, overlap=2))
    job.run()
    outs = sorted(os.listdir(os.path.join(work, "out16")))
    chunks = get_bunches(list(range(12)), 5, 2)
    assert len(outs) == len(chunks) >= 3
    for fname, idx in zip(outs, chunks):
        so = oracle.StreamingOracle(136, 200, np.uint16, min_size=16, keep_gauss=False)
        for k in idx:
            so.push_frame(frames[k])
        got = read_img(os.path.join(work, "out16", fname))
        assert got.dtype == np.uint16 and np.array_equal(got, so.finish()), fname


def test_decode_ahead_keeps_order_output_and_error_position(L, tmp_path):
    """PyramidStack(decode_threads=N) decodes files on a thread pool ahead of the GPU: same fused image,
    same callback sequence as the sequential loop, and a bad file stops the stack at ITS position."""
    from shinestacker_amd import PyramidStack
    from shinestacker_amd.errors import ShapeError
    from shinestacker_amd.imageio import write_img
    rng = np.random.default_rng(3)
    names = []
    for i in range(9):
        names.append(str(tmp_path / f"im{i:02d}.png"))
        write_img(names[-1], rng.integers(0, 256, (96, 160, 3)).astype(np.uint8))

    class Proc:
        id, name = 7, "p"

        def __init__(self):
            self.trace = []

        def callback(self, key, *a):
            self.trace.append((key,) + a)
            return True

        def sub_message_r(self, *_a, **_k):
            pass

    outs, traces = [], []
    for nthreads in (1, 4):
        algo = PyramidStack(decode_threads=nthreads)
        algo.process = Proc()
        algo.do_step_callback = True
        outs.append(algo.focus_stack(names))
        traces.append(algo.process.trace)
    assert np.array_equal(outs[0], outs[1]) and traces[0] == traces[1]
    assert [t[3] for t in traces[1] if t[0] == "after_step"] == list(range(18))
    # a frame of another shape at index 5
    write_img(names[5], rng.integers(0, 256, (96, 128, 3)).astype(np.uint8))
    algo = PyramidStack(decode_threads=4)
    algo.process = Proc()
    algo.do_step_callback = True
    with pytest.raises(ShapeError):
        algo.focus_stack(names)
    assert [t[3] for t in algo.process.trace if t[0] == "after_step"] == [0, 1, 2, 3, 4]


def test_frames_beyond_24_bit_pixel_counts(L, oracle):
    """108-megapixel frames: level 1 alone has more than 2^24 pixels, which the 24-bit index
    multiplies of the kernels must not see (they are used for in-tile / per-row arithmetic only)."""
    H, W = 9000, 12040
    frames = [oracle.synth_frame_u8(H, W, f, 2) for f in range(2)]
    so = oracle.StreamingOracle(H, W, np.uint8, keep_gauss=False)
    st = L.Stack(H, W)
    for f in frames:
        so.push_frame(f)
        st.push_frame(f)
    want = so.finish()
    assert st.levels == so.levels == 8
    for lv in (0, 1, 2):
        assert np.array_equal(st.tap(L.TAP_ENERGY, lv), so.best_e[lv]), lv
        assert np.array_equal(st.tap(L.TAP_FUSED_LAP, lv), so.best_lap[lv]), lv
    assert np.array_equal(st.finish(), want)
    st.close()
    with pytest.raises(Exception):
        L.Stack(20000, 20000)   # 400 MP: beyond the 32-bit in-frame addressing, rejected at create


def test_bunches_sharded_over_two_ranks_on_one_gpu(L, tmp_path):
    """SURVEY 8(e) bunch mode with the HIP stacker: the two ranks' blocks of bunches (run one after the other on
    this GPU) give the files of the single-process job, bit for bit."""
    from shinestacker_amd import FocusStackBunch, PyramidStack, StackJob
    from shinestacker_amd.imageio import read_img, write_img
    rng = np.random.default_rng(21)
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "in"))
    for i in range(11):
        write_img(os.path.join(work, "in", f"f{i:02d}.png"), rng.integers(0, 256, (80, 112, 3)).astype(np.uint8))

    def run(shard, out):
        job = StackJob("job", work, input_path="in")
        job.add_action(FocusStackBunch("b", PyramidStack(min_size=16), output_path=out, frames=4, overlap=1,
                                       shard=shard))
        job.run()
    run(None, "single")
    for rank in (0, 1):
        run((rank, 2), "sharded")
    names = sorted(os.listdir(os.path.join(work, "single")))
    assert names == sorted(f for f in os.listdir(os.path.join(work, "sharded")) if not f.startswith(".")) and len(names) == 4
    for f in names:
        assert np.array_equal(read_img(os.path.join(work, "single", f)), read_img(os.path.join(work, "sharded", f)))


@pytest.mark.parametrize("shape,dtype", [((4101, 4303), np.uint8), ((4210, 4097), np.uint16)])
def test_large_coarse_levels_on_the_wide_tile(L, oracle, shape, dtype):
    """Frames of 17 MP and more: level 1 (> 4 MP) runs on level 0's 32x64 tile configuration (level_fused_coarse) --
    odd sizes at every level, state and image against the streaming oracle."""
    H, W = shape
    scale = 1 if dtype == np.uint8 else 257
    frames = [(oracle.synth_frame_numpy(H, W, f, 2).astype(dtype) * scale).astype(dtype) for f in range(2)]
    so = oracle.StreamingOracle(H, W, dtype, keep_gauss=False)
    st = L.Stack(H, W, in_dtype=dtype)
    for f in frames:
        so.push_frame(f)
        st.push_frame(f)
    want = so.finish()
    assert st.levels == so.levels == 7
    for lv in range(so.levels):
        assert np.array_equal(st.tap(L.TAP_ENERGY, lv), so.best_e[lv]), lv
        assert np.array_equal(st.tap(L.TAP_FUSED_LAP, lv), so.best_lap[lv]), lv
    assert np.array_equal(st.finish(), want)
    st.close()


def test_pinned_frames_are_uploaded_without_the_bounce_copy_and_give_the_same_stack(L, oracle):
    """mi_stack_push_frame_pinned (BASELINE config 5 is upload-bound): frames that lie in pinned host memory -- `host_alloc`
    arrays or arrays pinned in place with `host_register` -- go to the device straight from the caller's buffer; the
    fused image equals the bounce-copy path's and the oracle's; pageable memory is refused by the pinned entry point; a
    producer cycling two pinned buffers throttles itself with wait_uploads."""
    import ctypes as C
    H, W, N = 300, 452, 7
    frames = [oracle.synth_frame_numpy(H, W, f, N) for f in range(N)]
    so = oracle.StreamingOracle(H, W, np.uint8)
    for f in frames:
        so.push_frame(f)
    want = so.finish()
    st = L.Stack(H, W, in_dtype=np.uint8, batch_frames=4)
    pinned = [L.host_alloc((H, W, 3), np.uint8) for _ in range(N)]
    for p, f in zip(pinned, frames):
        p[...] = f
        assert L.is_pinned(p) and not L.is_pinned(f)
        st.push_frame(p, zero_copy=True)
    assert len(st._inflight) == N
    st.wait_uploads(0)
    assert st._inflight == []
    assert np.array_equal(st.finish(), want)
    # the default is the copying path whatever memory the frame lies in: the caller's buffer is free on return, and a
    # pageable array with zero_copy=True takes the copying path too
    st.reset()
    scratch = pinned[0].copy()
    for f in frames:
        pinned[0][...] = f
        st.push_frame(pinned[0])              # one reused pinned buffer, overwritten right after the call
    assert st._inflight == []
    assert np.array_equal(st.finish(), want)
    pinned[0][...] = scratch
    st.reset()
    for f in frames:
        st.push_frame(f, zero_copy=True)
    assert st._inflight == [] and np.array_equal(st.finish(), want)
    # pageable memory through the pinned entry point: refused, nothing pushed
    st.reset()
    rc = L.load().mi_stack_push_frame_pinned(st._h, frames[0].ctypes.data, 0)
    assert rc == L.MI_ERR_INVALID and b"pinned" in L.load().mi_last_error()
    # an ordinary array pinned in place
    reg = [f.copy() for f in frames]
    for r in reg:
        L.host_register(r)
        st.push_frame(r, zero_copy=True)
    assert np.array_equal(st.finish(), want)
    for r in reg:
        L.host_unregister(r)
        assert not L.is_pinned(r)
    # two pinned buffers cycled by a "decoder"
    st.reset()
    ring = pinned[:2]
    for i, f in enumerate(frames):
        st.wait_uploads(1)          # the buffer about to be overwritten (pushed two frames ago) is on the device
        ring[i % 2][...] = f
        st.push_frame(ring[i % 2], zero_copy=True)
    assert np.array_equal(st.finish(), want)
    st.close()
    del pinned, ring


@pytest.mark.parametrize("tag", ["u8", "u16", "u8_nofma"])
def test_step_methods_equal_the_reference_methods_of_the_same_names(L, tag):
    """SURVEY 8(a) P2-P9 one method at a time: PyramidStack.convolve / reduce_layer / expand_layer / process_single_image /
    fuse_laplacian / get_fused_base / fuse_pyramids / collapse (pyramid.py:24-148) against recordings of the REFERENCE's own
    methods run over the cv2 shim (oracle/gen_golden.py::pyramid_steps_case): 2-D and 3-channel images, odd sizes, 8 and 16
    bit, with and without fused multiply-add.  Bit-equal (-0.0 == +0.0 being the only slack)."""
    from shinestacker_amd import PyramidStack
    z = np.load(os.path.join(GOLDEN, "pyramid_steps.npz"))
    meta = json.loads(str(z[f"{tag}_meta"]))
    dt = np.dtype(meta["dtype"])
    algo = PyramidStack(min_size=8, use_fma=meta["use_fma"])
    algo._set_dtype(dt)
    frames = z[f"{tag}_frames"]
    f32 = frames[0].astype(np.float32)

    def same(got, want, what):
        assert got.dtype == want.dtype and got.shape == want.shape, (what, got.dtype, got.shape, want.dtype, want.shape)
        assert np.array_equal(got, want), (what, float(np.abs(got - want).max()))
    same(algo.convolve(f32), z[f"{tag}_convolve3"], "convolve3")
    same(algo.convolve(np.ascontiguousarray(f32[..., 1])), z[f"{tag}_convolve1"], "convolve1")
    same(algo.reduce_layer(f32), z[f"{tag}_reduce3"], "reduce3")
    same(algo.reduce_layer(np.ascontiguousarray(f32[..., 0])), z[f"{tag}_reduce1"], "reduce1")
    same(algo.expand_layer(f32), z[f"{tag}_expand3"], "expand3")
    same(algo.expand_layer(np.ascontiguousarray(f32[..., 2])), z[f"{tag}_expand1"], "expand1")
    L_, n = meta["levels"], meta["n"]
    pyrs = [algo.process_single_image(frames[i], L_) for i in range(n)]
    for i, p in enumerate(pyrs):
        assert len(p) == L_ + 1
        for lv, a in enumerate(p):
            same(a, z[f"{tag}_pyr{i}_{lv}"], f"pyr{i}_{lv}")
    same(algo.fuse_laplacian(np.stack([p[0] for p in pyrs], axis=0)), z[f"{tag}_fuse_lap0"], "fuse_laplacian")
    same(algo.get_fused_base(np.stack([p[-1] for p in pyrs], axis=0)), z[f"{tag}_fused_base"], "get_fused_base")
    fused = algo.fuse_pyramids(pyrs)
    for lv, a in enumerate(fused):
        same(a, z[f"{tag}_fused{lv}"], f"fused{lv}")
    same(algo.collapse(fused), z[f"{tag}_collapsed"], "collapse")
    # and the whole thing is what focus_stack does: collapse(fuse_pyramids(...)).astype(dtype) == the exact-mode stack
    if L_ == int(np.log2(min(frames.shape[1:3]) / 8)):    # (the recording of u8_nofma was made with fewer levels than focus_stack picks)
        want = PyramidStack(min_size=8, use_fma=meta["use_fma"], arith="exact").focus_stack_arrays(list(frames))
        assert np.array_equal(z[f"{tag}_collapsed"].astype(dt), want)
    algo.close()
