"""CPU: the host-side mirror of the action API (no GPU compute).  The stacker used
here is a test double built on the oracle -- it only exists to drive actions.py."""
import json
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from shinestacker_amd import (BaseStackAlgo, CombinedActions, FocusStack, FocusStackBunch,
                              InvalidOptionError, PyramidStack, RunStopException, StackJob,
                              SubAction, get_bunches)
from shinestacker_amd.errors import BitDepthError, ImageLoadError, ShapeError
from shinestacker_amd.imageio import read_img, validate_image, write_img


class OracleStacker(BaseStackAlgo):
    """Follows the same protocol as PyramidStack.focus_stack, arithmetic by oracle/."""

    def __init__(self, oracle):
        super().__init__("pyramid", 2)
        self.oracle = oracle
        self.do_step_callback = False

    def _step(self, i):
        if self.do_step_callback:
            self.process.callback('after_step', self.process.id, self.process.name, i)
        if self.process.callback('check_running', self.process.id, self.process.name) is False:
            raise RunStopException(self.process.name)

    def focus_stack(self, filenames):
        frames, meta = [], None
        for i, p in enumerate(filenames):
            img, meta, _ = self.read_image_and_update_metadata(p, meta)
            frames.append(img)
            self._step(i)
        for i in range(len(filenames)):
            self._step(i + len(filenames))
        so = self.oracle.StreamingOracle(frames[0].shape[0], frames[0].shape[1],
                                         frames[0].dtype)
        for f in frames:
            so.push_frame(f)
        return so.finish()


@pytest.fixture()
def workdir(tmp_path):
    src = os.path.join(GOLDEN, "img_jpg_crop")
    os.makedirs(tmp_path / "input")
    for n in sorted(os.listdir(src)):
        shutil.copy(os.path.join(src, n), tmp_path / "input" / n)
    return str(tmp_path)


def recorder():
    trace = []

    def cb(key):
        def _f(*args):
            trace.append([key] + [a if isinstance(a, (int, str)) else str(a) for a in args])
        return _f
    keys = ("before_action", "after_action", "step_counts", "begin_steps", "end_steps",
            "after_step", "save_plot", "check_running")
    return trace, {k: cb(k) for k in keys}


def normalise(trace, work):
    out = []
    for t in trace:
        out.append([x.replace(work, "<W>").replace("/tmp/_golden_work", "<W>")
                    if isinstance(x, str) else x for x in t])
    return out


def test_get_bunches_golden():
    with open(os.path.join(GOLDEN, "plumbing.json")) as fh:
        gb = json.load(fh)["get_bunches"]
    for key, want in gb.items():
        n, fr, ov = map(int, key.split("_"))
        assert get_bunches(list(range(n)), fr, ov) == want


def test_focus_stack_job_trace_and_output(oracle, workdir):
    with open(os.path.join(GOLDEN, "plumbing.json")) as fh:
        gold = json.load(fh)
    trace, cbs = recorder()
    job = StackJob("job", workdir, input_path="input", callbacks=cbs)
    job.add_action(FocusStack("stack-pyramid", OracleStacker(oracle), output_path="out-stack",
                              prefix="pyr_"))
    job.run()
    assert sorted(os.listdir(os.path.join(workdir, "out-stack"))) == gold["stack_out_files"]
    assert normalise(trace, workdir) == normalise(gold["trace_stack"], workdir)
    out = read_img(os.path.join(workdir, "out-stack", gold["stack_out_files"][0]))
    assert np.array_equal(out, load_golden("plumbing_outputs")["stack"])


def test_focus_stack_bunch_trace_and_outputs(oracle, workdir):
    with open(os.path.join(GOLDEN, "plumbing.json")) as fh:
        gold = json.load(fh)
    trace, cbs = recorder()
    job = StackJob("job", workdir, input_path="input", callbacks=cbs)
    job.add_action(FocusStackBunch("bunches", OracleStacker(oracle), output_path="out-bunch",
                                   frames=3))
    job.run()
    files = sorted(os.listdir(os.path.join(workdir, "out-bunch")))
    assert files == gold["bunch_out_files"]
    assert normalise(trace, workdir) == normalise(gold["trace_bunch"], workdir)
    outs = load_golden("plumbing_outputs")
    for i, f in enumerate(files):
        assert np.array_equal(read_img(os.path.join(workdir, "out-bunch", f)), outs[f"bunch_{i}"])


def test_cancel_raises_runstop(oracle, workdir):
    calls = {"n": 0}

    def check(*_a):
        calls["n"] += 1
        return calls["n"] < 4
    job = StackJob("job", workdir, input_path="input", callbacks={"check_running": check})
    job.add_action(FocusStack("s", OracleStacker(oracle), output_path="o"))
    with pytest.raises(RunStopException):
        job.run()


def test_output_dir_scratched_and_chained(oracle, workdir):
    os.makedirs(os.path.join(workdir, "o"))
    open(os.path.join(workdir, "o", "stale.png"), "w").close()
    job = StackJob("job", workdir, input_path="input")
    a = FocusStack("s", OracleStacker(oracle), output_path="o")
    job.add_action(a)
    assert os.listdir(os.path.join(workdir, "o")) == []
    assert job.paths[-1] == "o" and a.input_path == "input"
    b = FocusStack("s2", OracleStacker(oracle))
    job.add_action(b)
    assert b.input_path == "o" and b.output_path == "s2"


def test_option_errors():
    with pytest.raises(InvalidOptionError):
        PyramidStack(float_type="float-16")
    assert PyramidStack(float_type="float-64").float_type is np.float64   # base_stack_algo.py:16-17
    with pytest.raises(InvalidOptionError):
        FocusStackBunch("b", PyramidStack(), frames=3, overlap=3)
    algo = PyramidStack()
    assert algo.name() == "pyramid" and algo.steps_per_frame() == 2


def test_image_validation_errors(tmp_path):
    a = np.zeros((8, 9, 3), np.uint8)
    with pytest.raises(ShapeError):
        validate_image(np.zeros((8, 10, 3), np.uint8), a.shape[:2], a.dtype)
    with pytest.raises(BitDepthError):
        validate_image(np.zeros((8, 9, 3), np.uint16), a.shape[:2], a.dtype)
    with pytest.raises(RuntimeError):
        read_img(str(tmp_path / "missing.png"))
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not a png")
    algo = PyramidStack()
    with pytest.raises(ImageLoadError):
        algo.read_image_and_update_metadata(str(bad), None)


def test_image_io_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    a8 = rng.integers(0, 256, (17, 23, 3), dtype=np.uint8)
    a16 = rng.integers(0, 65536, (17, 23, 3), dtype=np.uint16)
    write_img(str(tmp_path / "a.png"), a8)
    write_img(str(tmp_path / "a.tif"), a8)
    write_img(str(tmp_path / "b.tif"), a16)
    assert np.array_equal(read_img(str(tmp_path / "a.png")), a8)
    assert np.array_equal(read_img(str(tmp_path / "a.tif")), a8)
    b = read_img(str(tmp_path / "b.tif"))
    assert b.dtype == np.uint16 and np.array_equal(b, a16)


class Recorder(SubAction):
    def __init__(self):
        super().__init__()
        self.seen = []

    def run_frame(self, idx, ref_idx, img):
        self.seen.append((idx, ref_idx))
        return img


@pytest.mark.parametrize("step_process,want", [
    (False, [(0, 3), (1, 3), (2, 3), (3, 3), (4, 3), (5, 3)]),
    (True, [(3, 3), (4, 3), (5, 4), (2, 3), (1, 2), (0, 1)]),
])
def test_combined_actions_iteration_order(workdir, step_process, want):
    """stack_framework.py:214-232: fixed reference vs. chained neighbours."""
    rec = Recorder()
    job = StackJob("job", workdir, input_path="input")
    job.add_action(CombinedActions("combo", [rec], step_process=step_process,
                                   output_path="aligned"))
    job.run()
    assert rec.seen == want
    assert len(os.listdir(os.path.join(workdir, "aligned"))) == 6


def test_bunches_sharded_over_ranks(oracle, workdir, tmp_path):
    """SURVEY 8(e) bunch mode: whole bunches per process, no collective.  Two ranks (run one after the other
    here) produce exactly the files of the single-process job, each bunch fused once, same 'bunch: NNNN' titles."""
    from shinestacker_amd.actions import shard_steps
    for n, world in ((7, 2), (5, 8), (16, 4), (0, 3)):
        got = [i for r in range(world) for i in shard_steps(n, r, world)]
        assert got == list(range(n))
        sizes = [len(shard_steps(n, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1

    def run(shard, out, cbs=None):
        job = StackJob("job", workdir, input_path="input", callbacks=cbs or {})
        job.add_action(FocusStackBunch("bunches", OracleStacker(oracle), output_path=out, frames=3, overlap=1,
                                       shard=shard))
        job.run()
        return job

    trace1, cbs1 = recorder()
    run(None, "single", cbs1)
    titles1 = sorted(t[2] for t in trace1 if t[0] == "save_plot")
    files1 = sorted(os.listdir(os.path.join(workdir, "single")))
    assert len(files1) >= 2
    titles2 = []
    os.makedirs(os.path.join(workdir, "sharded"))
    open(os.path.join(workdir, "sharded", "stale.png"), "w").close()   # rank 0 (and only rank 0) empties the dir
    for rank in (0, 1):
        tr, cbs = recorder()
        job = run((rank, 2), "sharded", cbs)
        titles2 += [t[2] for t in tr if t[0] == "save_plot"]
        assert job is not None
    files2 = sorted(f for f in os.listdir(os.path.join(workdir, "sharded")) if not f.startswith("."))
    assert files2 == files1 and sorted(titles2) == titles1
    for f in files1:
        assert np.array_equal(read_img(os.path.join(workdir, "single", f)), read_img(os.path.join(workdir, "sharded", f)))
    with pytest.raises(InvalidOptionError):
        FocusStackBunch("b", OracleStacker(oracle), shard=(2, 2))


def test_bunches_sharded_two_processes(workdir):
    """The same with two real processes started the way torch.distributed.run starts ranks (RANK / WORLD_SIZE /
    LOCAL_RANK in the environment, shard='env'): rank 1 waits for rank 0's marker before it writes."""
    import subprocess
    import sys
    script = (
        "import os, sys, time\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import *\n"
        "from oracle import oracle as orc\n"
        "orc.build()\n"
        "from test_host_logic import OracleStacker\n"
        "from shinestacker_amd import FocusStackBunch, StackJob\n"
        "if os.environ['RANK'] == '0': time.sleep(0.5)\n"     # rank 1 must wait for rank 0's scratch
        "job = StackJob('job', %r, input_path='input')\n"
        "algo = OracleStacker(orc); algo.device = -1\n"
        "job.add_action(FocusStackBunch('bunches', algo, output_path='out2p', frames=3, overlap=1, shard='env'))\n"
        "job.run()\n"
        "assert algo.device == int(os.environ['LOCAL_RANK'])\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), workdir)
    os.makedirs(os.path.join(workdir, "out2p"))
    open(os.path.join(workdir, "out2p", "stale.png"), "w").close()
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_PORT="29511")
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    files = sorted(f for f in os.listdir(os.path.join(workdir, "out2p")) if not f.startswith("."))
    n_in = len(os.listdir(os.path.join(workdir, "input")))
    assert len(files) == len(get_bunches(list(range(n_in)), 3, 1)) and "stale.png" not in files


def test_combined_actions_sharded_over_ranks(workdir):
    """SURVEY 8(e) alignment without step_process: frames depend on the reference frame only, so they split over
    processes; every rank sees the same reference index, the union of the ranks' frames is every frame once."""
    seen, files = [], None
    os.makedirs(os.path.join(workdir, "aligned"))
    open(os.path.join(workdir, "aligned", "stale.png"), "w").close()
    for rank in range(3):
        rec = Recorder()
        rec.device = -1
        job = StackJob("job", workdir, input_path="input")
        job.add_action(CombinedActions("combo", [rec], output_path="aligned", shard=(rank, 3)))
        job.run()
        seen += rec.seen
        assert len(rec.seen) == 2 and rec.device == -1
    assert seen == [(0, 3), (1, 3), (2, 3), (3, 3), (4, 3), (5, 3)]
    files = sorted(f for f in os.listdir(os.path.join(workdir, "aligned")) if not f.startswith("."))
    assert files == sorted(os.listdir(os.path.join(workdir, "input")))
    with pytest.raises(InvalidOptionError):
        CombinedActions("combo", [Recorder()], step_process=True, shard=(0, 2))


def test_align_frames_sharded_indexes_by_global_frame(workdir, monkeypatch):
    """AlignFrames inside a sharded CombinedActions sized its per-frame table by the rank's own block and
    indexed it with the global frame index (IndexError on every rank > 0).  CPU form: estimator injected, the device
    apply step replaced by the identity."""
    from shinestacker_amd import AlignFrames
    import shinestacker_amd.align as al
    monkeypatch.setattr(al, "apply_transform", lambda img, m, cfg, device=0: img)
    est = lambda a, b, fc, mc, ac: (500, np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]))
    os.makedirs(os.path.join(workdir, "al_sh"))
    n_in = len(os.listdir(os.path.join(workdir, "input")))
    for rank in range(2):
        act = AlignFrames(estimator=est, subsample=1)
        job = StackJob("job", workdir, input_path="input")
        job.add_action(CombinedActions("combo", [act], output_path="al_sh", shard=(rank, 2)))
        job.run()
        assert len(act.n_matches) == n_in
        blk = range(0, n_in // 2 + n_in % 2) if rank == 0 else range(n_in // 2 + n_in % 2, n_in)
        assert all(act.n_matches[i] == 500 for i in blk if i != n_in // 2)
    assert sorted(f for f in os.listdir(os.path.join(workdir, "al_sh")) if not f.startswith(".")) == \
        sorted(os.listdir(os.path.join(workdir, "input")))


def test_png16_roundtrip_and_all_scanline_filters(tmp_path):
    """without OpenCV a 16-bit PNG lost its depth silently (Pillow has no 16-bit RGB).  The module's own
    zlib codec keeps it (utils.py:11-30: png is read IMREAD_UNCHANGED); a foreign file using every scan-line filter
    decodes to the same pixels; JPEG, which cannot hold 16 bits, raises instead of shifting."""
    import struct
    import zlib
    from shinestacker_amd import imageio as io
    rng = np.random.default_rng(3)
    a = rng.integers(0, 65536, (23, 31, 3)).astype(np.uint16)
    p = str(tmp_path / "a.png")
    io.write_img(p, a)
    b = io.read_img(p)
    assert b.dtype == np.uint16 and np.array_equal(a, b)
    # the same image encoded with filters None/Sub/Up/Average/Paeth in turn (PNG spec 9.2)
    h, w = a.shape[:2]
    be = np.ascontiguousarray(a[:, :, ::-1]).astype(">u2").view(np.uint8).reshape(h, w * 6).astype(np.int64)
    rows, prev, bpp = bytearray(), np.zeros(w * 6, np.int64), 6
    for r in range(h):
        f, cur, out = r % 5, be[r], np.zeros(w * 6, np.int64)
        for i in range(w * 6):
            A, B, Cc = (cur[i - bpp] if i >= bpp else 0), prev[i], (prev[i - bpp] if i >= bpp else 0)
            pa, pb, pc = abs(B - Cc), abs(A - Cc), abs(A + B - 2 * Cc)
            pred = [0, A, B, (A + B) // 2, A if pa <= pb and pa <= pc else (B if pb <= pc else Cc)][f]
            out[i] = (cur[i] - pred) & 255
        rows += bytes([f]) + bytes(out.astype(np.uint8))
        prev = cur

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    q = str(tmp_path / "foreign.png")
    with open(q, "wb") as fh:
        fh.write(io._PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(bytes(rows))) + chunk(b"IEND", b""))
    assert np.array_equal(io.read_img(q), a)
    with pytest.raises(ValueError):
        io.write_img(str(tmp_path / "x.jpg"), a)
    a8 = (a >> 8).astype(np.uint8)
    io.write_img(str(tmp_path / "b.png"), a8)
    assert np.array_equal(io.read_img(str(tmp_path / "b.png")), a8)


def test_mirrors_present_the_reference_api_surface():
    """the one-import swap only works if the mirrors take what the reference's classes take.
    tests/golden/api_surface.json was read from the reference's OWN modules with `inspect` (oracle/gen_golden.py::api_case):
    every positional / keyword parameter of the reference exists in the mirror at the same position with the same
    default (mirrors may ADD keyword-only extensions), the reference's `**kwargs` sinks are kept, and the protocol
    methods the actions call on each other exist."""
    import inspect
    import json
    import os
    import shinestacker_amd as sa
    from shinestacker_amd import actions, align
    with open(os.path.join(os.path.dirname(__file__), "golden", "api_surface.json")) as fh:
        gold = json.load(fh)
    mine = {"StackJob": sa.StackJob, "FocusStack": sa.FocusStack, "FocusStackBunch": sa.FocusStackBunch,
            "CombinedActions": sa.CombinedActions, "AlignFrames": sa.AlignFrames, "BalanceFrames": sa.BalanceFrames,
            "PyramidStack": sa.PyramidStack, "DepthMapStack": sa.DepthMapStack, "align_images": align.align_images,
            "detect_and_compute": align.detect_and_compute, "get_good_matches": align.get_good_matches,
            "find_transform": align.find_transform, "validate_align_config": align.validate_align_config,
            "get_bunches": actions.get_bunches, "img_subsample": align.img_subsample}
    assert sorted(mine) == sorted(gold)
    for name, ref in gold.items():
        sig = inspect.signature(mine[name]).parameters
        ref_pos = [p for p in ref["params"] if p["kind"] == "POSITIONAL_OR_KEYWORD"]
        my_pos = [k for k, v in sig.items() if v.kind == v.POSITIONAL_OR_KEYWORD]
        assert my_pos[:len(ref_pos)] == [p["name"] for p in ref_pos], (name, my_pos)
        for p in ref_pos:
            v = sig[p["name"]]
            assert (v.default is not inspect.Parameter.empty) == p["has_default"], (name, p["name"])
            if p["has_default"] and not (name == "CombinedActions" and p["name"] == "actions"):   # [] there, None here: same meaning
                assert repr(v.default) == p["default"], (name, p["name"], v.default, p["default"])
        if any(p["kind"] == "VAR_KEYWORD" for p in ref["params"]):
            assert any(v.kind == v.VAR_KEYWORD for v in sig.values()), name
        # what the mirror adds is keyword-only or defaulted: a reference call never has to change
        for k, v in sig.items():
            if k not in [p["name"] for p in ref["params"]] and v.kind in (v.POSITIONAL_OR_KEYWORD, v.KEYWORD_ONLY):
                assert v.default is not inspect.Parameter.empty, (name, k)
        for m in ref.get("protocol") or []:
            assert callable(getattr(mine[name], m, None)), (name, m)


def test_chain_refinement_guard_measures_the_displacement_at_the_corners():
    """pipeline._corner_shift: the largest displacement between two 2 x 3 transforms at the four frame corners -- what decides
    whether a refinement against the global reference frame is trusted (CHAIN_REFINE_MAX_SHIFT); the chained entry point
    refines by default."""
    import inspect
    from shinestacker_amd import pipeline
    h, w = 400, 600
    eye = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert pipeline._corner_shift(eye, eye, h, w) == 0.0
    assert pipeline._corner_shift(eye, eye + np.array([[0, 0, 1.5], [0, 0, -0.25]]), h, w) == 1.5
    rot = np.array([[1.0, -1e-3, 0], [1e-3, 1.0, 0]])          # a rotation about the origin moves the far corner most
    assert abs(pipeline._corner_shift(eye, rot, h, w) - 1e-3 * (w - 1)) < 1e-9
    assert pipeline.CHAIN_REFINE_MAX_SHIFT == 2.0
    assert inspect.signature(pipeline.align_and_stack_device).parameters["chain_refine"].default is True


def test_default_arith_follows_the_float_type_in_every_spelling(caplog):
    """defaults.resolve_arith: float-64 stacks run the exact order (the separable kernels are float-32) whichever way the caller
    names the type -- the reference's constant, NumPy's type, or the C code `_lib.Stack` takes (MI_F64) -- and the entry points of
    pipeline.py hand it the caller's float_type (round-5 advice: they passed none, so MI_F64 + no arith asked for 'separable'
    and mi_stack_create refused)."""
    import inspect
    from shinestacker_amd import _lib, constants, pipeline
    from shinestacker_amd.defaults import resolve_arith
    assert resolve_arith() == constants.DEFAULT_PY_ARITH == "separable"
    for ft in (constants.FLOAT_64, "float64", np.float64, _lib.MI_F64):
        assert resolve_arith(None, ft) == "exact", ft
        assert resolve_arith("separable", ft) == "separable"      # an explicit choice is the caller's (and the library's to refuse)
    for ft in (None, constants.FLOAT_32, np.float32, _lib.MI_F32, True):
        assert resolve_arith(None, ft) == "separable", ft
    for fn in (pipeline.align_and_stack, pipeline.align_and_stack_device, pipeline.bunches_then_stack):
        assert 'resolve_arith(stack_kwargs.get("arith"), stack_kwargs.get("float_type"))' in inspect.getsource(fn), fn.__name__


def test_auto_estimator_says_so_when_it_is_not_the_references(caplog, monkeypatch):
    """align.auto_estimator: without OpenCV the reference's SIFT + RANSAC recipe (align.py:90-151) cannot run and the GPU ECC
    estimator registers the frames -- another algorithm, so the swap goes to the log (once)."""
    import logging
    from shinestacker_amd import align
    monkeypatch.setattr(align, "have_opencv", lambda: False)
    monkeypatch.setattr(align, "_auto_fallback_logged", False)
    with caplog.at_level(logging.WARNING, logger="shinestacker_amd"):
        est = align.resolve_estimator("auto")
        align.resolve_estimator(None)
    assert callable(est) and est is not align.opencv_estimator
    msgs = [r.getMessage() for r in caplog.records if "estimator='auto'" in r.getMessage()]
    assert len(msgs) == 1 and "ECC" in msgs[0] and "align.py:90-151" in msgs[0]
    monkeypatch.setattr(align, "have_opencv", lambda: True)
    assert align.resolve_estimator("auto") is align.opencv_estimator
