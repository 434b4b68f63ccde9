"""CPU: the C-ABI library loads and exports every symbol include/mi355stack.h declares;
without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mi355stack.h")


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mi355stack.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for must in ("mi_stack_create", "mi_stack_push_frame", "mi_stack_push_frames_device",
                 "mi_stack_finish", "mi_stack_reset", "mi_stack_destroy", "mi_last_error",
                 "mi_stack_get_level", "mi_combine_select"):
        assert must in syms


def test_library_exports_every_declared_symbol(hiplib):
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in mi355stack.h but not exported"


def test_python_binding_covers_header(hiplib):
    assert sorted(hiplib.SIGNATURES) == header_symbols()
    assert hiplib.load().mi_abi_version() == 1


def _header_params_fields():
    """(name, ctype, count) of every member of mi_stack_params_t, parsed from include/mi355stack.h"""
    import re
    text = open(HEADER).read()
    body = text[text.index("typedef struct mi_stack_params {"):text.index("} mi_stack_params_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        for nm in names.split(","):
            m = re.match(r"\s*(\w+)(?:\[(\d+)\])?\s*$", nm)
            fields.append((m.group(1), ctype, int(m.group(2) or 1)))
    return fields


def test_params_struct_layout(hiplib):
    """mi_stack_params_t as the header declares it == the ctypes mirror (names, order, OFFSETS, size) == the stub shown
    in INTEGRATION.md: int32 x6, double, int32 x5 (float_type .. batch_frames), arith, pair_levels, reserved[3] -> 72 bytes."""
    ct = {"int32_t": ctypes.c_int32, "double": ctypes.c_double}
    fields = _header_params_fields()

    class FromHeader(ctypes.Structure):
        _fields_ = [(n, ct[t] * k if k > 1 else ct[t]) for n, t, k in fields]
    mine = hiplib.StackParams
    assert [f[0] for f in mine._fields_] == [n for n, _, _ in fields]
    for n, _, _ in fields:
        assert getattr(mine, n).offset == getattr(FromHeader, n).offset, n
        assert getattr(mine, n).size == getattr(FromHeader, n).size, n
    assert ctypes.sizeof(mine) == ctypes.sizeof(FromHeader) == 72
    assert mine.arith.offset == 52 and mine.pair_levels.offset == 56 and mine.reserved.offset == 60 and mine.gen_kernel.offset == 24
    # the reference-side stub of INTEGRATION.md lists the same members in the same order
    import re
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class _Params(C.Structure):"):doc.index("_lib.mi_last_error.restype")]
    assert re.findall(r'\("(\w+)", C\.', stub) == [n for n, _, _ in fields]
    p = hiplib.StackParams()
    hiplib.load().mi_stack_default_params(ctypes.byref(p))
    assert (p.min_size, p.kernel_size, p.gen_kernel, p.use_fma, p.arith) == (32, 5, 0.4, 1, 0)


def test_argument_validation_without_gpu(hiplib):
    lib = hiplib.load()
    assert lib.mi_stack_create(None, None) == hiplib.MI_ERR_INVALID
    assert b"null" in lib.mi_last_error()
    assert lib.mi_stack_levels(None, None) == hiplib.MI_ERR_INVALID


@pytest.mark.skipif(os.environ.get("MI_EXPECT_GPU") == "1", reason="GPU box")
def test_no_gpu_means_loud_failure(hiplib):
    if hiplib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    from shinestacker_amd import PyramidStack, DeviceError
    with pytest.raises(DeviceError):
        hiplib.Stack(64, 64)
    algo = PyramidStack()
    with pytest.raises(DeviceError):
        algo.focus_stack_arrays([np.zeros((64, 64, 3), np.uint8)])


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under shinestacker_amd/ may reference it."""
    pkg = os.path.join(ROOT, "shinestacker_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn
                assert "liboracle" not in txt, fn


def test_release_library_reads_no_environment_variable():
    """The timing-study knobs (MI_ABLATE, MI_TAPER, MI_SERIAL, MI_CHUNK, MI_LAUNCH_FRAMES, ...) exist only in the -DMI_STUDY
    build: the shipped library neither imports getenv nor contains their names, so a stray environment variable cannot
    change a user's stack."""
    import re
    import subprocess
    from shinestacker_amd import build as b
    lib = b.LIB if os.path.exists(b.LIB) else b.build_extension()
    data = open(lib, "rb").read()
    names = set(re.findall(rb"MI_[A-Z][A-Z0-9_]{3,}", data))
    knobs = {n for n in names if n in (b"MI_ABLATE", b"MI_TAPER", b"MI_SERIAL", b"MI_CHUNK", b"MI_LAUNCH_FRAMES", b"MI_ONLY_L0",
                                         b"MI_WIDE_LEVELS", b"MI_BD_PRIO", b"MI_CO_PRIO", b"MI_PAR_TILES", b"MI_ECC_STEP",
                                         b"MI_ECC_PER_BLOCK")}
    assert not knobs, knobs
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in und
