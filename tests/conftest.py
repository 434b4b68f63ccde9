import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        d = {k: z[k] for k in z.files}
    if "params" in d:
        d["params"] = json.loads(str(d["params"]))
    return d


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hiplib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    from shinestacker_amd import build, _lib
    build.build_extension()
    return _lib


def fusion_cases():
    return ["g1_u8", "g2_u16", "g3_ties", "g1b_nofma", "g1c_smooth", "g8_nolevels"]


def f64_cases():
    return ["g9_f64", "g9_f64_u16"]


def stack_kwargs(params):
    kw = {k: params[k] for k in ("min_size", "kernel_size", "gen_kernel") if k in params}
    kw["use_fma"] = params.get("use_fma", True)
    return kw
