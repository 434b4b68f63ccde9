"""Build libmi355stack.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m shinestacker_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.environ.get("MI355STACK_LIB") or os.path.join(CSRC, "libmi355stack.so")
SOURCES = ["capi.hip"]


def _deps():
    """every source the library is built from: csrc/*.hip, csrc/*.hpp and the public header"""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + \
        [os.path.join(ROOT, "include", "mi355stack.h"), os.path.abspath(__file__)]


HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fno-fast-math", "-fno-slp-vectorize", "-shared", "-fPIC", "-fvisibility=hidden",
               "-Wall", "-Wno-unused-function", "-pthread"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for p in _deps():
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_extension(force=False, verbose=False):
    # tuning flags from the environment always rebuild (they are not part of the time stamps)
    force = force or bool(os.environ.get("MI_TILE_CFG") or os.environ.get("MI_EXTRA_FLAGS"))
    if not force and not needs_build():
        return LIB
    # one builder at a time (bench.py is started once per GPU): the others wait, then find it built
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    extra = []
    cfg = os.environ.get("MI_TILE_CFG")  # "TH0,TW0,NT0,TH,TW,NT,PAD[,RU]" -- tuning builds only
    if cfg:
        names = ["MI_TILE0_H", "MI_TILE0_W", "MI_TILE0_NT", "MI_TILE_H", "MI_TILE_W", "MI_TILE_NT",
                 "MI_TILE_PAD", "MI_REDUCE_RU"]
        extra = [f"-D{n}={v}" for n, v in zip(names, cfg.split(","))]
    extra += os.environ.get("MI_EXTRA_FLAGS", "").split()
    cmd = [_hipcc(), *HIPCC_FLAGS, *extra, "-I", os.path.join(ROOT, "include"),
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)   # readers never see a half-written library
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
