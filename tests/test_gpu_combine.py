"""GPU: the local kernels of the winners-only combine (mi_combine_winner / _plan / _pack / _unpack) against their torch
statements on the same tensors, for a simulated world of three ranks, and Combiner.combine_winners() under RCCL with one
rank (fresh process: torch's HIP runtime has to be loaded before libmi355stack.so)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
torch.cuda.init()
sys.path.insert(0, os.getcwd())
from shinestacker_amd import _lib as L
from shinestacker_amd.multigpu import Combiner, TorchWinnerOps, first_max_rank
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
st = L.Stack(200, 296)
hip, ref = Combiner(st), TorchWinnerOps()
g = torch.Generator(device="cpu").manual_seed(3)
# worlds 3 / 5 / 8 use one / two 64-bit words of packed per-rank counts in the block scan, 13 / 16 all four
for n, world in ((1, 3), (1023, 3), (1024, 3), (1025, 3), (300007, 3), (70001, 5), (262144, 8), (99999, 13), (131077, 16), (4099, 2)):
    cand = torch.randint(0, 50, (world, n), generator=g).float().to(dev)     # many exact ties
    if world == 8:
        cand[5] = 100.0 + torch.arange(n, device=dev) % 7                     # one rank wins whole 1024-pixel blocks
    win = hip.winner(cand)
    assert torch.equal(win, first_max_rank(cand)), n
    plan, totals = hip.plan(win, world)
    assert totals == ref.plan(win, world)[1] and sum(totals) == n
    for width in (1, 3):
        arr = torch.rand(n * width, generator=g).to(dev)
        packs = [hip.pack(win, plan, world, r, arr, width, totals[r]) for r in range(world)]
        for r in range(world):
            assert torch.equal(packs[r], ref.pack(win, None, world, r, arr, width, totals[r])), (n, width, r)
        # rank 1 receives the rows of ranks 0 and 2 into an array that holds its own rows already
        for me in range(world):
            mine = torch.where((win == me).repeat_interleave(width), arr, torch.full_like(arr, -1.0))
            want = mine.clone()
            bufs = [None if r == me else packs[r] for r in range(world)]
            hip.unpack(win, plan, world, me, bufs, mine, width)
            ref.unpack(win, None, world, me, bufs, want, width)
            assert torch.equal(mine, want) and torch.equal(mine, arr), (n, width, me)
# the whole protocol with one rank: nothing to exchange, the stack's result is unchanged; sync_level(0) then sync
rng = np.random.default_rng(5)
frames = [rng.integers(0, 256, (200, 296, 3), dtype=np.uint8) for _ in range(5)]
for arith in ("exact", "separable"):
    a = L.Stack(200, 296, arith=arith)
    for f in frames: a.push_frame(f)
    want = a.finish()
    b = L.Stack(200, 296, arith=arith)
    for f in frames: b.push_frame(f)
    Combiner(b).combine_winners()
    assert np.array_equal(b.finish(), want)
dist.destroy_process_group()
print("WINNERS_OK")
'''


def test_winner_kernels_and_protocol_world_1():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "WINNERS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


TWO_RANKS = r'''
import os, sys, json
import numpy as np
import torch, torch.distributed as dist
torch.cuda.init()
sys.path.insert(0, os.getcwd())
from shinestacker_amd import _lib as L
from shinestacker_amd.multigpu import Combiner, HostStagedComm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)      # ONE GPU for both processes: RCCL refuses that
H, W, N = 300, 452, 12
rng = np.random.default_rng(17)
frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(N)]
frames[7] = frames[1].copy()      # the same frame on both ranks: exact ties across ranks, the lower global index must win
frames[9] = frames[4].copy()
frames[3] = frames[1].copy()      # ... and inside one rank
per = N // world
for arith in ("exact", "separable"):
    st = L.Stack(H, W, arith=arith)
    st.set_first_index(rank * per)
    for f in frames[rank * per:(rank + 1) * per]:
        st.push_frame(f)
    cb = Combiner(st, comm=HostStagedComm(dist.group.WORLD))      # the library's HIP kernels, not TorchWinnerOps
    assert type(cb).winner is Combiner.winner
    cb.combine_winners(with_index=True, root_energy=True)
    if rank == 0:
        whole = L.Stack(H, W, arith=arith)
        for f in frames:
            whole.push_frame(f)
        for lv in range(st.levels):
            for tap in (L.TAP_ENERGY, L.TAP_INDEX, L.TAP_FUSED_LAP):
                assert np.array_equal(st.tap(tap, lv), whole.tap(tap, lv)), (arith, lv, tap)
        assert np.array_equal(st.finish(), whole.finish()), arith
        whole.close()
        print("TIMINGS", arith, json.dumps(cb.timings))
    # the image alone needs the winners' Laplacians / base pixels only
    st2 = L.Stack(H, W, arith=arith)
    st2.set_first_index(rank * per)
    for f in frames[rank * per:(rank + 1) * per]:
        st2.push_frame(f)
    Combiner(st2, comm=HostStagedComm(dist.group.WORLD)).combine_winners(with_index=False, root_energy=False)
    if rank == 0:
        ref = L.Stack(H, W, arith=arith)
        for f in frames:
            ref.push_frame(f)
        assert np.array_equal(st2.finish(), ref.finish()), arith
        ref.close()
    st.close(); st2.close()
# INTERLEAVED shards (rank r holds frames r, r + world, ...: set_first_index(r, world)): the rank order is not the frame order
# any more, ties go by the global frame index -- frame 6 (rank 0) repeats frame 5 (rank 1): 5 must win although its rank is higher
frames[6] = frames[5].copy()
for arith in ("exact", "separable"):
    for kw in (dict(with_index=True, root_energy=True), dict(with_index=False, root_energy=False)):
        st = L.Stack(H, W, arith=arith)
        st.set_first_index(rank, world)
        for f in frames[rank::world]:
            st.push_frame(f)
        Combiner(st, comm=HostStagedComm(dist.group.WORLD)).combine_winners(**kw)
        if rank == 0:
            whole = L.Stack(H, W, arith=arith)
            for f in frames:
                whole.push_frame(f)
            if kw["with_index"]:
                for lv in range(st.levels):
                    for tap in (L.TAP_ENERGY, L.TAP_INDEX, L.TAP_FUSED_LAP):
                        assert np.array_equal(st.tap(tap, lv), whole.tap(tap, lv)), ("interleaved", arith, lv, tap)
                for tap in (L.TAP_BASE_IDX_E, L.TAP_BASE_IDX_D):
                    assert np.array_equal(st.tap(tap, st.levels), whole.tap(tap, st.levels)), ("interleaved", arith, tap)
            assert np.array_equal(st.finish(), whole.finish()), ("interleaved", arith)
            whole.close()
        st.close()
# a rank's own view: the index taps show global frame numbers, and an exported handle takes no more frames until reset
st = L.Stack(H, W, arith="separable")
st.set_first_index(rank, world)
for f in frames[rank::world]:
    st.push_frame(f)
own = st.tap(L.TAP_INDEX, 0)
assert set(np.unique(own)) <= set(range(rank, N, world))
try:
    st.push_frame(frames[0])
    raise SystemExit("push after export did not fail")
except RuntimeError:
    pass
st.reset()
st.push_frame(frames[0])
st.close()
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("TWO_RANKS_OK")
'''


def test_device_combine_with_two_ranks_on_one_gpu(tmp_path):
    """`Combiner.combine_winners` with world == 2 and the library's HIP kernels (mi_combine_winner / plan / pack / unpack on
    the device-resident slabs): two processes share the one GPU, the collectives travel over gloo through the host
    (`HostStagedComm`) because RCCL refuses two ranks on one device.  Cross-rank duplicate frames: the first maximum in
    global frame order must win (pyramid.py:48-55).  Both arithmetics, both exchange variants; contiguous frame blocks and
    interleaved shards (set_first_index(rank, world): ties by the global frame index, mi_combine_winner_idx)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "two_ranks.py"
    script.write_text(TWO_RANKS)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "TWO_RANKS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


FORCED = r'''
import os, sys, json
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.getcwd())
from shinestacker_amd import _lib as L
from shinestacker_amd.multigpu import Combiner
rng = np.random.default_rng(5)
frames = [rng.integers(0, 256, (200, 296, 3), dtype=np.uint8) for _ in range(5)]
a = L.Stack(200, 296, arith="separable")
for f in frames: a.push_frame(f)
want = a.finish()
b = L.Stack(200, 296, arith="separable")
for f in frames: b.push_frame(f)
cb = Combiner(b, force=True)
cb.combine_winners()
assert np.array_equal(b.finish(), want)
assert set(cb.timings) == {"wait_level0_ms", "exchange_level0_ms", "wait_rest_ms", "exchange_rest_ms"}
print("FORCED_OK", json.dumps(cb.timings))
'''


def test_forced_combine_at_world_1_runs_the_kernels_and_changes_nothing():
    """bench.py's `combine_ms`: the per-rank kernel work of the exchange (winner map, plan, pack, unpack) run on a single
    rank's own rows -- nothing moves, the result is unchanged, the phases are timed."""
    r = subprocess.run([sys.executable, "-c", FORCED], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "FORCED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_flow_with_two_ranks_on_one_gpu(scaling):
    """bench.py's multi-rank path end to end on ONE GPU (MI_BENCH_ONE_GPU=1: both ranks on device 0, collectives staged
    through the host over gloo): frame sharding with global indices, the device combine kernels, rank 0's collapse, the
    max-over-ranks timing and the verification of the result against the oracle fed ALL frames."""
    import json
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MI_BENCH_ONE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--frames", "8", "--height", "600",
                        "--width", "900", "--steps", "1", "--warmup", "1", "--scaling", scaling, "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["verified"] is True, d
    assert d["combine_ms"] > 0 and "SHARING ONE GPU" in d["config"]["parallelism"]
