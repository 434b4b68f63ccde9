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
for n in (1, 1023, 1024, 1025, 300007):
    world = 3
    cand = torch.randint(0, 50, (world, n), generator=g).float().to(dev)     # many exact ties
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
